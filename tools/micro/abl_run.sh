for z in 0 1; do for b in none 1 2 4 8 6 15; do
  if [ $b = none ]; then L=""; else L=tools/micro/libablate_$b.so; fi
  echo "== zero=$z ablate=$b"; AV2X_ZERO_DATA=$z AV2X_ABLATE_LIB=$L python tools/loop_peak.py 128x64w8d,128x128w8d 2>&1 | grep ks=
done; done
