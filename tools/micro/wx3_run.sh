cd /root/repo
for b in 0e 64e 0s1 0e; do
  ./tools/micro/wx3_ablate_$b 64 64 4 25 88 256; ./tools/micro/wx3_ablate_$b 64 64 4 25 88 16; ./tools/micro/wx3_ablate_$b 32 128 4 25 88 256; ./tools/micro/wx3_ablate_$b 64 64 4 100 352 256
done
