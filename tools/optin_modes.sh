for m in cobevt v2xvit when2com; do
  for f in "--gemm split3" "--amp"; do
    timeout 300 python bench.py --model $m $f --steps 10 --warmup 3 2>&1 | tail -1 > gpurun_out/optin.json
    python - "$m" "$f" <<'PY'
import sys, json
r = json.loads(open("gpurun_out/optin.json").read())
print("OPTIN", sys.argv[1], sys.argv[2], r["value"], r.get("single_stream", {}).get("frames_per_s"))
PY
  done
done
