timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/r02_final_n1.json 2> gpurun_out/bench_err.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02_final_n1.json").read().strip().splitlines()[-1])
print(d["value"], d["single_stream"]["frames_per_s"], d["train_step"]["ms_per_step"], d["roofline"]["frac"], d["roofline"]["mfma_executed"]["frac"])
PY
