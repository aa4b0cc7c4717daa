#!/bin/bash
# the round's bench lines: headline + the other models / agent counts / opt-in modes -> gpurun_out/r02_final_*.json
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
run() { name=$1; shift; python $R/bench.py "$@" 2>/dev/null | tail -1 > $O/r02_final_$name.json; python - <<PY
import json
d=json.load(open("$O/r02_final_$name.json")); r=d.get("roofline") or {}
print("$name", d["value"], d["unit"], "ms", d["ms_per_step"], "seq", (d.get("single_stream") or {}).get("frames_per_s"), "roof", r.get("achieved"), r.get("frac"), (r.get("mfma_executed") or {}).get("tflops"), r.get("kernel"))
PY
}
run n1 
run agents8 --agents 8 --cpu-frames 0
run cobevt_n4 --model cobevt --cpu-frames 0
run cobevt_n8 --model cobevt --agents 8 --cpu-frames 0
run v2xvit_n4 --model v2xvit --cpu-frames 0
run v2xvit_n8 --model v2xvit --agents 8 --cpu-frames 0
run when2com_n4 --model when2com --cpu-frames 0
run v2vnet_n4 --model v2vnet --cpu-frames 0
run n1_split3 --gemm split3 --cpu-frames 0
run n1_amp --amp --cpu-frames 0
run v2xvit_n8_amp --model v2xvit --agents 8 --amp --cpu-frames 0
AV2X_WINOGRAD=0 run n1_direct --cpu-frames 0
