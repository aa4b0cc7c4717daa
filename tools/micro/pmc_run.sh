#!/bin/bash
# HBM-side traffic of the conv kernels of the sequential-frame bench: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE;
# counters only, with --kernel-trace) -> profiles/pmc_hbm.json via tools/pmc_traffic.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/bench.py --steps 3 --warmup 1 --inflight 1 --only-headline --no-roofline > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -- python $R/bench.py --steps 10 --warmup 2 --inflight 1 --only-headline --no-roofline > /dev/null 2>&1
done
cd $R && python tools/pmc_traffic.py --fetch /tmp/pmc_FETCH_SIZE --write /tmp/pmc_WRITE_SIZE -o gpurun_out/r02_pmc_hbm.json
python - <<PY
import json
d=json.load(open("$R/gpurun_out/r02_pmc_hbm.json"))
for k,v in d["per_kernel"].items():
    for w,e in v.items(): print(k,w,e)
PY
