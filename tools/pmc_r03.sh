#!/bin/bash
# Round-3 HBM-side traffic of the conv kernels: per bench mode two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; counters only, with
# --kernel-trace), a calibration pair (tools/pmc_calib.py, incl. the dword-gather pattern), merged into gpurun_out/r03_pmc_hbm.json.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
F=""; W=""
pass() {   # name, bench args
    local name=$1; shift
    python $R/bench.py "$@" --steps 2 --warmup 1 --only-headline --no-roofline > /dev/null 2>&1
    for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmc_${name}_$c
        timeout 900 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${name}_$c -- python $R/bench.py "$@" --steps 4 --warmup 1 --only-headline --no-roofline > /dev/null 2>&1
    done
    F="$F /tmp/pmc_${name}_FETCH_SIZE"; W="$W /tmp/pmc_${name}_WRITE_SIZE"
}
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_calib_$c
    rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_calib_$c -- python $R/tools/pmc_calib.py > /dev/null 2>&1
done
pass head3 --inflight 3
pass head1 --inflight 1
pass amp --amp --inflight 1
pass cobevt8 --model cobevt --agents 8 --inflight 1
pass v2xvit8 --model v2xvit --agents 8 --inflight 1
pass v2xvit8amp --model v2xvit --agents 8 --amp --inflight 1
pass cam8 --modalities cam,lidar --agents 8 --inflight 1
cd $R && python tools/pmc_traffic.py --fetch $F --write $W --calib-fetch /tmp/pmc_calib_FETCH_SIZE --calib-write /tmp/pmc_calib_WRITE_SIZE -o gpurun_out/r03_pmc_hbm.json
python - <<PY
import json
d=json.load(open("$R/gpurun_out/r03_pmc_hbm.json"))
print(d["calibration"])
print({k: len(v) for k, v in d["per_kernel"].items()})
PY
