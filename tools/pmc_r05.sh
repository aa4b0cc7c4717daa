#!/bin/bash
# Round 5: HBM-side traffic of the x3-mode bench lines (conv_wino_x3, conv_wino4_x3, conv_igemm_x3p at the workgroup counts each
# configuration launches).  Two rocprofv3 PMC passes per mode (FETCH_SIZE, WRITE_SIZE; counters only, with --kernel-trace), merged INTO
# profiles/pmc_hbm.json (tools/pmc_traffic.py --merge keeps the earlier entries and the r03 calibration).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
F=""; W=""
pass() {   # name, bench args
    local name=$1; shift
    python $R/bench.py "$@" --steps 2 --warmup 1 --only-headline --no-roofline > /dev/null 2>&1
    for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmc_${name}_$c
        timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${name}_$c -- python $R/bench.py "$@" --steps 3 --warmup 1 --only-headline --no-roofline > /dev/null 2>&1
    done
    F="$F /tmp/pmc_${name}_FETCH_SIZE"; W="$W /tmp/pmc_${name}_WRITE_SIZE"
}
pass head3 --inflight 3
pass head1 --inflight 1
pass agents8 --agents 8 --inflight 3
pass v2xvit8 --model v2xvit --agents 8 --inflight 1
pass cobevt8 --model cobevt --agents 8 --inflight 1
pass cam8 --modalities cam,lidar --agents 8 --inflight 1
cd $R && python tools/pmc_traffic.py --fetch $F --write $W --merge profiles/pmc_hbm.json -o gpurun_out/r05_pmc_hbm.json && python -c "
import json; d=json.load(open('gpurun_out/r05_pmc_hbm.json'))['per_kernel']
for k in sorted(d):
    if 'bf16x3' in k: print(k, {w: round((e['fetch_bytes']+e['write_bytes'])/1e6,1) for w,e in d[k].items()})"
