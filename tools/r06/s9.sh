#!/bin/bash
# round 6, GPU session 9: evidence for the other BASELINE configurations (rocprofv3 kernel summaries, one frame at a time) and the HBM-side traffic
# of the headline's dominant kernel re-measured (two PMC passes), named per round
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06i; mkdir -p $O
run() { name=$1; shift
  python $R/bench.py --only-headline --no-roofline --steps 3 --warmup 2 "$@" > /dev/null 2>&1
  rm -rf /tmp/pm_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm_$name -- python $R/bench.py --only-headline --no-roofline --inflight 1 --steps 12 --warmup 3 "$@" > $O/bench_${name}_inflight1_profiled.json 2> /dev/null
  python $R/tools/kernel_stats_csv.py "$(find /tmp/pm_$name -name '*kernel_stats.csv' | head -1)" > $O/kernel_stats_$name.txt
  head -8 $O/kernel_stats_$name.txt | cut -c1-150
}
run agents8 --agents 8
run cobevt_n8 --model cobevt --agents 8
run v2xvit_n8 --model v2xvit --agents 8
run v2xvit_n8_amp --model v2xvit --agents 8 --amp
run cam_lidar_n8 --modalities cam,lidar --agents 8
F=""; W=""
pass() { local name=$1; shift
    python $R/bench.py "$@" --steps 2 --warmup 1 --only-headline --no-roofline > /dev/null 2>&1
    for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmc_${name}_$c
        timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${name}_$c -- python $R/bench.py "$@" --steps 3 --warmup 1 --only-headline --no-roofline > /dev/null 2>&1
    done
    F="$F /tmp/pmc_${name}_FETCH_SIZE"; W="$W /tmp/pmc_${name}_WRITE_SIZE"
}
pass head3 --inflight 3
pass agents8 --agents 8 --inflight 3
cd $R && python tools/pmc_traffic.py --fetch $F --write $W --merge profiles/pmc_hbm.json -o $O/pmc_hbm.json && python -c "
import json; d=json.load(open('$O/pmc_hbm.json'))['per_kernel']
for k in sorted(d):
    if 'w4_' in k or 'wino4' in k: print(k, {w: round((e['fetch_bytes']+e['write_bytes'])/1e6,1) for w,e in d[k].items()})"
