#!/bin/bash
# Round 5: rocprofv3 kernel summaries of the other BASELINE configurations (one frame at a time) -> gpurun_out/r05_kernel_stats_*.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
run() { name=$1; shift
  python $R/bench.py --only-headline --no-roofline --steps 3 --warmup 2 "$@" > /dev/null 2>&1
  rm -rf /tmp/pm_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm_$name -- python $R/bench.py --only-headline --no-roofline --inflight 1 --steps 12 --warmup 3 "$@" > $O/r05_bench_${name}_inflight1_profiled.json 2> /dev/null
  python $R/tools/kernel_stats_csv.py "$(find /tmp/pm_$name -name '*kernel_stats.csv' | head -1)" > $O/r05_kernel_stats_$name.txt
  head -12 $O/r05_kernel_stats_$name.txt | cut -c1-150
}
run cobevt_n8 --model cobevt --agents 8
run v2xvit_n8 --model v2xvit --agents 8
run v2xvit_n8_amp --model v2xvit --agents 8 --amp
run cam_lidar_n8 --modalities cam,lidar --agents 8
