#!/bin/bash
# Round-3 rocprofv3 kernel summaries -> gpurun_out/r03_* (copied into profiles/ by hand).  Run on the GPU box:
#   gpurun -- 'bash tools/prof_r03.sh'
# One `rocprofv3 --kernel-trace --stats` run per mode (no counters here); each bench run is preceded by an unprofiled one that
# fills the tile-tuning cache, so the profiled process times no candidate tiles.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
prof() {   # name, bench args...
    local name=$1; shift
    python $R/bench.py "$@" --steps 3 --warmup 1 --only-headline --no-roofline > /dev/null 2>&1
    rm -rf /tmp/pr_$name
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_$name -- python $R/bench.py "$@" --only-headline > $O/r03_bench_${name}_profiled.json 2> $O/r03_bench_${name}.err
    python $R/tools/kernel_stats_csv.py "$(find /tmp/pr_$name -name '*kernel_stats.csv' | head -1)" > $O/r03_kernel_stats_${name}.txt
    head -12 $O/r03_kernel_stats_${name}.txt | cut -c1-150
}
prof cam_lidar_n8_inflight1 --modalities cam,lidar --agents 8 --steps 10 --warmup 2 --inflight 1 --per-shape
prof headline --steps 30 --warmup 5
prof headline_inflight1 --steps 30 --warmup 5 --inflight 1 --per-shape
prof cobevt_n8 --model cobevt --agents 8 --steps 10 --warmup 2 --inflight 1
prof v2xvit_n8 --model v2xvit --agents 8 --steps 10 --warmup 2 --inflight 1
prof v2xvit_n8_amp --model v2xvit --agents 8 --amp --steps 10 --warmup 2 --inflight 1
