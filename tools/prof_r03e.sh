#!/bin/bash
# Round-3 (final) rocprofv3 kernel summaries of the V2X-ViT autocast frame (8 agents) -> gpurun_out/r03e_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
prof() {   # name, bench args...
    local name=$1; shift
    python $R/bench.py "$@" --steps 3 --warmup 1 --only-headline --no-roofline > /dev/null 2>&1
    rm -rf /tmp/pr_$name
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_$name -- python $R/bench.py "$@" --only-headline > $O/r03e_bench_${name}_profiled.json 2> $O/r03e_bench_${name}.err
    python $R/tools/kernel_stats_csv.py "$(find /tmp/pr_$name -name '*kernel_stats.csv' | head -1)" > $O/r03e_kernel_stats_${name}.txt
    head -30 $O/r03e_kernel_stats_${name}.txt | cut -c1-150
}
prof v2xvit_n8_amp_inflight1 --model v2xvit --agents 8 --amp --cpu-frames 0 --steps 10 --warmup 2 --inflight 1
