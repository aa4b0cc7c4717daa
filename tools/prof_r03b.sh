#!/bin/bash
# Round-3 (second half) rocprofv3 kernel summaries of the bf16-activation V2X-ViT AMP frame -> gpurun_out/r03b_*.
#   gpurun -- 'bash tools/prof_r03b.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
prof() {   # name, bench args...
    local name=$1; shift
    python $R/bench.py "$@" --steps 3 --warmup 1 --only-headline --no-roofline > /dev/null 2>&1
    rm -rf /tmp/pr_$name
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_$name -- python $R/bench.py "$@" --only-headline > $O/r03b_bench_${name}_profiled.json 2> $O/r03b_bench_${name}.err
    python $R/tools/kernel_stats_csv.py "$(find /tmp/pr_$name -name '*kernel_stats.csv' | head -1)" > $O/r03b_kernel_stats_${name}.txt
    head -24 $O/r03b_kernel_stats_${name}.txt | cut -c1-150
}
prof v2xvit_n8_amp --model v2xvit --agents 8 --amp --steps 10 --warmup 2 --inflight 1
