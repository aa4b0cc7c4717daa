#!/bin/bash
# rocprofv3 kernel stats of the sequential-frame bench (the run `roofline.*` is cross-checked against) -> gpurun_out/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/bench.py --steps 5 --warmup 2 --inflight 1 --only-headline --no-roofline > /dev/null 2>&1   # fills the tuning cache
rm -rf /tmp/pr1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr1 -- python $R/bench.py --steps 30 --warmup 5 --inflight 1 --only-headline --per-shape > $R/gpurun_out/r02_bench_seq.json 2> /dev/null
cp $(find /tmp/pr1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r02_kernel_stats_seq.csv
python $R/bench.py --steps 30 --warmup 5 --inflight 1 --only-headline --per-shape > $R/gpurun_out/r02_bench_seq_noprof.json 2> /dev/null
head -12 $R/gpurun_out/r02_kernel_stats_seq.csv | cut -c1-160
