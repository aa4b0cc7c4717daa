#!/bin/bash
# rocprofv3 kernel stats of the bench: the sequential-frame run (what `roofline.*` is cross-checked against) and the
# default pipelined run -> gpurun_out/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/bench.py --steps 5 --warmup 2 --inflight 1 --only-headline --no-roofline > /dev/null 2>&1   # fills the tuning cache
rm -rf /tmp/pr1 /tmp/pr2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr1 -- python $R/bench.py --steps 30 --warmup 5 --inflight 1 --only-headline --per-shape > $R/gpurun_out/r02_bench_seq_profiled.json 2> /dev/null
cp $(find /tmp/pr1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r02_kernel_stats_seq.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr2 -- python $R/bench.py --steps 30 --warmup 5 --only-headline --no-roofline > $R/gpurun_out/r02_bench_pipelined_profiled.json 2> /dev/null
cp $(find /tmp/pr2 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r02_kernel_stats_pipelined.csv
python $R/bench.py --steps 30 --warmup 5 --inflight 1 --only-headline --per-shape > $R/gpurun_out/r02_bench_seq_noprof.json 2> /dev/null
head -14 $R/gpurun_out/r02_kernel_stats_seq.csv | cut -c1-170
