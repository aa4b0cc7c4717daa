#!/bin/bash
# rocprofv3 kernel stats of the training-step bench (round 3) -> gpurun_out/r03_kernel_stats_train.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/train_bench.py --steps 2 --warmup 2 > /dev/null 2>&1
rm -rf /tmp/pt1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt1 -- python $R/tools/train_bench.py --steps 10 --warmup 3 > $R/gpurun_out/r03_train_bench_profiled.json 2> /dev/null
python $R/tools/kernel_stats_csv.py "$(find /tmp/pt1 -name '*kernel_stats.csv' | head -1)" > $R/gpurun_out/r03_kernel_stats_train.txt
head -30 $R/gpurun_out/r03_kernel_stats_train.txt | cut -c1-150
