#!/bin/bash
# rocprofv3 kernel stats of the training step bench -> gpurun_out/r02_kernel_stats_train.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/train_bench.py --steps 2 --warmup 2 > /dev/null 2>&1   # fills the tuning cache
rm -rf /tmp/pt1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt1 -- python $R/tools/train_bench.py --steps 10 --warmup 3 > $R/gpurun_out/r02_train_bench_profiled.json 2> /dev/null
cp $(find /tmp/pt1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r02_kernel_stats_train.csv
python $R/tools/train_bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 > $R/gpurun_out/r02_train_bench.json
cat $R/gpurun_out/r02_train_bench.json
head -40 $R/gpurun_out/r02_kernel_stats_train.csv | cut -c1-200
