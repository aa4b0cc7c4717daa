#!/bin/bash
# Round-3 final: rocprofv3 kernel summaries of the training step and of the 8-agent Where2Comm frame (one frame at a time) -> gpurun_out/r03f_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
python $R/tools/train_bench.py --steps 2 --warmup 2 > /dev/null 2>&1
rm -rf /tmp/pt1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt1 -- python $R/tools/train_bench.py --steps 10 --warmup 3 > $O/r03f_train_bench_profiled.json 2> /dev/null
python $R/tools/kernel_stats_csv.py "$(find /tmp/pt1 -name '*kernel_stats.csv' | head -1)" > $O/r03f_kernel_stats_train.txt
head -16 $O/r03f_kernel_stats_train.txt | cut -c1-150
rm -rf /tmp/pa8
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa8 -- python $R/bench.py --agents 8 --inflight 1 --cpu-frames 0 --only-headline > $O/r03f_bench_agents8_inflight1_profiled.json 2> /dev/null
python $R/tools/kernel_stats_csv.py "$(find /tmp/pa8 -name '*kernel_stats.csv' | head -1)" > $O/r03f_kernel_stats_agents8_inflight1.txt
head -12 $O/r03f_kernel_stats_agents8_inflight1.txt | cut -c1-150
