#!/bin/bash
# Round 4: rocprofv3 kernel summary of the camera + LiDAR frame (8 agents, one frame at a time) on the x3 code -> gpurun_out/r04_*cam*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
python $R/bench.py --modalities cam,lidar --agents 8 --inflight 1 --cpu-frames 0 --only-headline --steps 3 --warmup 2 > /dev/null 2>&1
rm -rf /tmp/pc8
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc8 -- python $R/bench.py --modalities cam,lidar --agents 8 --inflight 1 --cpu-frames 0 --only-headline --steps 10 --warmup 2 > $O/r04_bench_cam_lidar_n8_inflight1_profiled.json 2> /dev/null
python $R/tools/kernel_stats_csv.py "$(find /tmp/pc8 -name '*kernel_stats.csv' | head -1)" > $O/r04_kernel_stats_cam_lidar_n8_inflight1.txt
head -34 $O/r04_kernel_stats_cam_lidar_n8_inflight1.txt | cut -c1-150
