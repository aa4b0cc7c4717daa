#!/bin/bash
# Round 4: GPU tests, the default bench line, the other configs, and the rocprofv3 kernel summary of the default command -> gpurun_out/r04_*.
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out
if [ "$1" != "notest" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04_gputest.txt; tail -3 gpurun_out/r04_gputest.txt
fi
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err || tail -5 gpurun_out/r04_bench_default.err
run() { name=$1; shift; python bench.py "$@" --cpu-frames 0 2>gpurun_out/r04_$name.err > gpurun_out/r04_$name.json || tail -3 gpurun_out/r04_$name.err; python -c "import json; d=json.load(open('gpurun_out/r04_$name.json')); r=d.get('roofline',{}); print('$name', d['value'], d['ms_per_step'], r.get('bound'), r.get('frac'), r.get('traffic'), r.get('traffic_over_algorithmic'), (r.get('kernel') or '')[:40])"; }
run default_nocpu
run f32 --gemm f32
run agents8 --agents 8
run cobevt_n8 --model cobevt --agents 8
run v2xvit_n8 --model v2xvit --agents 8
run v2xvit_n8_amp --model v2xvit --agents 8 --amp
run cam_lidar_n8 --modalities cam,lidar --agents 8 --steps 10 --warmup 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pr_head
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_head -- python $R/bench.py --only-headline > $R/gpurun_out/r04_bench_headline_profiled.json 2> $R/gpurun_out/r04_bench_headline.err
python $R/tools/kernel_stats_csv.py "$(find /tmp/pr_head -name '*kernel_stats.csv' | head -1)" > $R/gpurun_out/r04_kernel_stats_headline.txt
head -14 $R/gpurun_out/r04_kernel_stats_headline.txt | cut -c1-150
