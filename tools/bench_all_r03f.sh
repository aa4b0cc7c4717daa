#!/bin/bash
# Round-3 final pass: the bench lines whose roofline.traffic was null re-measured with the merged PMC table (profiles/pmc_hbm.json after
# tools/pmc_r03f.sh), and the rocprofv3 kernel summary of the DEFAULT bench command on the final code.
R=${GRAFT_REPO_ROOT:-.}
cd $R
run() { name=$1; shift; python bench.py "$@" --cpu-frames 0 2>/dev/null > gpurun_out/r03f_$name.json; python -c "import json; d=json.load(open('gpurun_out/r03f_$name.json')); r=d.get('roofline',{}); print('$name', d['value'], d['ms_per_step'], r.get('bound'), r.get('frac'), r.get('traffic'), r.get('traffic_over_algorithmic'), (r.get('kernel') or '')[:40])"; }
run agents8 --agents 8
run v2xvit_n8 --model v2xvit --agents 8
run cobevt_n8 --model cobevt --agents 8
run cam_lidar_n8 --modalities cam,lidar --agents 8 --steps 10 --warmup 2
run v2xvit_n8_amp --model v2xvit --agents 8 --amp
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pr_head
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_head -- python $R/bench.py --only-headline > $R/gpurun_out/r03f_bench_headline_profiled.json 2> $R/gpurun_out/r03f_bench_headline.err
python $R/tools/kernel_stats_csv.py "$(find /tmp/pr_head -name '*kernel_stats.csv' | head -1)" > $R/gpurun_out/r03f_kernel_stats_headline.txt
head -14 $R/gpurun_out/r03f_kernel_stats_headline.txt | cut -c1-150
