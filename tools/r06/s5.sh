#!/bin/bash
# round 6, GPU session 5: evidence -- SQ tables of both F(4x4) forms, rocprofv3 kernel stats of the default-mode headline (pipelined and one frame at a time)
cd "$(dirname "$0")/../.."
O=$PWD/gpurun_out/r06e; mkdir -p $O
BULK=0 tools/micro/w4x3_ablate.sh build 0 > $O/build.log 2>&1
tools/r06/pmc_w4.sh "4 25 88 256" > $O/pmc_sq_w4_80wgs.txt 2>&1
tools/r06/pmc_w4.sh "4 100 352 256" > $O/pmc_sq_w4_1100wgs.txt 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/bench.py --only-headline --steps 3 --warmup 2 > /dev/null 2>&1
rm -rf /tmp/p1 /tmp/p3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $R/bench.py --only-headline --inflight 1 > $O/bench_inflight1_profiled.json 2> /dev/null
python $R/tools/kernel_stats_csv.py "$(find /tmp/p1 -name '*kernel_stats.csv' | head -1)" > $O/kernel_stats_inflight1.txt
python $R/tools/frame_timeline.py "$(find /tmp/p1 -name '*kernel_trace.csv' | head -1)" > $O/timeline_inflight1.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python $R/bench.py --only-headline > $O/bench_headline_profiled.json 2> /dev/null
python $R/tools/kernel_stats_csv.py "$(find /tmp/p3 -name '*kernel_stats.csv' | head -1)" > $O/kernel_stats_headline.txt
cp "$(find /tmp/p3 -name '*kernel_trace.csv' | head -1)" $O/kernel_trace_headline.csv 2>/dev/null; gzip -f $O/kernel_trace_headline.csv
head -12 $O/kernel_stats_headline.txt | cut -c1-150
head -30 $O/pmc_sq_w4_80wgs.txt
