#!/bin/bash
# Round 5: rocprofv3 kernel summaries + the launch-by-launch timeline of one frame (single stream) -> gpurun_out/r05${TAG}_*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
TAG=${TAG:-a}
ARGS=${ARGS:-}
python $R/bench.py --only-headline --steps 3 --warmup 2 $ARGS > /dev/null 2>&1
rm -rf /tmp/p1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $R/bench.py --only-headline --inflight 1 $ARGS > $O/r05${TAG}_bench_inflight1_profiled.json 2> /dev/null
python $R/tools/kernel_stats_csv.py "$(find /tmp/p1 -name '*kernel_stats.csv' | head -1)" > $O/r05${TAG}_kernel_stats_inflight1.txt
python $R/tools/frame_timeline.py "$(find /tmp/p1 -name '*kernel_trace.csv' | head -1)" > $O/r05${TAG}_timeline_inflight1.txt
rm -rf /tmp/p3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python $R/bench.py --only-headline $ARGS > $O/r05${TAG}_bench_headline_profiled.json 2> /dev/null
python $R/tools/kernel_stats_csv.py "$(find /tmp/p3 -name '*kernel_stats.csv' | head -1)" > $O/r05${TAG}_kernel_stats_headline.txt
head -14 $O/r05${TAG}_kernel_stats_headline.txt | cut -c1-160
tail -3 $O/r05${TAG}_timeline_inflight1.txt
