#!/bin/bash
# round 6, GPU session 6: full suite + the default bench line (as the driver runs it) + hipGraph replay one frame at a time
cd "$(dirname "$0")/../.."
O=gpurun_out/r06f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/gputests.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_like.json 2> $O/bench_driver_like.err; python -c "
import json; r=json.loads(open('$O/bench_driver_like.json').readline()); print(r['value'], r['ms_per_step']); print(json.dumps(r['summary']))"
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; r=json.loads(open('$O/bench_default.json').readline()); print(r['value'], r['ms_per_step']); print(json.dumps(r['summary']))"
for g in 0 1; do echo "graph=$g"; timeout 300 python bench.py --only-headline --no-configs --inflight 1 --graph $g --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print(r['value'], r['ms_per_step'])"; done 2>&1 | tee $O/single_graph.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/smoke.txt
