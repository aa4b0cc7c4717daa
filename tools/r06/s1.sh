#!/bin/bash
# round 6, GPU session 1: validate the ping-pong F(4x4) kernel, time it against the four-wave form, batching experiment, full suite, default bench
cd "$(dirname "$0")/../.."
O=gpurun_out/r06a; mkdir -p $O
export TMPDIR=/tmp
echo "== pp kernel tests"; timeout 600 python -m pytest tests/test_gpu_wino4_x3.py -x -q 2>&1 | tail -15 | tee $O/test_wino4.txt
echo "== isolated launches (both forms)"
BULK=0 tools/micro/w4x3_ablate.sh build 0 > $O/build.log 2>&1
for a in "4 25 88 256" "3 25 88 256" "4 50 176 128" "3 50 176 128" "4 100 352 256" "1 100 352 256" "8 25 88 256" "8 50 176 128"; do timeout 120 ./tools/micro/w4x3_ablate_0 $a; done 2>&1 | tee $O/isolated.txt
echo "== headline A/B"
for i in 1 2; do for pp in 1 0; do echo "pp=$pp"; AV2X_W4X3_PP=$pp timeout 300 python bench.py --only-headline --no-configs --steps 60 --warmup 10 2>$O/ab.err | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print(r['value'], r['ms_per_step'])"; done; done 2>&1 | tee $O/headline_ab.txt
echo "== single-stream A/B"
for pp in 1 0; do echo "pp=$pp"; AV2X_W4X3_PP=$pp timeout 300 python bench.py --only-headline --no-configs --inflight 1 --steps 60 --warmup 10 2>>$O/ab.err | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print(r['value'], r['ms_per_step'])"; done 2>&1 | tee $O/single_ab.txt
echo "== batch vs inflight"; timeout 600 python tools/batch_bench.py 4 2>&1 | grep agents | tee $O/batch4.txt
echo "== full gpu suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/gputests.txt
echo "== default bench"; timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
