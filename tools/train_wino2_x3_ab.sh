cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for m in where2com v2vnet when2com; do for x in 0 1; do
  AV2X_TRAIN_WINO2_X3=$x python tools/train_bench.py --model $m --steps 8 --warmup 3 2>/tmp/e.txt > /tmp/o.json || tail -3 /tmp/e.txt
  python -c "import json; d=json.load(open('/tmp/o.json')); print('$m x3=$x', {k: v for k, v in d.items() if k.startswith('ms_')})"
done; done
AV2X_TRAIN_WINO2_X3=1 python -m pytest tests -m gpu -x -q -k "train" 2>&1 | tail -5
