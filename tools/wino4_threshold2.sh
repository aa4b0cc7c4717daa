#!/bin/bash
# The same sweep (tools/wino4_threshold.sh) on the sequential headline frame and on the 8-agent frames of the other models.
for t in 256 36 20; do
  for a in "--inflight 1" "--modalities cam,lidar --agents 8 --steps 10 --warmup 2" "--model cobevt --agents 8" "--model v2xvit --agents 8"; do
    AV2X_WINO4_MIN_WGS=$t python bench.py $a --cpu-frames 0 --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('min_wgs $t', '$a', d['value'], 'frames/s', d['ms_per_step'], 'ms')"
  done
done
