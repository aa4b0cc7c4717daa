#!/bin/bash
# Which layers run as Winograd F(4x4,3x3): the per-IMAGE workgroup threshold of Engine.wino4_rule swept on the headline frame
# (256 = the two shrink convolutions only; 36 adds the 128 -> 128 layers at 50 x 176; 20 adds the 256 -> 256 layers at 25 x 88).
for t in 256 36 20; do
  for a in "" "--agents 8" "--agents 1"; do
    AV2X_WINO4_MIN_WGS=$t python bench.py $a --cpu-frames 0 --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('min_wgs $t', '$a', 'pipelined', d['value'], 'sequential', d['single_stream'].get('value') if isinstance(d.get('single_stream'),dict) else d.get('single_stream'), 'err', d.get('parity_max_abs_err_vs_oracle'))"
  done
done
