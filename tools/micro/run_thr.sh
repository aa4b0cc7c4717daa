cd $GRAFT_REPO_ROOT
for t in 256 32 16; do
  AV2X_WINO4_MIN_WGS=$t python bench.py --cpu-frames 0 --no-train 2>/dev/null > gpurun_out/r04_thr_$t.json
  python -c "
import json; d=json.load(open('gpurun_out/r04_thr_$t.json')); print('thr $t', 'pipelined', d['value'], 'single', d['single_stream']['frames_per_s'], 'from_points', d['from_points']['pipelined']['frames_per_s'], 'parity', d.get('parity_max_abs_err_vs_oracle'))"
done
for t in 256 16; do
  AV2X_WINO4_MIN_WGS=$t python bench.py --cpu-frames 0 --agents 8 --only-headline 2>/dev/null > gpurun_out/r04_thr8_$t.json
  python -c "
import json; d=json.load(open('gpurun_out/r04_thr8_$t.json')); print('thr $t agents8', d['value'])"
done
