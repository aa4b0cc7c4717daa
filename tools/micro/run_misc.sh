cd $GRAFT_REPO_ROOT
python tools/v2xvit_margin.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_v2xvit_margin.txt; cat gpurun_out/r04_v2xvit_margin.txt
bash tools/pmc_r04.sh 2>&1 | tail -12
python bench.py --cpu-frames 0 > gpurun_out/r04d_default_nocpu.json 2> gpurun_out/r04d_default_nocpu.err || tail -5 gpurun_out/r04d_default_nocpu.err
python -c "
import json; d=json.load(open('gpurun_out/r04d_default_nocpu.json')); r=d['roofline']; print(d['value'], r['frac'], r['traffic'], r['traffic_over_algorithmic']); print(json.dumps(r.get('hbm_bound_kernels'))[:1500])"
