run() { name=$1; shift; python bench.py "$@" --cpu-frames 0 2>/dev/null > gpurun_out/r03e_$name.json; python -c "import json; d=json.load(open('gpurun_out/r03e_$name.json')); r=d.get('roofline',{}); print('$name', d['value'], d['ms_per_step'], r.get('bound'), r.get('frac'), (r.get('kernel') or '')[:40])"; }
run cam_lidar_n8 --modalities cam,lidar --agents 8 --steps 10 --warmup 2
run v2xvit_n8 --model v2xvit --agents 8
run v2xvit_n8_amp --model v2xvit --agents 8 --amp
run cobevt_n8 --model cobevt --agents 8
run agents8 --agents 8
run when2com_n4 --model when2com
run v2vnet_n4 --model v2vnet
python tools/train_bench.py --steps 6 --warmup 2 > gpurun_out/r03e_train_step.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r03e_train_step.json')); print('train', d['ms_per_step'], d['steps_per_s'])"
