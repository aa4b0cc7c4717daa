cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train_when2com.py -x -q -s -k "full" 2>&1 | grep -v amdgpu.ids | tail -5
for m in when2com v2vnet; do python tools/train_bench.py --model $m --agents 4 --steps 6 --warmup 2 2>/dev/null | tail -1 > gpurun_out/r04_train_$m.json; python -c "
import json; d=json.load(open('gpurun_out/r04_train_$m.json')); print('$m', d['ms_per_step'], d['ms_forward'], d['ms_loss_backward'], d['ms_optimizer'], d['peak_mem_gib'], d['loss_first_last'])"; done
bash tools/prof_cam_r04.sh
