#!/bin/bash
# The training-step bench of every model that trains -> gpurun_out/${PFX:-r04l}_train_*.json (tools/train_bench.py); PFX = the round's tag.
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out
run() { name=$1; shift; python tools/train_bench.py "$@" 2>gpurun_out/${PFX:-r04l}_train_$name.err > gpurun_out/${PFX:-r04l}_train_$name.json || tail -3 gpurun_out/${PFX:-r04l}_train_$name.err
  python -c "import json; d=json.load(open('gpurun_out/${PFX:-r04l}_train_$name.json')); print('$name', {k: v for k, v in d.items() if k.startswith('ms_') or k == 'peak_mem_gib'})"; }
run where2com --steps 10 --warmup 3
run where2com_cam_lidar --modalities cam,lidar --steps 6 --warmup 2
run cobevt --model cobevt --steps 6 --warmup 2
run v2xvit --model v2xvit --steps 6 --warmup 2
run when2com --model when2com --steps 8 --warmup 2
run v2vnet --model v2vnet --steps 6 --warmup 2
