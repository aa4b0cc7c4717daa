cd $GRAFT_REPO_ROOT
bash tools/micro/w4x3_ablate.sh run 0 0s 32 31
timeout 600 python -m pytest tests/test_gpu_wino4_x3.py -x -q 2>&1 | tail -3
