#!/bin/bash
# round 6, GPU session 4: issue-rate probe + the functional GPU tests added since session 3
cd "$(dirname "$0")/../.."
O=gpurun_out/r06d; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_probe_bin tools/micro/issue_probe.hip > $O/build.log 2>&1
timeout 300 /tmp/issue_probe_bin | tee $O/issue_probe.txt
timeout 1200 python -m pytest tests/test_w2c_variants.py tests/test_camera.py tests/test_gpu_sharded.py -m gpu -q 2>&1 | tail -15 | tee $O/tests.txt
