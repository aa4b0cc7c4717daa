#!/bin/bash
# round 6, GPU session 3: the new functional tests
cd "$(dirname "$0")/../.."
O=gpurun_out/r06c; mkdir -p $O
timeout 1200 python -m pytest tests/test_w2c_variants.py tests/test_gpu_sharded.py tests/test_gpu_soak.py tests/test_camera.py tests/test_native_shims.py "tests/test_gpu_forward.py" -m gpu -q 2>&1 | tail -40 | tee $O/tests.txt
