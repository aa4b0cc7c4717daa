#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06g; mkdir -p $O
timeout 1200 python -m pytest tests/test_w2c_variants.py tests/test_gpu_train.py -m gpu -q -x 2>&1 | tail -25 | tee $O/tests.txt
