#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06h; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/gputests.txt
