#!/bin/bash
# Timing-only builds of ln_qkv_window_out_bf16_kernel with pieces removed (AV2X_QW_ABLATE bits: 1 no weight stream after the first
# fragments, 2 no attention phases, 4 no to_out stores).  Output: tools/micro/libqw_ablate_<bits>.so (git-ignored), run through
# AV2X_QW_LIB=... python tools/qw_bench.py.  The product library never defines the macro.
set -e
cd "$(dirname "$0")/../.."
for b in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -I include -I airv2x_perception_amd/csrc \
     -DAV2X_QW_ABLATE=$b -shared -o tools/micro/libqw_ablate_$b.so airv2x_perception_amd/csrc/linear_bf16.hip airv2x_perception_amd/csrc/capi.hip
done
ls -la tools/micro/libqw_ablate_*.so
