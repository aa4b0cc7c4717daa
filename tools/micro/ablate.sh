#!/bin/bash
# Test-only builds of the conv kernel with pieces of the main loop removed (AV2X_ABLATE bits: 1 global loads, 2 LDS stores,
# 4 barriers, 8 LDS fragment reads).  Output: tools/micro/libablate_<bits>.so (git-ignored), used by tools/loop_peak.py --ablate.
set -e
cd "$(dirname "$0")/../.."
for b in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -I include -I airv2x_perception_amd/csrc \
     -DAV2X_ABLATE=$b -shared -o tools/micro/libablate_$b.so airv2x_perception_amd/csrc/conv_igemm.hip airv2x_perception_amd/csrc/capi.hip &
done
wait
ls -la tools/micro/libablate_*.so
