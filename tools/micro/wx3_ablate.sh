#!/bin/bash
# build (here, no GPU needed): tools/micro/wx3_ablate.sh build 0 1 2 ...   run (GPU box): tools/micro/wx3_ablate.sh run 0 1 2 ...
cd "$(dirname "$0")/../.."
mode=$1; shift
if [ "$mode" = build ]; then
  for b in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -fno-slp-vectorize -mllvm -disable-vector-combine -I include -I airv2x_perception_amd/csrc -DAV2X_WX3_ABLATE=$b $EXTRA \
       -o tools/micro/wx3_ablate_${b}${SUFFIX} tools/micro/wx3_ablate.hip airv2x_perception_amd/csrc/capi.hip &
  done
  wait
else
  for b in "$@"; do
    for t in "64 64" "32 128"; do ./tools/micro/wx3_ablate_$b $t 4 25 88 256; ./tools/micro/wx3_ablate_$b $t 4 25 88 512; done
  done
fi
