#!/bin/bash
# build (here, no GPU needed): tools/micro/w4x3_ablate.sh build 0 1 2 ...   run (GPU box): tools/micro/w4x3_ablate.sh run 0 1 2 ...
cd "$(dirname "$0")/../.."
mode=$1; shift
if [ "$mode" = build ]; then
  for b in "$@"; do
    /opt/rocm/bin/hipcc -DAV2X_W4X3_BULK=${BULK:-1} --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -fno-slp-vectorize -mllvm -disable-vector-combine -I include -I airv2x_perception_amd/csrc -DAV2X_W4X3_ABLATE=$b ${PPAB:+-DAV2X_W4PP_ABLATE=$PPAB} ${BMASK:+-DAV2X_W4X3_BMASK=$BMASK} \
       -o tools/micro/w4x3_ablate_${b}${SUFFIX} tools/micro/w4x3_ablate.hip airv2x_perception_amd/csrc/capi.hip &
  done
  wait
else
  for b in "$@"; do ./tools/micro/w4x3_ablate_$b 4; ./tools/micro/w4x3_ablate_$b 1; done
fi
