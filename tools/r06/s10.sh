#!/bin/bash
# round 6, GPU session 10: which steps carry the B-fragment loads (timing only: results do not depend on it)
cd "$(dirname "$0")/../.."
O=gpurun_out/r06j; mkdir -p $O
for m in 0x1CE 0xAAA 0x555 0x565 0xCAA 0x2DA 0x6B4; do BMASK=$m SUFFIX=_m$m BULK=0 tools/micro/w4x3_ablate.sh build 0 > $O/build_$m.log 2>&1; done
for rep in 1 2; do
for m in 0x1CE 0xAAA 0x555 0x565 0xCAA 0x2DA 0x6B4; do
  for a in "4 25 88 256" "4 50 176 128" "4 100 352 256"; do echo -n "mask $m: "; timeout 120 ./tools/micro/w4x3_ablate_0_m$m $a | grep "pp=0"; done
done; done 2>&1 | tee $O/bmask.txt
