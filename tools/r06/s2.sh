#!/bin/bash
# round 6, GPU session 2: what bounds the ping-pong kernel's phases (timing-only ablations)
cd "$(dirname "$0")/../.."
O=gpurun_out/r06b; mkdir -p $O
for ab in 1 2 4 6 8 9; do PPAB=$ab SUFFIX=_pp$ab BULK=0 tools/micro/w4x3_ablate.sh build 0 > $O/build_$ab.log 2>&1; done
BULK=0 tools/micro/w4x3_ablate.sh build 0 > $O/build_0.log 2>&1
for v in 0 0_pp1 0_pp2 0_pp4 0_pp6 0_pp8 0_pp9; do
  for a in "4 25 88 256" "4 50 176 128" "4 100 352 256"; do echo -n "variant $v: "; timeout 120 ./tools/micro/w4x3_ablate_$v $a | grep "pp=1"; done
done 2>&1 | tee $O/pp_ablations.txt
