#!/bin/bash
# Round-3 (final code) HBM-side traffic for the bench lines whose roofline.traffic was still null: the 8-agent Where2Comm frame
# (conv_wino_f32_h at 504 / 572 / 964 ... workgroups), V2X-ViT fp32 at 8 agents (g128x128w8), the camera + LiDAR frame (conv_wino4_f32)
# and the headline's own mix again.  Two rocprofv3 PMC passes per mode (FETCH_SIZE, WRITE_SIZE; counters only, with --kernel-trace),
# merged INTO profiles/pmc_hbm.json (tools/pmc_traffic.py --merge keeps the earlier entries and the r03 calibration).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
F=""; W=""
pass() {   # name, bench args
    local name=$1; shift
    python $R/bench.py "$@" --steps 2 --warmup 1 --only-headline --no-roofline > /dev/null 2>&1
    for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmc_${name}_$c
        timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${name}_$c -- python $R/bench.py "$@" --steps 3 --warmup 1 --only-headline --no-roofline > /dev/null 2>&1
    done
    F="$F /tmp/pmc_${name}_FETCH_SIZE"; W="$W /tmp/pmc_${name}_WRITE_SIZE"
}
pass head3 --inflight 3
pass agents8 --agents 8 --inflight 3
pass agents8s --agents 8 --inflight 1
pass v2xvit8 --model v2xvit --agents 8 --inflight 1
pass cobevt8 --model cobevt --agents 8 --inflight 1
pass cam8 --modalities cam,lidar --agents 8 --inflight 1
pass cam8p --modalities cam,lidar --agents 8
cd $R && python tools/pmc_traffic.py --fetch $F --write $W --merge profiles/pmc_hbm.json -o gpurun_out/r03f_pmc_hbm.json
