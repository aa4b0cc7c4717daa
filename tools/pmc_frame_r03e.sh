#!/bin/bash
# HBM-side bytes of the kernels of one V2X-ViT autocast frame (8 agents), per kernel family: two counters-only rocprofv3 passes
# (FETCH_SIZE, WRITE_SIZE; never combined with other trace domains) over the bench command itself.
#   bytes = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB   (gfx950 calibration: profiles/r03_pmc_hbm.json)
# -> gpurun_out/r03e_pmc_v2xvit_amp_frame.json {family: {launches, bytes_per_launch, fetch_kib_per_launch, write_kib_per_launch}}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--model v2xvit --agents 8 --amp --cpu-frames 0 --steps 4 --warmup 1 --inflight 1 --only-headline --no-roofline"
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pf_$c
    timeout 900 rocprofv3 --kernel-trace --pmc $c -d /tmp/pf_$c --output-format csv -- python $R/bench.py $ARGS > /tmp/pf_$c.json 2> /tmp/pf_$c.err
    tail -c 300 /tmp/pf_$c.err
done
python - <<'PY'
import csv, glob, json, collections, os
FAMILIES = {"ln_qkv_window_out_bf16_kernel": "ln_qkv_window_out_bf16", "linear_bf16_occ_kernel": "linear_bf16", "conv_halo_bf16": "conv_halo_bf16", "hgt_attention_bf16x8": "hgt_attention_bf16",
            "split_combine_kernel": "split_combine", "gap3_stage1": "gap3_stage1", "layernorm_bf16_kernel": "layernorm_bf16",
            "conv_igemm_bf16": "conv_igemm_bf16", "warp_affine_kernel": "warp_affine"}
tot = {c: collections.defaultdict(float) for c in ("FETCH_SIZE", "WRITE_SIZE")}
cnt = collections.defaultdict(set)
for c in tot:
    f = glob.glob(f"/tmp/pf_{c}/**/*counter_collection.csv", recursive=True)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c:
            continue
        fam = next((v for k, v in FAMILIES.items() if k in r["Kernel_Name"]), None)
        if fam is None:
            continue
        tot[c][fam] += float(r["Counter_Value"])
        if c == "FETCH_SIZE":
            cnt[fam].add(r["Dispatch_Id"])
out = {}
for fam, ids in cnt.items():
    n = len(ids)
    fk, wk = tot["FETCH_SIZE"][fam] / n, tot["WRITE_SIZE"][fam] / n
    out[fam] = {"launches": n, "fetch_kib_per_launch": round(fk, 1), "write_kib_per_launch": round(wk, 1),
                "bytes_per_launch": round(2 * fk * 1024 + wk * 1024)}
json.dump({"command": "bench.py --model v2xvit --agents 8 --amp --steps 4 --warmup 1 --inflight 1 --only-headline --no-roofline",
           "note": "averages over every launch of the family in the run (the launch mix of the frame); bytes = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB "
                   "(gfx950 calibration of profiles/r03_pmc_hbm.json)", "per_family": out},
          open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03e_pmc_v2xvit_amp_frame.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
