#!/bin/bash
# HBM-side bytes of av2x_linear_bf16 per output width (tools/lin16_bench.py shapes): two counters-only passes (FETCH_SIZE, WRITE_SIZE)
# -> gpurun_out/r03b_pmc_linear_bf16.json {"256->N": {"fetch_kib":..., "write_kib":..., "bytes": 2*FETCH*1024 + WRITE*1024}}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pl_$c
    timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pl_$c --output-format csv -- python $R/tools/lin16_bench.py > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, json, collections, os
out = {}
acc = {c: collections.defaultdict(list) for c in ("FETCH_SIZE", "WRITE_SIZE")}
for c in acc:
    f = glob.glob(f"/tmp/pl_{c}/**/*counter_collection.csv", recursive=True)
    for r in csv.DictReader(open(f[0])):
        if "linear_bf16" not in r["Kernel_Name"] or r["Counter_Name"] != c:
            continue
        acc[c][(r["Kernel_Name"][:40], r["Dispatch_Id"])].append(float(r["Counter_Value"]))
# dispatches come in the bench's order: per shape 3 warm-up + 20 timed launches = 23 consecutive dispatches
names = ["256->1280", "256->2304", "256->256", "256->256 gelu", "256->256 fp32+res"]
for c in acc:
    vals = [sum(v) for k, v in sorted(acc[c].items(), key=lambda kv: int(kv[0][1]))]
    for i, nm in enumerate(names):
        chunk = vals[23 * i: 23 * (i + 1)]
        if chunk:
            out.setdefault(nm, {})[c] = sum(chunk) / len(chunk)
for nm, d in out.items():
    d["bytes"] = 2 * d.get("FETCH_SIZE", 0) * 1024 + d.get("WRITE_SIZE", 0) * 1024
json.dump({"tokens": 281600, "note": "bytes per launch = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB (gfx950 calibration of profiles/r03_pmc_hbm.json), "
           "tools/lin16_bench.py shapes at 281 600 tokens", "per_shape": out}, open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03b_pmc_linear_bf16.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
