#!/bin/bash
# HBM-side traffic of one training step's convolution kernels (tools/train_bench.py --model M [--amp]): two counters-only rocprofv3 passes
# (FETCH_SIZE, WRITE_SIZE; --kernel-trace only), bytes per launch = 2 x FETCH + WRITE KiB (the guide's gfx950 correction), grouped the way
# train_bench's roofline groups its launches (wino = conv_wino*, wgrad = conv_*wgrad*, igemm = the other conv_igemm* / conv_fixup kernels).
# Usage: tools/pmc_train_r04.sh <model> [--amp]   ->  gpurun_out/r04_pmc_train_<model>[_amp].json (copy to profiles/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
M=${1:-where2com}; AMP=$2
TAG=$M$([ -n "$AMP" ] && echo _amp)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pt_$c
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d /tmp/pt_$c --output-format csv -- python $R/tools/train_bench.py --model $M $AMP --steps 3 --warmup 2 > /dev/null 2>/tmp/pt_$c.err || tail -3 /tmp/pt_$c.err
done
TAG=$TAG OUT=$R/gpurun_out/r04_pmc_train_$TAG.json python - <<'PY'
import csv, glob, json, os, collections
acc = {"FETCH_SIZE": collections.defaultdict(list), "WRITE_SIZE": collections.defaultdict(list)}
def group(n):
    if "wgrad" in n: return "wgrad"
    if "conv_wino" in n: return "wino"
    if "conv_igemm" in n or "conv_fixup" in n or "conv_halo" in n: return "igemm"
    return None
for c in acc:
    for f in glob.glob(f"/tmp/pt_{c}/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and group(r["Kernel_Name"]):
                per[(group(r["Kernel_Name"]), r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (g, d), v in per.items():
            acc[c][g].append(v)
out = {"command": f"tools/train_bench.py --model {os.environ['TAG']}", "unit": "bytes", "formula": "(2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 per launch", "per_group": {}}
for g in set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"]):
    f, w = acc["FETCH_SIZE"].get(g, []), acc["WRITE_SIZE"].get(g, [])
    fm, wm = (sum(f) / len(f) if f else 0.0), (sum(w) / len(w) if w else 0.0)
    out["per_group"][g] = {"launches": len(f), "fetch_kib_per_launch": round(fm, 1), "write_kib_per_launch": round(wm, 1), "bytes_per_launch": (2 * fm + wm) * 1024}
json.dump(out, open(os.environ["OUT"], "w"), indent=1)
print(json.dumps(out))
PY
