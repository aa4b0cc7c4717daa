#!/bin/bash
# Kernel sequence of one camera-branch frame (debug aid): rocprofv3 --kernel-trace of a 2-agent camera-only bench, names in launch order
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/bench.py --modalities cam --agents 2 --steps 1 --warmup 1 --only-headline --no-roofline --inflight 1 > /dev/null 2>&1
rm -rf /tmp/pr_tr
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/pr_tr -- python $R/bench.py --modalities cam --agents 2 --steps 1 --warmup 0 --only-headline --no-roofline --inflight 1 > /dev/null 2>&1
python - <<PY
import csv, glob, re
f = glob.glob("/tmp/pr_tr/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"])[:48] for r in rows]
# the last 400 launches = the timed frame's tail
out = names[-420:]
prev, cnt = None, 0
for n in out + [None]:
    if n == prev:
        cnt += 1
        continue
    if prev is not None:
        print(f"{cnt:3d} x {prev}")
    prev, cnt = n, 1
mc = glob.glob("/tmp/pr_tr/*/*_memory_copy_trace.csv")
if mc:
    rr = list(csv.DictReader(open(mc[0])))
    print("memory copies:", len(rr))
    from collections import Counter
    print(Counter((r.get("Direction"), r.get("Bytes", r.get("Size"))) for r in rr).most_common(12))
PY
