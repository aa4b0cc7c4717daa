#!/bin/bash
# SQ counters of ln_qkv_window_out_slab_kernel (AV2X_QW_SLAB=0: ln_qkv_window_out_bf16_kernel) alone (tools/qw_bench.py), one counters-only pass per group.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
    i=$((i+1)); rm -rf /tmp/pq_$i
    timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pq_$i --output-format csv -- python $R/tools/qw_bench.py > /dev/null 2>/tmp/pq_$i.err || tail -3 /tmp/pq_$i.err
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pq_*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "ln_qkv_window" in r["Kernel_Name"]:
            per[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (c, d), v in per.items():
        acc[c].append(v)
for c, v in sorted(acc.items()):
    print(f"{c:28s} {sum(v) / len(v):16.0f}  per launch ({len(v)} launches)")
PY
