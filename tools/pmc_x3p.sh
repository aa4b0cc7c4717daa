#!/bin/bash
# SQ counters of conv_igemm_x3p alone (tools/split3_bench.py on the 8 x 100 x 352 x 1024 -> 256 Linear shape), counters-only passes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TILE=${1:-x3p 128x128}; KSUB=${2:-conv_igemm_x3p}
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1)); rm -rf /tmp/pp_$i
    AV2X_S3_ONLY="8,100,352,1024,256" timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pp_$i --output-format csv -- python $R/tools/split3_bench.py "$TILE" > /dev/null 2>/tmp/pp_$i.err || tail -3 /tmp/pp_$i.err
done
KSUB=$KSUB python - <<'PY'
import csv, glob, collections, os
acc = collections.defaultdict(list)
dur = []
for f in glob.glob("/tmp/pp_*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if os.environ["KSUB"] in r["Kernel_Name"]:
            per[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (c, d), v in per.items():
        acc[c].append(v)
for f in glob.glob("/tmp/pp_1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if os.environ["KSUB"] in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
if dur:
    print(f"# kernel duration under the profiler: {sum(dur)/len(dur):.1f} us avg over {len(dur)} launches")
for c, v in sorted(acc.items()):
    print(f"{c:28s} {sum(v) / len(v):16.0f}  per launch ({len(v)} launches)")
PY
