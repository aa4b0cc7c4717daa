#!/bin/bash
# SQ / TA / TCP counters of conv_wino_x3 alone (tools/wino_bench.py on one layer, only the split-3 tile), one counters-only
# rocprofv3 pass per group; averages per launch.  Usage: tools/pmc_x3.sh <layer> <tile> [kernel substring]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
LAYER=${1:-shrink3_n4}; TILE=${2:-x3_64x64}; KSUB=${3:-conv_wino_x3}
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1)); rm -rf /tmp/px_$i
    timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/px_$i --output-format csv -- python $R/tools/wino_bench.py --layers $LAYER --tiles $TILE --iters 5 --only-x3 > /dev/null 2>/tmp/px_$i.err || tail -3 /tmp/px_$i.err
done
KSUB=$KSUB python - <<'PY'
import csv, glob, collections, os
acc = collections.defaultdict(list)
dur = []
for f in glob.glob("/tmp/px_*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if os.environ["KSUB"] in r["Kernel_Name"]:
            per[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (c, d), v in per.items():
        acc[c].append(v)
for f in glob.glob("/tmp/px_1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if os.environ["KSUB"] in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
if dur:
    print(f"# kernel duration under the profiler: {sum(dur)/len(dur):.1f} us avg over {len(dur)} launches")
for c, v in sorted(acc.items()):
    print(f"{c:28s} {sum(v) / len(v):16.0f}  per launch ({len(v)} launches)")
PY
