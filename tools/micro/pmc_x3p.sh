#!/bin/bash
# SQ counters of conv_igemm_x3p alone (tools/micro/x3p_run.py), one counters-only rocprofv3 pass per group; averages per launch.
# Usage: tools/micro/pmc_x3p.sh "<cin> <cout> <bn>" ...     (BM=256 in the environment selects the 256-row tile)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for V in "$@"; do
i=0
rm -rf /tmp/px_*
python $R/tools/micro/x3p_run.py $V 20 2>&1 | tail -1
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_DATA_FIFO_FULL" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 180 rocprofv3 --kernel-trace --pmc $grp -d /tmp/px_$i --output-format csv -- python $R/tools/micro/x3p_run.py $V 10 > /dev/null 2>/tmp/px_$i.err || tail -3 /tmp/px_$i.err
done
echo "== x3p $V (BM=${BM:-128})"
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
dur = []
for f in glob.glob("/tmp/px_*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "conv_igemm_x3p" in r["Kernel_Name"]:
            per[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (c, d), v in per.items():
        acc[c].append(v)
for f in glob.glob("/tmp/px_1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_igemm_x3p" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
if dur:
    print(f"# kernel duration under the profiler: {sum(dur)/len(dur):.1f} us avg over {len(dur)} launches")
for c, v in sorted(acc.items()):
    print(f"{c:28s} {sum(v) / len(v):16.0f}  per launch ({len(v)} launches)")
PY
done
