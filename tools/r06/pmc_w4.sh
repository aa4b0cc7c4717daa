#!/bin/bash
# SQ / TCC counters of the two forms of the split-3 F(4x4,3x3) kernel (four waves: conv_wino4_x3; eight waves ping-pong: conv_wino4_x3_pp) on one
# launch shape, standalone harness, one counters-only rocprofv3 pass per group, averages per launch.  Usage: tools/r06/pmc_w4.sh "<n h w cin>"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
SHAPE=${1:-"4 25 88 256"}
i=0
rm -rf /tmp/pw_*
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 120 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pw_$i --output-format csv -- $R/tools/micro/w4x3_ablate_0 $SHAPE > /dev/null 2>/tmp/pw_$i.err || tail -3 /tmp/pw_$i.err
done
python - "$SHAPE" <<'PY'
import csv, glob, collections, sys
for kern in ("conv_wino4_x3<", "conv_wino4_x3_pp<"):
    acc = collections.defaultdict(list)
    dur = []
    for f in glob.glob("/tmp/pw_*/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                per[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (c, d), v in per.items():
            acc[c].append(v)
    for f in glob.glob("/tmp/pw_1/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"== {kern}...>  shape (n h w cin) {sys.argv[1]}")
    if dur:
        print(f"# kernel duration under the profiler: {sum(dur)/len(dur):.1f} us avg over {len(dur)} launches")
    for c, v in sorted(acc.items()):
        print(f"{c:28s} {sum(v) / len(v):16.0f}  per launch ({len(v)} launches)")
PY
