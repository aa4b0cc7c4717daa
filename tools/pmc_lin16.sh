#!/bin/bash
# PMC counters of av2x_linear_bf16 on the V2X-ViT shapes (tools/lin16_bench.py), one counter group per pass (counters only).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_lin16
mkdir -p $O
i=0
for grp in "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_EA0_WRREQ_STALL_sum"; do
    i=$((i+1))
    rm -rf /tmp/pl_$i
    timeout 600 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pl_$i --output-format csv -- python $R/tools/lin16_bench.py > /dev/null 2>$O/err_$i.txt
    f=$(find /tmp/pl_$i -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "linear_bf16" not in k: continue
    acc[(k[:60], r.get("Grid_Size"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
done 2>&1 | tee $O/summary.txt
