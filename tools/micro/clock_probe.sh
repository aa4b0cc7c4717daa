#!/bin/bash
# Effective shader clock of the conv main loop on random vs zero-filled operands (DVFS probe):
# GRBM_GUI_ACTIVE (GPU-busy cycles) / kernel duration, from separate rocprofv3 passes (PMC, then kernel trace).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for z in 0 1; do
  rm -rf /tmp/cp_$z /tmp/ct_$z
  AV2X_ZERO_DATA=$z AV2X_PEAK_ONLY=4 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/cp_$z -- python $R/tools/loop_peak.py 128x128w8d > /dev/null 2>&1
  AV2X_ZERO_DATA=$z AV2X_PEAK_ONLY=4 rocprofv3 --kernel-trace --output-format csv -d /tmp/ct_$z -- python $R/tools/loop_peak.py 128x128w8d > /dev/null 2>&1
  python3 - <<PY
import csv, glob
def rows(pat):
    out=[]
    for f in glob.glob(pat, recursive=True):
        out += list(csv.DictReader(open(f)))
    return out
pmc = [r for r in rows("/tmp/cp_$z/**/*counter_collection.csv") if "conv_igemm" in r.get("Kernel_Name","")]
tr = [r for r in rows("/tmp/ct_$z/**/*kernel_trace.csv") if "conv_igemm" in r.get("Kernel_Name","")]
cyc = sorted(float(r["Counter_Value"]) for r in pmc if r["Counter_Name"]=="GRBM_GUI_ACTIVE")
dur = sorted((int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for r in tr)
if cyc and dur:
    c, d = cyc[len(cyc)//2], dur[len(dur)//2]
    print(f"zero_data=$z  launches pmc={len(cyc)} trace={len(dur)}  median GRBM_GUI_ACTIVE={c:.0f} cycles  median duration={d/1e3:.1f} us  -> effective clock {c/d:.3f} GHz")
else:
    print("zero_data=$z: no rows", len(pmc), len(tr))
PY
done
