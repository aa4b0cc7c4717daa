#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV (*_kernel_trace.csv) of a single-stream bench run -> the launch-by-launch timeline of ONE steady-state
frame: start offset, duration, gap to the previous kernel's end, workgroups, kernel.  The frame is delimited by the first kernel of the
frame period, marked by a kernel that runs once per frame (the sparse first convolution; count_nonzero_kernel in traces made before round 5).  Usage: python tools/frame_timeline.py <kernel_trace.csv> [--marker name] [--frame k]"""
import argparse
import csv
import re


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--marker", default=None, help="a kernel that runs once per frame (default: conv3x3s2_sparse_kernel, else count_nonzero_kernel)")
    ap.add_argument("--frame", type=int, default=-3, help="which frame (index into the marker occurrences; negative = from the end)")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for mk in ([a.marker] if a.marker else ["conv3x3s2_sparse_kernel", "count_nonzero_kernel"]):
        marks = [i for i, r in enumerate(rows) if mk in r["Kernel_Name"]]
        if marks:
            a.marker = mk
            break
    i0 = marks[a.frame]
    i1 = marks[a.frame + 1] if a.frame + 1 < 0 or a.frame + 1 < len(marks) else len(rows)
    t0 = int(rows[i0]["Start_Timestamp"])
    prev_end = None
    busy = gaps = 0
    print(f"# frame = launches {i0}..{i1 - 1} of {a.path.split('/')[-1]}  (marker {a.marker})")
    print(f"{'start_us':>9} {'dur_us':>8} {'gap_us':>7} {'wgs':>6}  kernel")
    for r in rows[i0:i1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        wg = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1) // max(1, int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1))
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f} {wg:6d}  {short(r['Kernel_Name'])}")
        busy += e - s
        gaps += max(0, s - prev_end) if prev_end is not None else 0
        prev_end = max(prev_end or e, e)
    print(f"# frame wall {(prev_end - t0) / 1e3:.1f} us: kernels {busy / 1e3:.1f} us, gaps {gaps / 1e3:.1f} us over {i1 - i0} launches")


if __name__ == "__main__":
    main()
