#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV of a PIPELINED bench run (frames in flight: kernels of different frames overlap, per-launch durations are not
chip-exclusive time) -> per kernel and frame: launches, summed duration, and CHIP TIME = sum over launches of duration x min(1, workgroups / 256
CUs): what a launch takes away from the other frames' kernels.  The steady-state part of the trace (its last 60 %) is used; a frame is counted
by the kernel that runs once per frame (the sparse first convolution).  Usage: python tools/chip_time.py <kernel_trace.csv[.gz]>"""
import collections
import csv
import gzip
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)[:48]


def main(path, marker="conv3x3s2_sparse_kernel"):
    rows = list(csv.DictReader(gzip.open(path, "rt") if path.endswith(".gz") else open(path)))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    t1 = max(int(r["End_Timestamp"]) for r in rows)
    cut = t0 + 0.4 * (t1 - t0)
    acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
    frames = 0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s < cut:
            continue
        g = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        w = max(1, int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1))
        k = short(r["Kernel_Name"])
        a = acc[k]
        a[0] += 1
        a[1] += (e - s) / 1e3
        a[2] += (e - s) / 1e3 * min(1.0, (g // w) / 256.0)
        frames += marker in k
    span = (t1 - cut) / 1e3
    print(f"# {path.split('/')[-1]}: steady-state window {span / 1e3:.1f} ms, {frames} frames -> {span / max(1, frames):.1f} us per frame (under the profiler)")
    print(f"{'kernel':50s} {'launches/frame':>14s} {'sum of durations us/frame':>26s} {'chip time us/frame':>19s}")
    tot = [0.0, 0.0]
    for k, (c, d, ct) in sorted(acc.items(), key=lambda kv: -kv[1][2]):
        tot[0] += d / max(1, frames)
        tot[1] += ct / max(1, frames)
        if ct / max(1, frames) >= 3.0:
            print(f"{k:50s} {c / max(1, frames):14.1f} {d / max(1, frames):26.1f} {ct / max(1, frames):19.1f}")
    print(f"{'TOTAL (all kernels)':50s} {'':14s} {tot[0]:26.1f} {tot[1]:19.1f}")


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]))
