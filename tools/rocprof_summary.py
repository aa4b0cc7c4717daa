#!/usr/bin/env python3
"""Condense a rocprofv3 rocpd sqlite database (--kernel-trace --stats) into the per-kernel
table the judge reads (name, calls, total us, average us, percent).  Usage:
    python tools/rocprof_summary.py gpurun_out/prof/.../NNN_results.db > profiles/rNN_kernel_stats.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\((HIP_vector_type|float|int|ConvParams|AgentPtrs|at::|hipcub|rocprim|unsigned|char).*$", "", name)
    return name[:110]


def main(path):
    c = sqlite3.connect(path)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print(f"# rocprofv3 --kernel-trace --stats   source: {path}")
    print("# durations in microseconds (as stored by rocprofv3's top_kernels view)")
    print(f"{'kernel':<112} {'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>6}")
    for n, calls, tot, avg, pct in rows:
        print(f"{short(n):<112} {calls:>6} {tot:>12.1f} {avg:>10.2f} {pct:>6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
