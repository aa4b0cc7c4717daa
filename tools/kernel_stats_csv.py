#!/usr/bin/env python3
"""rocprofv3 `--kernel-trace --stats --output-format csv`: *_kernel_stats.csv -> the per-kernel table kept under profiles/
(name, calls, total us, average us, percent).  Usage: python tools/kernel_stats_csv.py <kernel_stats.csv> > profiles/rNN_….txt"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\((HIP_vector_type|float|int|ConvParams|AgentPtrs|at::|hipcub|rocprim|unsigned|char|long|WinoParams|Lift|Lss|Depth).*$", "", name)
    return name[:110]


def main(path):
    rows = list(csv.DictReader(open(path)))
    print(f"# rocprofv3 --kernel-trace --stats --output-format csv   source: {path.split('/')[-1]}")
    print("# durations in microseconds (rocprofv3 reports nanoseconds)")
    print(f"{'kernel':<112} {'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>6}")
    for r in rows:
        print(f"{short(r['Name']):<112} {int(r['Calls']):>6} {float(r['TotalDurationNs']) / 1e3:>12.1f} {float(r['AverageNs']) / 1e3:>10.2f} {float(r['Percentage']):>6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
