#!/usr/bin/env python3
"""Ordered kernel timeline of ONE step from a `rocprofv3 --kernel-trace` run of bench.py: name, grid, duration and the idle gap in
front of each launch.  The step is cut at the optimizer kernel (adamw / sgd): the launches between the last two of them.
usage: python3 tools/prof_timeline.py <rocprof output dir> [out.md]"""
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN(?:12_GLOBAL__N_1|7iseg_mm)(\d+)", name)
    if m:
        n = int(m.group(1))
        s = name[m.end():m.end() + n]
        return s + name[m.end() + n:m.end() + n + 36]
    return name[:70]


def main():
    d = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    opt = [i for i, r in enumerate(rows) if re.search(r"adamw_kernel|sgd_kernel", r["Kernel_Name"])]
    if len(opt) < 2:
        raise SystemExit("fewer than two optimizer launches in the trace")
    a, b = opt[-2] + 1, opt[-1] + 1
    step = rows[a:b]
    t0 = int(rows[a - 1]["End_Timestamp"])
    print(f"{len(step)} launches, {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us wall, "
          f"{sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e3:.1f} us of kernels\n", file=out)
    print("| # | at us | gap us | us | kernel | grid | block |\n|---|---|---|---|---|---|---|", file=out)
    prev = t0
    for i, r in enumerate(step):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"| {i} | {(s - t0) / 1e3:.1f} | {(s - prev) / 1e3:.1f} | {(e - s) / 1e3:.1f} | `{short(r['Kernel_Name'])}` | "
              f"{r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} | {r['Workgroup_Size_X']} |", file=out)
        prev = max(prev, e)


if __name__ == "__main__":
    main()
