"""kernel shares of one rocprofv3 --kernel-trace run of tools/bench_configs.py (markdown section for profiles/)
usage: python3 tools/prof_config.py <rocprof output dir> <title> <steps in the trace> [top]"""
import collections
import csv
import glob
import re
import sys

d = collections.defaultdict(list)
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
steps = int(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 12
tot = sum(sum(v) for v in d.values())
n = sum(len(v) for v in d.values())
print(f"## {sys.argv[2]} -- {tot / steps / 1e3:.1f} ms of kernels per step, {n // steps} launches per step")
print("| share | calls/step | avg us | kernel |")
print("|---|---|---|---|")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:top]:
    name = re.sub(r"\(anonymous namespace\)::", "", k)[:110]
    print(f"| {100 * sum(v) / tot:.1f} % | {len(v) // steps} | {sum(v) / len(v):.1f} | `{name}` |")
print()
