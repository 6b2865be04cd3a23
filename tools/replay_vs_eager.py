"""Where a replayed step spends more (or less) kernel time than the eager step of the same configuration: per-kernel sums of the LAST <steps>
steps of two rocprofv3 --kernel-trace runs of tools/bench_configs.py (one eager, one --graph-step).
usage: python3 tools/replay_vs_eager.py <eager dir> <replay dir> <steps> [rows]"""
import collections
import csv
import glob
import re
import sys


def last_steps(path, steps):
    f = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
    rows.sort()
    # a step ends with the last optimizer launch: find the period from the tail (the sequence of names repeats exactly)
    names = [r[2] for r in rows]
    n = len(names)
    period = None
    for p in range(50, n // (steps + 1) + 1):
        if names[n - p:] == names[n - 2 * p:n - p] and names[n - 2 * p:n - p] == names[n - 3 * p:n - 2 * p]:
            period = p
            break
    if period is None:
        raise SystemExit(f"{path}: no repeating tail")
    tail = rows[n - steps * period:]
    wall = (tail[-1][1] - tail[0][0]) / steps / 1e3
    per = collections.defaultdict(float)
    cnt = collections.Counter()
    for s, e, k in tail:
        per[k] += (e - s) / 1e3 / steps
        cnt[k] += 1
    return period, wall, per, {k: c // steps for k, c in cnt.items()}


steps = int(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 25
pe, we, e, ce = last_steps(sys.argv[1], steps)
pr, wr, r, cr = last_steps(sys.argv[2], steps)
print(f"eager : {pe} launches / step, {sum(e.values()):9.1f} us in kernels, {we:9.1f} us wall per step")
print(f"replay: {pr} launches / step, {sum(r.values()):9.1f} us in kernels, {wr:9.1f} us wall per step")
print("| replay - eager, us / step | eager us | replay us | launches e / r | kernel |")
print("|---|---|---|---|---|")
keys = sorted(set(e) | set(r), key=lambda k: -abs(r.get(k, 0.0) - e.get(k, 0.0)))
for k in keys[:top]:
    name = re.sub(r"\(anonymous namespace\)::", "", k)[:120]
    print(f"| {r.get(k, 0.0) - e.get(k, 0.0):+8.1f} | {e.get(k, 0.0):8.1f} | {r.get(k, 0.0):8.1f} | {ce.get(k, 0)} / {cr.get(k, 0)} | `{name}` |")
