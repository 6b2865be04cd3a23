#!/bin/bash
# HBM-side traffic per launch of every kernel of a script: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they cannot share one), grouped by
# (kernel, grid); read = 2 x FETCH_SIZE KiB (gfx950 wide coalesced reads report half: MI355X_MICROARCH.md, HBM), write = WRITE_SIZE KiB.
#   bash tools/ktraffic.sh <out.txt> python3 $PWD/tools/x.py [args]
set -e
out=$1; shift
export TMPDIR=/tmp
R=$(pwd)
d=$(mktemp -d /tmp/ktraffic.XXXXXX)
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$d/$c" -- "$@" > "$d/$c.log" 2>&1) || { tail -20 "$d/$c.log"; exit 1; }
done
python3 - "$d" > "$R/$out" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{d}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[(r["Kernel_Name"][:80], r["Grid_Size"])][c].append(float(r["Counter_Value"]))
rows = []
for k, v in acc.items():
    f = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"])) * 2 * 1024
    w = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"])) * 1024
    rows.append((f + w, f, w, len(v["FETCH_SIZE"]), k))
for t, f, w, n, k in sorted(rows, reverse=True)[:40]:
    print(f"{t/1e6:9.1f} MB  (read {f/1e6:8.1f}  write {w/1e6:8.1f})  n={n:3d}  grid {k[1]:>9s}  {k[0]}")
PY
rm -rf "$d"
