#!/bin/bash
# LDS bank conflicts per kernel of the flagship step (one rocprofv3 --pmc pass over bench.py --eager-step):   bash tools/step_lds_conflicts.sh <out dir>
# or of another configuration:   bash tools/step_lds_conflicts.sh <out dir> tools/bench_configs.py cfg3 --steps 3 --warmup 2
# per kernel and grid: launches, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (share of the LDS cycles spent on conflicts), SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES
set -e
out=$1; shift
if [ $# -eq 0 ]; then set -- bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --eager-step; fi
script=$1; shift
export TMPDIR=/tmp
R=$(pwd)
mkdir -p "$out"
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$R/$out/lds" -- python3 "$R/$script" "$@" > "$R/$out/lds.log" 2>&1) || { tail -5 "$out/lds.log"; exit 1; }
python3 - "$out" <<'PY' > "$out/step_lds_conflicts.md"
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/lds/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"], row.get("Grid_Size", ""))
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_WAVE_CYCLES":
            n[k] += 1
print("| conflict cycles (M) | share of LDS cycles | LDS wait / wave cycles | launches | kernel | grid |")
print("|---|---|---|---|---|---|")
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_BANK_CONFLICT", 0.0))[:40]:
    c, a = d.get("SQ_LDS_BANK_CONFLICT", 0.0), d.get("SQ_LDS_IDX_ACTIVE", 0.0)
    w, wc = d.get("SQ_WAIT_INST_LDS", 0.0), d.get("SQ_WAVE_CYCLES", 1.0)
    name = re.sub(r"\(anonymous namespace\)::", "", k[0])[:100]
    print(f"| {c / 1e6:.2f} | {c / a if a else 0:.2f} | {w / wc:.3f} | {n[k]} | `{name}` | {k[1]} |")
PY
rm -rf "$out/lds"
cat "$out/step_lds_conflicts.md"
