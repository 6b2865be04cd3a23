#!/bin/bash
# LDS counters of one binary's kernels:  tools/pmc_bin_lds.sh <out_dir> <binary> [args...]
out=$1; shift
export TMPDIR=/tmp
mkdir -p "$out"
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$out/lds" -- "$@" > "$out/lds.log" 2>&1 || tail -3 "$out/lds.log"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/lds/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"][:90]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:24s} {sum(v)/len(v):16.0f}   (n={len(v)})")
PY
