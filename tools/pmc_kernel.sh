#!/bin/bash
# SQ / HBM counters of one single-kernel script (one rocprofv3 --pmc pass per counter group):
#   tools/pmc_kernel.sh <out_dir> <script.py> [kernel-name filter]
# Prints, per kernel matching the filter, the per-launch average of every counter.
set -e
out=$1; script=$2; flt=${3:-}
export TMPDIR=/tmp
mkdir -p "$out"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SALU"; do
    rocprofv3 --pmc $grp --output-format csv -d "$out/g$i" -- python3 "$script" > "$out/g$i.log" 2>&1 || { tail -5 "$out/g$i.log"; }
    i=$((i+1))
done
python3 - "$out" "$flt" <<'PY'
import csv, glob, sys, collections
out, flt = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if flt in k:
            acc[k[:90]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.0f}   (n={len(v)})")
PY
