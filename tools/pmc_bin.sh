#!/bin/bash
# SQ counters of a stand-alone binary (one rocprofv3 --pmc pass per group): tools/pmc_bin.sh <out_dir> <kernel-name filter> <binary> [args]
set -e
out=$1; flt=$2; shift 2
export TMPDIR=/tmp
mkdir -p "$out"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM" "SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_WAIT_ANY"; do
    rocprofv3 --pmc $grp --output-format csv -d "$out/g$i" -- "$@" > "$out/g$i.log" 2>&1 || { tail -5 "$out/g$i.log"; }
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
