#!/bin/bash
# wave-state counters of one single-kernel script:  tools/pmc_sq.sh <out_dir> <script.py> <kernel-name filter>
set -e
out=$1; script=$2; flt=${3:-}
export TMPDIR=/tmp
mkdir -p "$out"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS SQ_THREAD_CYCLES_VALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE"; do
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$out/g$i" -- python3 "$script" > "$out/g$i.log" 2>&1 || { tail -3 "$out/g$i.log"; }
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
            acc[k[:100]][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/g0/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            dur[r["Kernel_Name"][:100]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
for k, d in acc.items():
    print(k, "avg us", sum(dur[k]) / max(len(dur[k]), 1))
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.0f}   (n={len(v)})")
PY
