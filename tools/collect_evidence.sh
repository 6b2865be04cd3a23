#!/bin/bash
# One call on the GPU box: kernel trace + the three PMC passes of the flagship bench step, summarised into <out>/ (copy the summaries to
# profiles/ afterwards):   bash tools/collect_evidence.sh gpurun_out/r02z
set -e
out=$1
export TMPDIR=/tmp
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$out/trace.log" 2>&1
python3 tools/prof_groups.py "$out/trace" 25 60 "$out/kernel_groups.md"
cp "$out"/trace/*/*kernel_stats.csv "$out/kernel_stats.csv"
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU"; do
    set -- $pass
    name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > "$out/$name.log" 2>&1
    echo "pass $name done"
done
python3 tools/pmc_summary.py "$out" 5 "$out/pmc.json" 30
python3 bench.py --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"
tail -1 "$out/bench.json"
