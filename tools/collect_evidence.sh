#!/bin/bash
# One call on the GPU box: kernel trace (groups + one step's ordered timeline) and the three PMC passes of the flagship bench step, summarised
# into <out>/ (copy the summaries to profiles/ afterwards), then the bench line itself:   bash tools/collect_evidence.sh gpurun_out/r04z
# The PMC summary is stamped with the source tree it was taken on (tools/source_stamp.py); bench.py reports its figures only on that tree.
set -e
out=$1
export TMPDIR=/tmp
mkdir -p "$out"
# (the traced and counted runs enqueue the step eagerly: the same kernels, shapes and buffers the replayed headline runs, attributable per launch)
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --eager-step > "$out/trace.log" 2>&1
python3 tools/prof_groups.py "$out/trace" 25 70 "$out/kernel_groups.md"
python3 tools/prof_timeline.py "$out/trace" "$out/step_timeline.md"
cp "$out"/trace/*/*kernel_stats.csv "$out/kernel_stats.csv"
rm -rf "$out/trace"
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU"; do
    set -- $pass
    name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --eager-step > "$out/$name.log" 2>&1
    echo "pass $name done"
done
python3 tools/pmc_summary.py "$out" 5 "$out/pmc.json" 40
rm -rf "$out/fetch" "$out/write" "$out/sq"
cp "$out/pmc.json" profiles/r06_pmc.json      # (on the box only: lets the bench run below report the counters it was just profiled with)
python3 bench.py --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"
tail -1 "$out/bench.json"
