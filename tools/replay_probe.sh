#!/bin/bash
# cfg3 (or any configuration) replayed against eager: process-to-process spread of both forms on ONE box, then the per-kernel difference.
#   bash tools/replay_probe.sh <out dir> <config> [repeats]
set -e
out=$1; cfg=$2; rep=${3:-3}
export TMPDIR=/tmp
R=$(pwd)
mkdir -p "$out"
: > "$out/spread.txt"
for i in $(seq $rep); do
  for form in "" "--graph-step"; do
    python3 tools/bench_configs.py $cfg $form --steps 40 --warmup 8 2>/dev/null | grep ms_per_step | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config'], d['step'], d['ms_per_step'])" >> "$out/spread.txt"
  done
done
cat "$out/spread.txt"
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d "$R/$out/tr_eager" -- python3 "$R/tools/bench_configs.py" $cfg --steps 6 --warmup 4 > "$R/$out/tr_eager.log" 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d "$R/$out/tr_replay" -- python3 "$R/tools/bench_configs.py" $cfg --graph-step --steps 6 --warmup 4 > "$R/$out/tr_replay.log" 2>&1)
grep ms_per_step "$out/tr_eager.log" "$out/tr_replay.log" | cut -c1-200
python3 tools/replay_vs_eager.py "$out/tr_eager" "$out/tr_replay" 4 30 > "$out/diff.md"
cat "$out/diff.md"
rm -rf "$out/tr_eager" "$out/tr_replay"
