#!/bin/bash
# Everything a round's profiles/ needs beyond tools/collect_evidence.sh, in one call on the GPU box:   bash tools/collect_round.sh gpurun_out/r03round
#   configs.jsonl        tools/bench_configs.py, eager and --graph-step (builder-run numbers of DESIGN 5.1)
#   configs_kernels.md   kernel shares of cfg1 / cfg3 / cfg4 / cfg5 (tools/prof_config.py)
#   dp_overhead_{c10d,native}.txt   tools/dp_overhead.py (world of one rank through RCCL), one file per exchange
#   kernels.txt          single-kernel loops (fused-stage backward, strided data gradient, DCNv3)
set -e
out=$1
export TMPDIR=/tmp
R=$(pwd)
mkdir -p "$out"
: > "$out/configs.jsonl"
for c in cfg1 cfg3 cfg4 cfg4_infer cfg5 v2; do      # one process per configuration: the last of six in one process ran 15 % slow
  python3 tools/bench_configs.py $c --steps 20 --warmup 5 2>/dev/null | grep ms_per_step >> "$out/configs.jsonl"
done
echo "eager configs done"
for c in cfg1 cfg3 cfg4 cfg4_infer cfg5 v2; do
  python3 tools/bench_configs.py $c --graph-step --steps 20 --warmup 5 2>/dev/null | grep ms_per_step >> "$out/configs.jsonl"
done
echo "graph configs done"
: > "$out/configs_kernels.md"
for c in cfg1 cfg3 cfg4 cfg5; do
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d "$R/$out/prof_$c" -- python3 "$R/tools/bench_configs.py" $c --steps 5 --warmup 2 > "$R/$out/prof_$c.log" 2>&1)
  python3 tools/prof_config.py "$out/prof_$c" $c 7 20 >> "$out/configs_kernels.md"
  rm -rf "$out/prof_$c" "$out/prof_$c.log"
  echo "$c profiled"
done
python3 tools/dp_overhead.py 30 > "$out/dp_overhead_c10d.txt" 2>&1 || true                 # c10d work objects (the default exchange of a hand-launched job)
python3 tools/dp_overhead.py 30 native > "$out/dp_overhead_native.txt" 2>&1 || true        # stream-ordered RCCL through the C ABI (what bench.py --gpus N runs after its probe)
echo "dp overhead done"
{ python3 tools/kbench_mlp_bwd.py; python3 tools/kbench_conv_strided.py; python3 tools/kbench_dcn.py; python3 tools/kbench_stream.py fpn; python3 tools/kbench_stream.py cold; } > "$out/kernels.txt" 2>&1 || true
echo "kernel loops done"
