#!/bin/bash
# Same-box A/B of the flagship step: this tree against this tree with round 4's arithmetic and routes (the library copy built by
#   python3 tools/ab_build.py "ISEG_GELU_NOPOLY ISEG_GELU_SIG3" conv_igemm elementwise gemm gemm_nn gemm_nt gemm_tn mlp_fused mlp_wgrad
# and saved as iseg_amd/lib/ab/libiseg_hip_r4arith.so; pair launch and matrix-core depthwise switched off by their knobs).
# Kernel-trace families of both, then the replayed bench line of both, interleaved twice:   bash tools/collect_ab_trace.sh gpurun_out/r05ab
set -e
out=$1
export TMPDIR=/tmp
mkdir -p "$out"
lib=iseg_amd/lib/libiseg_hip.so
alt=iseg_amd/lib/ab/libiseg_hip_r4arith.so
test -f "$alt"
cp "$lib" "$out/lib_r5.so"
restore() { cp "$out/lib_r5.so" "$lib"; rm -f "$out/lib_r5.so"; }
trap restore EXIT
run_side() {      # $1 = label, remaining = env assignments
    label=$1; shift
    env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > "$out/bench_$label.json" 2> "$out/bench_$label.err"
    python3 - "$out/bench_$label.json" "$label" <<'EOF'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["ms_per_step"], "ms/step", d["value"], d["unit"], flush=True)
EOF
}
R4ENV="ISEG_WGRAD_PAIR=0 ISEG_GEMM_TN_PAIR=0 ISEG_DW_MFMA=0"
for rep in 1 2; do
    run_side "r5_$rep" ISEG_AB=r5
    cp "$alt" "$lib"
    run_side "r4arith_$rep" $R4ENV
    cp "$out/lib_r5.so" "$lib"
done
# traces (eager steps, attributable per launch); python3 is the traced program itself, the knobs come from the exported environment
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_r5" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --eager-step > "$out/trace_r5.log" 2>&1
python3 tools/prof_groups.py "$out/trace_r5" 25 40 "$out/kernel_groups_r5.md"
rm -rf "$out/trace_r5"
cp "$alt" "$lib"
export $R4ENV
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_r4" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --eager-step > "$out/trace_r4.log" 2>&1
python3 tools/prof_groups.py "$out/trace_r4" 25 40 "$out/kernel_groups_r4arith.md"
rm -rf "$out/trace_r4"
head -20 "$out/kernel_groups_r5.md" "$out/kernel_groups_r4arith.md"
