#!/bin/bash
# the fused ConvNeXt-MLP kernels with the conflict-free piece layout (built) against the row-major pieces of rounds 2-5 (iseg_amd/lib/ab/libiseg_hip_rowmajor.so,
# tools/ab_build.py "ISEG_MLP_PIECE_ROWMAJOR" mlp_fused mlp_wgrad): parity tests, kernel loops, flagship step -- interleaved on one box
set -e
out=gpurun_out/r06mlp
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_mlp_fused_gpu.py tests/test_blocks_gpu.py tests/test_attention_gpu.py -x -q 2>&1 | tail -3
lib=iseg_amd/lib/libiseg_hip.so
alt=iseg_amd/lib/ab/libiseg_hip_rowmajor.so
cp $lib /tmp/lib_cur.so
trap 'cp /tmp/lib_cur.so '"$lib" EXIT
: > $out/kernels.txt
for rep in 1 2; do
  for which in built rowmajor; do
    if [ $which = built ]; then cp /tmp/lib_cur.so $lib; else cp $alt $lib; fi
    echo "== $which" >> $out/kernels.txt
    timeout -k 10 300 python3 tools/kbench_mlp.py 2>&1 | grep -v amdgpu.ids >> $out/kernels.txt
    timeout -k 10 300 python3 tools/kbench_mlp_bwd.py 2>&1 | grep -v amdgpu.ids >> $out/kernels.txt
  done
done
cat $out/kernels.txt
: > $out/step.txt
for rep in 1 2 3; do
  for which in built rowmajor; do
    if [ $which = built ]; then cp /tmp/lib_cur.so $lib; else cp $alt $lib; fi
    timeout -k 10 600 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$which', d['ms_per_step'], d['value'])" | tee -a $out/step.txt
  done
done
