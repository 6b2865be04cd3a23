#!/bin/bash
# A/B of the LDS-DMA GEMM tile forms that fill the CUs in single-round launches (round 6), interleaved on one box:
#   bash tools/gemm_fill_ab.sh            kernel loops (tools/kbench_pitch.py: the four stage-2 and stage-3 data-path launches) and the flagship step
# with ISEG_GEMM_DMA_128X192 / ISEG_GEMM_DMA_64X192 = 0 (256 x 128 / 128 x 128 tiles) and 1 (default).  profiles/r06_gemm_fill_the_cus_ab.txt
set -e
mkdir -p gpurun_out/r06pack
: > gpurun_out/r06pack/kernels.txt
for i in 1 2; do
  for v in 0 1; do
    echo "128X192=$v 64X192=$v" >> gpurun_out/r06pack/kernels.txt
    ISEG_GEMM_DMA_128X192=$v ISEG_GEMM_DMA_64X192=$v KB_PADS=0 timeout -k 10 300 python3 tools/kbench_pitch.py 2>&1 | grep "C=" >> gpurun_out/r06pack/kernels.txt
  done
done
cat gpurun_out/r06pack/kernels.txt
for i in 1 2 3; do
  for v in 0 1; do
    ISEG_GEMM_DMA_128X192=$v ISEG_GEMM_DMA_64X192=$v timeout -k 10 600 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('forms=$v', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r06pack/step_ab.txt
  done
done
