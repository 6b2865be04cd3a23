#!/bin/bash
# LDS-DMA GEMM with a K tail (K % 64 != 0, round 5) against the register-staged kernel the same problems took before (ISEG_GEMM_DMA_MIN_K=4096 sends
# them back there), on the InternImage-B / Swin-T stage-0 and stage-1 products:   bash tools/kbench_gemm_ktail.sh
for shape in "131072 112 112" "131072 448 112" "131072 112 448" "32768 224 224" "32768 896 224" "65536 96 96" "65536 288 96" "65536 384 96" "16384 192 192"; do
    echo "== $shape: LDS-DMA"
    python3 tools/kbench_gemm_dma_ab.py $shape | cut -c1-90
    echo "== $shape: register-staged"
    ISEG_GEMM_DMA_MIN_K=4096 python3 tools/kbench_gemm_dma_ab.py $shape | cut -c1-90
done
