set -e
mkdir -p gpurun_out/r06wide
: > gpurun_out/r06wide/k.txt
for i in 1 2; do for v in 0 1; do echo "128X256=$v" >> gpurun_out/r06wide/k.txt; ISEG_GEMM_DMA_128X256=$v KB_PADS=0 timeout -k 10 300 python3 tools/kbench_pitch.py 2>&1 | grep "C=384" >> gpurun_out/r06wide/k.txt; done; done
cat gpurun_out/r06wide/k.txt
ISEG_GEMM_DMA_128X256=1 timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -2
for i in 1 2 3; do for v in 0 1; do ISEG_GEMM_DMA_128X256=$v timeout -k 10 600 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('128X256=$v', d['ms_per_step'], d['value'])"; done; done
