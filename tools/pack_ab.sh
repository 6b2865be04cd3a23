set -e
mkdir -p gpurun_out/r06pack
: > gpurun_out/r06pack/ab4.txt
for i in 1 2; do
  for v in 0 1; do
    echo "64X192=$v" >> gpurun_out/r06pack/ab4.txt
    ISEG_GEMM_DMA_64X192=$v KB_PADS=0 timeout -k 10 300 python3 tools/kbench_pitch.py 2>&1 | grep "C=768" >> gpurun_out/r06pack/ab4.txt
  done
done
cat gpurun_out/r06pack/ab4.txt
ISEG_GEMM_DMA_64X192=1 timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -2
for c in cfg3 cfg5 cfg1; do
  for f in "" "--graph-step"; do
    python3 tools/bench_configs.py $c $f --steps 20 --warmup 6 2>/dev/null | grep ms_per_step | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config'], d['step'], d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'])" | tee -a gpurun_out/r06pack/host_enqueue.txt
  done
done
