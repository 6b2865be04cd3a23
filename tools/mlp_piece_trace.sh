#!/bin/bash
set -e
out=gpurun_out/r06mlp
mkdir -p $out
lib=iseg_amd/lib/libiseg_hip.so
alt=iseg_amd/lib/ab/libiseg_hip_rowmajor.so
cp $lib /tmp/lib_cur.so
trap 'cp /tmp/lib_cur.so '"$lib" EXIT
for which in built rowmajor; do
  if [ $which = built ]; then cp /tmp/lib_cur.so $lib; else cp $alt $lib; fi
  bash tools/ktrace.sh $out/trace_$which.md 40 python3 /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --eager-step
  grep "convnext_mlp" $out/trace_$which.md | cut -c1-130
done
