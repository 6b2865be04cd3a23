set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in cfg1 cfg3 cfg4; do
  timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_$c -- python3 $R/tools/bench_configs.py $c --steps 5 --warmup 2 > $R/gpurun_out/prof_$c.log 2>&1
  python3 $R/tools/prof_config.py $R/gpurun_out/prof_$c $c 7 16 > $R/gpurun_out/prof_${c}_r03.md
  rm -rf $R/gpurun_out/prof_$c
  echo "$c done"
done
