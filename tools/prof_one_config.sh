# full kernel list + one-step timeline of one configuration of tools/bench_configs.py
# usage (on the GPU box): bash tools/prof_one_config.sh <cfg> <out dir under gpurun_out>
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
c=$1
O=$R/gpurun_out/$2
mkdir -p $O
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tools/bench_configs.py $c --steps 5 --warmup 2 > $O/run.log 2>&1
python3 $R/tools/prof_config.py $O/trace $c 7 120 > $O/kernels.md
python3 $R/tools/prof_timeline.py $O/trace $O/timeline.md
rm -rf $O/trace
echo "$c done"
