#!/bin/bash
# HBM traffic / matrix-core counters per kernel of one configuration of tools/bench_configs.py (three separate --pmc passes, as tools/collect_evidence.sh
# does for the flagship):   bash tools/pmc_config.sh cfg3 gpurun_out/pmc_cfg3   ->  <out>/pmc.json, <out>/pmc.txt
set -e
cfg=$1; out=$2
export TMPDIR=/tmp
R=$(pwd)
mkdir -p "$out"
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU"; do
    set -- $pass
    name=$1; shift
    (cd /tmp && rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$R/$out/$name" -- python3 "$R/tools/bench_configs.py" $cfg --steps 3 --warmup 2 > "$R/$out/$name.log" 2>&1)
    echo "pass $name done"
done
python3 tools/pmc_summary.py "$out" 5 "$out/pmc.json" 60
rm -rf "$out/fetch" "$out/write" "$out/sq"
python3 - "$out/pmc.json" > "$out/pmc.txt" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k in d.get("kernels", [])[:60]:
    print(f"{k.get('ms_per_step', 0):7.3f} ms/step  {k.get('avg_us', 0):8.1f} us  rd {k.get('read_bytes', 0) / 1e6:8.1f} MB  wr {k.get('write_bytes', 0) / 1e6:8.1f} MB  "
          f"{k.get('hbm_gbs', 0):7.0f} GB/s  mfma {k.get('mfma_occupancy', 0):.3f}  {k['kernel'][:70]} grid {k.get('grid')}")
PY
