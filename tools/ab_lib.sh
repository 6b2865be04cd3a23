#!/bin/bash
# interleaved flagship bench lines of the built library against an alternative one:   bash tools/ab_lib.sh iseg_amd/lib/ab/libiseg_hip_X.so [reps]
set -e
alt=$1; reps=${2:-3}
lib=iseg_amd/lib/libiseg_hip.so
cp $lib /tmp/lib_cur.so
trap 'cp /tmp/lib_cur.so '"$lib" EXIT
for rep in $(seq $reps); do
  cp /tmp/lib_cur.so $lib
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('built', d['ms_per_step'])"
  cp $alt $lib
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('alt  ', d['ms_per_step'])"
done
