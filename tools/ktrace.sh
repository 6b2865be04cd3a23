#!/bin/bash
# kernel-only durations of any script (rocprofv3 --kernel-trace), grouped by (kernel, grid):   bash tools/ktrace.sh <out.md> <rows> python3 tools/x.py [args]
set -e
out=$1; rows=$2; shift 2
export TMPDIR=/tmp
R=$(pwd)
d=$(mktemp -d /tmp/ktrace.XXXXXX)
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d "$d" -- "$@" > "$d.log" 2>&1) || { tail -20 "$d.log"; exit 1; }
python3 "$R/tools/prof_groups.py" "$d" 1 "$rows" "$R/$out"
rm -rf "$d" "$d.log"
