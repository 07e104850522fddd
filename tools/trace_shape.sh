#!/bin/bash
# usage: tools/trace_shape.sh <tag> <logn> <kind>  -- kernel-trace stats of tools/prof_shape.py
set -e
TAG=$1; LOGN=$2; KIND=$3
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/prof_shape.py $LOGN $KIND > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log | cut -c1-600
S=$(find $OUT/trace -name "*kernel_stats.csv" | head -n 1)
python3 - "$S" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:64]:64s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
PY
