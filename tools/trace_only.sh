#!/bin/bash
# usage (on the GPU box): tools/trace_only.sh <tag> [logn=30] [u32|zipf|dup<k>|u64|pairs] [library suffix=product]  -- rocprofv3 kernel-trace stats of one
# configuration of tools/variant_run.py, under gpurun_out/<tag>/
set -e
TAG=$1; LOGN=${2:-30}; KIND=${3:-u32}; LIBN=${4:-product}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/variant_run.py $LIBN $LOGN $KIND > $OUT/trace.log 2>&1
S=$(find $OUT/trace -name "*kernel_stats.csv" | head -n 1)
python3 - "$S" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f} {r['Percentage']}%")
PY
