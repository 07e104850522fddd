#!/bin/bash
# usage (on the GPU box): tools/sq_counters.sh <tag> [logn=30] [u32|u64|pairs]
# SQ_* counters (one --pmc pass, 8 SQ slots) of the sorts of tools/variant_run.py -> gpurun_out/<tag>/sq_k2
# (profiles/r02_sq_counters.json also holds round 1's direct kernel, measured before it was removed)
set -e
TAG=$1; LOGN=${2:-30}; KIND=${3:-u32}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for K in 2; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU \
    --output-format csv -d $OUT/sq_k$K -- python3 $GRAFT_REPO_ROOT/tools/variant_run.py product $LOGN $KIND > $OUT/sq_k$K.log 2>&1
done
echo done
