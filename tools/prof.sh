#!/bin/bash
# usage: tools/prof.sh <tag> <logn>   -- kernel-trace stats + two PMC passes, summaries under gpurun_out/<tag>/
set -e
TAG=$1; LOGN=${2:-28}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/prof_run.py $LOGN 3 > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -- python3 $GRAFT_REPO_ROOT/tools/prof_run.py $LOGN 1 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc2 -- python3 $GRAFT_REPO_ROOT/tools/prof_run.py $LOGN 1 > $OUT/pmc2.log 2>&1
find $OUT -name "*.csv" | head -20
