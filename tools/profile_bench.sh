#!/bin/bash
# usage (on the GPU box): tools/profile_bench.sh <tag> [config] [extra bench.py args]
# 1. plain bench.py  2. rocprofv3 --kernel-trace --stats of the same command  3./4. PMC passes (FETCH_SIZE, WRITE_SIZE,
# separate runs as the MI355X guide prescribes).  Afterwards (in the container): tools/make_profile_summary.py <tag> <round> [config]
set -e
TAG=$1; CFG=${2:-c2}; shift; shift || true
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python3 bench.py --config $CFG --no-other-configs "$@" > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --config $CFG --no-other-configs --no-cpu-baseline > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --config $CFG --steps 1 --warmup 0 --no-other-configs --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --config $CFG --steps 1 --warmup 0 --no-other-configs --no-cpu-baseline > $OUT/pmc_write.log 2>&1
echo done
