#!/bin/bash
# One rocprofv3 --pmc pass of the bench with an arbitrary counter set (kernel averages via scripts/pmc_generic.py). Usage:
#   scripts/profile_counters.sh TAG "COUNTER COUNTER ..." [bench args...]   -> gpurun_out/ctr_TAG.txt
TAG=$1; CTRS=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ctr_$TAG
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --pmc $CTRS -d $OUT/raw --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras "$@" > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python scripts/pmc_generic.py $OUT/raw > gpurun_out/ctr_$TAG.txt
rm -rf $OUT/raw
cat gpurun_out/ctr_$TAG.txt
