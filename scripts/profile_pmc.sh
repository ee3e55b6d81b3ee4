#!/bin/bash
# PMC passes of one bench command (separate runs: MI355X_MICROARCH.md, rocprofv3 PMC slots). Usage:
#   scripts/profile_pmc.sh TAG [bench args...]      -> gpurun_out/pmc_TAG/{sq,fetch,write}.txt
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/sq --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $OUT/sq.log 2>&1 || exit 1
python scripts/pmc_generic.py $OUT/sq > $OUT/sq.txt
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $OUT/fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/write --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $OUT/write.log 2>&1 || exit 1
python scripts/summarize_pmc.py $OUT/fetch $OUT/write $OUT/traffic.json > $OUT/traffic.txt
rm -rf $OUT/sq $OUT/fetch $OUT/write
cat $OUT/sq.txt $OUT/traffic.txt
