#!/bin/bash
# Round profile of the default bench command: kernel stats, then HBM traffic counters in separate passes
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; never together with trace domains other than the
# kernel trace). Outputs land in gpurun_out/prof_$TAG; copy what is to be judged into profiles/.
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python bench.py --steps 5 --warmup 1 > $OUT/bench.log 2>&1 || exit 1
echo "stats done" >> $OUT/progress.txt
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.log 2>&1 || exit 1
echo "fetch done" >> $OUT/progress.txt
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_write.log 2>&1 || exit 1
python scripts/summarize_pmc.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json 4 > $OUT/pmc_summary.txt
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/pmc_sq --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_sq.log 2>&1 || exit 1
python scripts/pmc_generic.py $OUT/pmc_sq > $OUT/sq_counters.txt
echo "sq done" >> $OUT/progress.txt
timeout -k 10 400 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc_cache --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --lanes 1 > $OUT/pmc_cache.log 2>&1 && python scripts/pmc_generic.py $OUT/pmc_cache > $OUT/cache_counters.txt
rm -rf $OUT/pmc_cache $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
tail -1 $OUT/bench.log | cut -c1-400
