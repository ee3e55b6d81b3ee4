#!/bin/bash
# Where do enc_cand's vector instructions go? The kernel cut short after each phase (-DCAND_ABL=n, encode_match.hip: wrong records,
# so the bench runs with --skip-verify), SQ counters per build -> gpurun_out/r04_cand_phases.txt. The shipped library is restored.
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
P=lzfse_rust_amd
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fvisibility=hidden"
cp $P/liblzfse_mi.so /tmp/lib_keep.so
CTRS=${CTRS:-"SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU"}
OUT=gpurun_out/r04_cand_phases${TAG}.txt; : > $OUT
for N in 1 2 3 4 5 0; do
  D=""; [ $N != 0 ] && D="-DCAND_ABL=$N"
  hipcc $FLAGS $D -c $P/csrc/encode_match.hip -o /tmp/abl_em.o || exit 1
  OBJS=$(ls $P/build/prod_*.o | grep -v "prod_encode_match.o")
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/abl_em.o -o $P/liblzfse_mi.so
  (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --pmc $CTRS -d /tmp/abl_$N --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --lanes 1 > /tmp/abl_$N.log 2>&1) || { tail -5 /tmp/abl_$N.log; cp /tmp/lib_keep.so $P/liblzfse_mi.so; exit 1; }
  echo "CAND_ABL=$N $(python scripts/pmc_generic.py /tmp/abl_$N | grep enc_cand)" | tee -a $OUT
  rm -rf /tmp/abl_$N
done
cp /tmp/lib_keep.so $P/liblzfse_mi.so
