#!/bin/bash
# How sensitive is enc_cand to vector instructions? Builds with CAND_PAD_VALU extra (dependent) vector instructions per wave at the
# end of the kernel: 1nnn = nnn full-rate v_add_u32, 2nnn = nnn half-rate v_alignbit_b32. The records stay right.
cd "$(dirname "$0")/.."
P=lzfse_rust_amd
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fvisibility=hidden"
cp $P/liblzfse_mi.so /tmp/lib_keep.so
for N in 0 1064 1128 2064 2128; do
  D=""; [ $N != 0 ] && D="-DCAND_PAD_VALU=$N"
  hipcc $FLAGS $D -c $P/csrc/encode_match.hip -o /tmp/pad_em.o || exit 1
  OBJS=$(ls $P/build/prod_*.o | grep -v "prod_encode_match.o")
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/pad_em.o -o $P/liblzfse_mi.so
  for REP in 1 2; do
    timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > /tmp/pad.json 2>/tmp/pad.err || { tail -3 /tmp/pad.err; cp /tmp/lib_keep.so $P/liblzfse_mi.so; exit 1; }
    python -c "
import json; d=json.load(open('/tmp/pad.json')); print('PAD=$N enc_cand', d['exclusive_kernel_ms']['enc_cand'], 'encode', d['encode_MBps'])"
  done
done
cp /tmp/lib_keep.so $P/liblzfse_mi.so
