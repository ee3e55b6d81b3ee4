#!/bin/bash
# A/B of compile-time variants on the GPU box: scripts/ab_defs.sh "SRC1 SRC2" "DEFS_A" "DEFS_B" ... [-- bench args]
# Every variant rebuilds the named sources (without .hip) with its defines over the shipped objects; two bench runs each;
# exclusive kernel times. The shipped library is restored afterwards.
cd "$(dirname "$0")/.."
P=lzfse_rust_amd; SRCS=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fvisibility=hidden"
VARS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VARS+=("$1"); shift; done; [ "$1" == "--" ] && shift
cp $P/liblzfse_mi.so /tmp/lib_keep.so
for D in "${VARS[@]}"; do
  OBJS=$(ls $P/build/prod_*.o)
  for S in $SRCS; do
    hipcc $FLAGS $D -c $P/csrc/$S.hip -o /tmp/abd_$S.o || { cp /tmp/lib_keep.so $P/liblzfse_mi.so; exit 1; }
    OBJS=$(echo "$OBJS" | grep -v "prod_$S\.o"); OBJS="$OBJS
/tmp/abd_$S.o"
  done
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $P/liblzfse_mi.so
  for REP in 1 2; do
    timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > /tmp/ab.json 2>/tmp/ab.err || { tail -3 /tmp/ab.err; cp /tmp/lib_keep.so $P/liblzfse_mi.so; exit 1; }
    python - "$D" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(f"defs='{sys.argv[1]}'", "value", d["value"], "encode", d["encode_MBps"], "decode", d["decode_MBps"], {k: round(v, 2) for k, v in d.get("exclusive_kernel_ms", {}).items() if v > 0.15})
PY
  done
done
cp /tmp/lib_keep.so $P/liblzfse_mi.so
