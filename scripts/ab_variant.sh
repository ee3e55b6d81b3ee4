#!/bin/bash
# A/B timing of a compile-time variant of one kernel source on the GPU box (the library is restored afterwards):
#   scripts/ab_variant.sh encode_match "-DFOO -DBAR=1" [bench args...]
set -e
cd "$(dirname "$0")/.."
P=lzfse_rust_amd; SRC=$1; DEFS=$2; shift; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fvisibility=hidden"
cp $P/liblzfse_mi.so /tmp/lib_keep.so
for D in "" "$DEFS"; do
  hipcc $FLAGS $D -c $P/csrc/$SRC.hip -o /tmp/ab_$SRC.o
  OBJS=$(ls $P/build/prod_*.o | grep -v "prod_$SRC.o")
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/ab_$SRC.o -o $P/liblzfse_mi.so
  timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > /tmp/ab.json 2>/dev/null
  python - "$D" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(f"defs='{sys.argv[1]}'", "value", d["value"], "encode", d["encode_MBps"], "decode", d["decode_MBps"], {k: v for k, v in d.get("exclusive_kernel_ms", {}).items() if v > 0.5})
PY
done
cp /tmp/lib_keep.so $P/liblzfse_mi.so
