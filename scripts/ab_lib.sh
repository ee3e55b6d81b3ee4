#!/bin/bash
# A/B of whole library builds on the GPU box: scripts/ab_lib.sh LIB_A.so LIB_B.so ... [-- bench args]   (the shipped library is restored afterwards)
cd "$(dirname "$0")/.."
P=lzfse_rust_amd
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" == "--" ] && shift
cp $P/liblzfse_mi.so /tmp/lib_keep.so
for L in "${LIBS[@]}"; do
  cp $L $P/liblzfse_mi.so
  for REP in 1 2; do
  timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > /tmp/ab.json 2>/tmp/ab.err || { tail -3 /tmp/ab.err; cp /tmp/lib_keep.so $P/liblzfse_mi.so; exit 1; }
  python - "$L" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(f"lib={sys.argv[1]}", "value", d["value"], "encode", d["encode_MBps"], "decode", d["decode_MBps"], {k: round(v, 2) for k, v in d.get("exclusive_kernel_ms", {}).items() if v > 0.4})
PY
  done
done
cp /tmp/lib_keep.so $P/liblzfse_mi.so
