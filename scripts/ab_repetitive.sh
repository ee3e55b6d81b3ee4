#!/bin/bash
# scripts/repetitive_bench.py with each of the given library builds: scripts/ab_repetitive.sh LIB.so ...
cd "$(dirname "$0")/.."
P=lzfse_rust_amd
cp $P/liblzfse_mi.so /tmp/lib_keep.so
for L in "$@"; do
  cp $L $P/liblzfse_mi.so
  echo "== $L"
  timeout -k 10 300 python scripts/repetitive_bench.py 2>&1 | tail -12
done
cp /tmp/lib_keep.so $P/liblzfse_mi.so
