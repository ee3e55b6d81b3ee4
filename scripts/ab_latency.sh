#!/bin/bash
# A/B of library builds on the small-batch / single-stream decode floors: scripts/ab_latency.sh LIB.so ...
cd "$(dirname "$0")/.."
P=lzfse_rust_amd
cp $P/liblzfse_mi.so /tmp/lib_keep.so
for L in "$@"; do
  cp $L $P/liblzfse_mi.so
  echo "== $L"
  timeout -k 10 100 python scripts/single_latency.py 2>&1 | tail -2
  timeout -k 10 200 python bench.py --workload chunks1g --emulate-world 8 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('shard-8: value', d['value'], 'decode', d['decode_MBps'], {k:v for k,v in d['kernel_ms_per_step'].items() if k.startswith('dec')})"
  timeout -k 10 200 python bench.py --workload text64m --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('text64m: decode', d['decode_MBps'], {k:v for k,v in d['kernel_ms_per_step'].items() if k.startswith('dec')})"
done
cp /tmp/lib_keep.so $P/liblzfse_mi.so
