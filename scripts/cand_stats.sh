#!/bin/bash
# What phase 3 of enc_cand meets (heads per wave, iterations): the -DCAND_STATS build of encode_match.hip, one encode per workload.
#   hipcc ... -DCAND_STATS -c encode_match.hip -> build_abl/liblzfse_mi_stats.so (see the build lines in scripts/cand_phases.sh)
cd "$(dirname "$0")/.."
P=lzfse_rust_amd
cp $P/liblzfse_mi.so /tmp/lib_keep.so
cp build_abl/liblzfse_mi_stats.so $P/liblzfse_mi.so
for W in "snappy --replicas 16" "text64m" ; do
  echo "== $W"
  timeout -k 10 200 python bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-extras --lanes 1 2>&1 >/dev/null | grep cand_stats | head -2
done
cp /tmp/lib_keep.so $P/liblzfse_mi.so
