#!/bin/bash
# One rank's shard of an N-way split of config 5 (ONE 1 GiB input, 256 x 4 MiB streams) on one GPU: bounds the
# strong-scaling curve from single-GPU records. Usage (GPU box): scripts/shard_bench.sh TAG   -> gpurun_out/shard_TAG/
TAG=${1:-r03}
cd "$(dirname "$0")/.."
OUT=gpurun_out/shard_$TAG
mkdir -p $OUT
python bench.py --workload chunks1g --no-cpu-baseline --no-extras > $OUT/shard_1.json 2> $OUT/err.log || exit 1
for N in 2 4 8; do
  python bench.py --workload chunks1g --emulate-world $N --no-cpu-baseline --no-extras > $OUT/shard_$N.json 2>> $OUT/err.log || exit 1
done
python - "$OUT" > $OUT/summary.txt <<'PY'
import json, sys
o = sys.argv[1]
base = json.load(open(f"{o}/shard_1.json"))
for n in (1, 2, 4, 8):
    d = json.load(open(f"{o}/shard_{n}.json"))
    print(f"N={n}: {d['config']['streams_rank0']:4d} streams  value {d['value']/1e3:6.2f} GB/s  encode {d['encode_MBps']/1e3:6.2f}  decode {d['decode_MBps']/1e3:6.2f}"
          f"  -> {n} ranks at this rate = {n * d['value'] / base['value']:.2f} x one GPU ({100 * d['value'] / base['value']:.0f} % efficiency)")
PY
cat $OUT/summary.txt
