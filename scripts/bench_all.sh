#!/bin/bash
# Every bench line of a round besides the default one (scripts/profile_all.sh): the other BASELINE configs, the small
# batch, and the per-file table. Outputs: gpurun_out/all_$TAG/*.json (copy what is to be judged into profiles/).
TAG=${1:-r03}
OUT=gpurun_out/all_$TAG
mkdir -p $OUT
for w in text64m chunks4m chunks1g; do
  timeout -k 10 400 python bench.py --workload $w --steps 5 --warmup 1 > $OUT/bench_$w.json 2> $OUT/bench_$w.err || exit 1
  echo "$w done" >> $OUT/progress.txt
done
timeout -k 10 300 python bench.py --replicas 64 --steps 20 --warmup 3 > $OUT/bench_snappy_r64.json 2> $OUT/bench_snappy_r64.err || exit 1
timeout -k 10 300 python bench.py --per-file 256 > $OUT/snappy_table.json 2> $OUT/snappy_table.err || exit 1
timeout -k 10 300 python bench.py --per-file 64 > $OUT/snappy_table_r64.json 2> $OUT/snappy_table_r64.err || exit 1
timeout -k 10 300 python bench.py --per-file 16 > $OUT/snappy_table_r16.json 2> $OUT/snappy_table_r16.err || exit 1
echo "tables done" >> $OUT/progress.txt
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d["value"], d.get("encode_MBps"), d.get("decode_MBps"))
PY
done
