#!/bin/bash
# After scripts/profile_all.sh TAG + `python bench.py > gpurun_out/prof_TAG/bench_final.json` + scripts/bench_all.sh TAG on a GPU
# box: copies what is to be judged from gpurun_out/ into profiles/ (tracked).
TAG=${1:-r03}
P=gpurun_out/prof_$TAG A=gpurun_out/all_$TAG
tail -1 $P/bench_final.json > profiles/${TAG}_bench.json
grep '^{' $P/bench.log | tail -1 > profiles/${TAG}_bench_rocprof.json
cp $P/kernel_stats.csv profiles/${TAG}_bench_kernel_stats.csv
cp $P/pmc_traffic.json profiles/${TAG}_pmc_traffic.json
cp $P/pmc_summary.txt profiles/${TAG}_pmc_summary.txt
cp $P/sq_counters.txt profiles/${TAG}_sq_counters.txt
for w in text64m chunks4m chunks1g snappy_r64; do tail -1 $A/bench_$w.json > profiles/${TAG}_bench_$w.json; done
tail -1 $A/snappy_table.json > profiles/${TAG}_snappy_table.json
for r in r64 r16; do tail -1 $A/snappy_table_$r.json > profiles/${TAG}_snappy_table_$r.json; done
S=gpurun_out/shard_$TAG
if [ -d $S ]; then for n in 2 4 8; do tail -1 $S/shard_$n.json > profiles/${TAG}_shard_$n.json; done; tail -1 $S/shard_1.json > profiles/${TAG}_shard_1.json; cp $S/summary.txt profiles/${TAG}_shard_summary.txt; fi
[ -f $P/cache_counters.txt ] && cp $P/cache_counters.txt profiles/${TAG}_cache_counters.txt
python - "$TAG" <<'PY'
import json, sys
t = sys.argv[1]
d = json.load(open(f"profiles/{t}_bench.json"))
print("default", d["value"], d["encode_MBps"], d["decode_MBps"], "traffic", d["roofline"]["traffic"], d["pcie_inclusive"]["encode_MBps"], d["pcie_inclusive"]["decode_MBps"])
for w in ("snappy_r64", "chunks1g", "chunks4m", "text64m"):
    x = json.load(open(f"profiles/{t}_bench_{w}.json"))
    print(w, x["value"], x["encode_MBps"], x["decode_MBps"])
PY
