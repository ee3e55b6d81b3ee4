#!/bin/bash
# After scripts/profile_all.sh TAG + `python bench.py > gpurun_out/prof_TAG/bench_final.json` + scripts/bench_all.sh TAG on a GPU
# box: copies what is to be judged from gpurun_out/ into profiles/ (tracked).
TAG=${1:-r02}
P=gpurun_out/prof_$TAG A=gpurun_out/all_$TAG
tail -1 $P/bench_final.json > profiles/${TAG}_bench.json
grep '^{' $P/bench.log | tail -1 > profiles/${TAG}_bench_rocprof.json
cp $P/kernel_stats.csv profiles/${TAG}_bench_kernel_stats.csv
cp $P/pmc_traffic.json profiles/${TAG}_pmc_traffic.json
cp $P/pmc_summary.txt profiles/${TAG}_pmc_summary.txt
cp $P/sq_counters.txt profiles/${TAG}_sq_counters.txt
for w in text64m chunks4m chunks1g snappy_r64; do tail -1 $A/bench_$w.json > profiles/${TAG}_bench_$w.json; done
tail -1 $A/snappy_table.json > profiles/${TAG}_snappy_table.json
python - "$TAG" <<'PY'
import json, sys
t = sys.argv[1]
d = json.load(open(f"profiles/{t}_bench.json"))
print("default", d["value"], d["encode_MBps"], d["decode_MBps"], "traffic", d["roofline"]["traffic"], d["pcie_inclusive"]["encode_MBps"], d["pcie_inclusive"]["decode_MBps"])
for w in ("snappy_r64", "chunks1g", "chunks4m", "text64m"):
    x = json.load(open(f"profiles/{t}_bench_{w}.json"))
    print(w, x["value"], x["encode_MBps"], x["decode_MBps"])
PY
