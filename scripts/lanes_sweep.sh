#!/bin/bash
# encode / decode rates by the number of lanes a batch call is cut into (bench.py --lanes sets both directions): scripts/lanes_sweep.sh "W1" "W2" ... -- L1 L2 ...
cd "$(dirname "$0")/.."
WS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do WS+=("$1"); shift; done; shift
for W in "${WS[@]}"; do
  for L in "$@"; do
    timeout -k 10 200 python bench.py $W --lanes $L --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /tmp/l.json 2>/dev/null
    python - "$W" $L <<'PY'
import json, sys
d = json.load(open("/tmp/l.json"))
print(sys.argv[1], "lanes", sys.argv[2], "value", d["value"], "encode", d["encode_MBps"], "decode", d["decode_MBps"])
PY
  done
done
