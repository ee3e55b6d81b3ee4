#!/bin/bash
# The round's fuzz campaigns, every mode of scripts/fuzz_gpu.py: scripts/fuzz_all.sh SEED0 [scale]
cd "$(dirname "$0")/.."
S=${1:-900}; K=${2:-1}
run() { timeout -k 10 800 python scripts/fuzz_gpu.py $1 $2 $3 2>&1 | tail -1 | sed "s/^/$3 seed $2: /"; }
run $((10 * K)) $((S + 1)) ""
run $((6 * K)) $((S + 2)) chain
run $((5 * K)) $((S + 3)) ring
run $((5 * K)) $((S + 4)) big
run $((8 * K)) $((S + 5)) stream
run $((3 * K)) $((S + 6)) walk
run $((3 * K)) $((S + 7)) pipe
run $((2 * K)) $((S + 8)) pipeck
