#!/bin/bash
# Everything the round's profiles/ files come from, on one GPU box (about 15 minutes); then scripts/collect_profiles.sh TAG here.
TAG=${1:-r05}
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
bash scripts/profile_all.sh $TAG || exit 1
cp gpurun_out/prof_$TAG/pmc_traffic.json profiles/${TAG}_pmc_traffic.json
python bench.py > gpurun_out/prof_$TAG/bench_final.json 2> gpurun_out/prof_$TAG/bench_final.err || exit 1
echo "final bench done" >> gpurun_out/prof_$TAG/progress.txt
bash scripts/bench_all.sh $TAG || exit 1
bash scripts/shard_bench.sh $TAG || exit 1
