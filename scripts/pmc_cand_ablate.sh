#!/bin/bash
# TCP access counts of enc_cand under the LZFSE_MI_CAND_DEBUG ablations (1: no forward compare, 2: no backward, 4: one hop)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for dbg in 0 1 3 7; do
  LZFSE_MI_CAND_DEBUG=$dbg timeout -k 10 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d gpurun_out/pmc_abl_$dbg --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_abl_$dbg.log 2>&1 || exit 1
  echo "dbg $dbg" >> gpurun_out/pmc_abl.txt
  python scripts/pmc_generic.py gpurun_out/pmc_abl_$dbg | grep enc_cand >> gpurun_out/pmc_abl.txt
done
