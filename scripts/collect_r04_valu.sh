#!/bin/bash
# Round 4, question 1: what bounds enc_cand? (a) the VALU issue-rate microbenchmark, (b) the VALU-busy counters of the kernel.
#   gpurun -- bash scripts/collect_r04_valu.sh   -> gpurun_out/r04_valu_rate.txt, r04_counters_avail.txt, ctr_r04_*.txt
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out build_abl
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o build_abl/valu_rate scripts/micro/valu_rate.hip || exit 1
timeout -k 10 240 build_abl/valu_rate > gpurun_out/r04_valu_rate.txt 2>&1 || exit 1
cat gpurun_out/r04_valu_rate.txt
(cd /tmp && TMPDIR=/tmp timeout -k 10 120 rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r04_counters_avail_full.txt 2>&1)
grep -o -E "\b(SQ|SQC|TA|TCP|TCC|GRBM|SPI)_[A-Z0-9_]+|\b[A-Za-z]+(Busy|Util[a-z]*)\b" gpurun_out/r04_counters_avail_full.txt | sort -u > gpurun_out/r04_counters_avail.txt
wc -l gpurun_out/r04_counters_avail.txt
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04_bench_base.json 2> gpurun_out/r04_bench_base.err || exit 1
bash scripts/profile_counters.sh r04_valu_a "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU GRBM_GUI_ACTIVE" --lanes 1 || echo "pass a failed"
bash scripts/profile_counters.sh r04_valu_b "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" --lanes 1 || echo "pass b failed"
bash scripts/profile_counters.sh r04_valu_c "VALUBusy SALUBusy VALUUtilization" --lanes 1 || echo "pass c failed"
bash scripts/profile_counters.sh r04_valu_d "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --lanes 1 || echo "pass d failed"
