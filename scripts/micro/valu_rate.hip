// What does a wave64 vector instruction cost a gfx950 SIMD? (the question under "what bounds enc_cand": DESIGN.md section 3)
//
// Every wave runs a long stream of INDEPENDENT instructions of one kind (8 rotating destination registers, so no instruction
// waits for the one before it; one row runs a dependent chain on purpose), W waves per SIMD on every CU of the chip: one
// workgroup of 256 threads is one wave per SIMD of a CU, W workgroups per CU. Reported: s_memtime ticks per wave-instruction per
// SIMD = the slowest wave's own span / (W x instructions per wave); a tick is one shader cycle at 2.39 GHz (memtime_rate.hip), the
// last column is that span against the host's wall time of the W = 8 launch. (s_memtime counts per XCD: spans of single waves
// are compared, never stamps of different waves.)
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate scripts/micro/valu_rate.hip && /tmp/valu_rate     (profiles/r04_valu_rate.txt)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Op {
    OP_ADD, OP_AND, OP_XOR, OP_ALIGNBIT, OP_BFE, OP_CNDMASK, OP_MOV_DPP, OP_ADD_DPP, OP_LSHL, OP_LSHL64, OP_MUL_LO, OP_MAD_U24, OP_FFBL, OP_BCNT,
    OP_CMP, OP_CMP_SGPR, OP_CND_SGPR, OP_CND_E64_VCC, OP_CND_E32_DIFF, OP_ADDC, OP_CND_AFTER_CMP, OP_CND_IMM, OP_MOV, OP_SUB, OP_OR, OP_LSHR, OP_LSHL_ADD_U64, OP_CMP_U64, OP_MOV64, OP_SAVEEXEC, OP_S_OR64, OP_ADD_DEP, OP_ADD3, OP_LSHL_ADD, OP_MIN, OP_ADD_SDWA, OP_READLANE, OP_MBCNT, OP_FMA, OP_SALU_ADD, OP_MIX_VS, OP_MIX_V2S, OP_N
};
static const char *op_name[OP_N] = {
    "v_add_u32", "v_and_b32", "v_xor_b32", "v_alignbit_b32", "v_bfe_u32", "v_cndmask_b32 (vcc)", "v_mov_b32_dpp wave_shr:1", "v_add_u32_dpp row_shr:1",
    "v_lshlrev_b32", "v_lshlrev_b64", "v_mul_lo_u32", "v_mad_u32_u24", "v_ffbl_b32", "v_bcnt_u32_b32", "v_cmp_eq_u32 (vcc)", "v_cmp_eq_u32 (sgpr pair)", "v_cndmask_b32 (sgpr pair)", "v_cndmask_b32_e64 d, a, d, vcc", "v_cndmask_b32_e32 d, a, b, vcc", "v_addc_co_u32_e32 (vcc in and out)", "v_cmp + 7 v_cndmask_b32 (vcc)", "v_cndmask_b32 0, 1, vcc", "v_mov_b32", "v_sub_u32", "v_or_b32", "v_lshrrev_b32", "v_lshl_add_u64", "v_cmp_ne_u64 (vcc)", "v_mov_b64", "s_and_saveexec_b64 + s_mov exec", "s_or_b64", "v_add_u32 (each waits for the one before)",
    "v_add3_u32", "v_lshl_add_u32", "v_min_u32", "v_add_u32_sdwa", "v_readlane_b32", "v_mbcnt_lo_u32_b32", "v_fma_f32", "s_add_u32 (scalar)",
    "1 v_add_u32 + 1 s_add_u32 (per VALU)", "2 v_add_u32 + 1 s_add_u32 (per VALU)"};

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t iters, uint32_t seed, unsigned long long *span, uint32_t *hwid, uint32_t *sink) {
    uint32_t r0 = seed + threadIdx.x, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, r4 = r0 * 9, r5 = r0 * 11, r6 = r0 * 13, r7 = r0 * 15;
    uint32_t a = seed ^ 0x1234567u, b = threadIdx.x | 1;
    unsigned long long w0 = a, w1 = b, w2 = a + 1, w3 = b + 1, w4 = seed;
    const unsigned long long m64 = 0x5555555555555555ull ^ seed;
    uint32_t s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t it = 0; it < iters; it++) {
#define BODY8(INS) REP8(INS)
        // 64 instructions per iteration, 8 destinations in rotation
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (OP == OP_ADD) asm volatile("v_add_u32 %0, %8, %0\nv_add_u32 %1, %8, %1\nv_add_u32 %2, %8, %2\nv_add_u32 %3, %8, %3\nv_add_u32 %4, %8, %4\nv_add_u32 %5, %8, %5\nv_add_u32 %6, %8, %6\nv_add_u32 %7, %8, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
            if (OP == OP_AND) asm volatile("v_and_b32 %0, %8, %0\nv_and_b32 %1, %8, %1\nv_and_b32 %2, %8, %2\nv_and_b32 %3, %8, %3\nv_and_b32 %4, %8, %4\nv_and_b32 %5, %8, %5\nv_and_b32 %6, %8, %6\nv_and_b32 %7, %8, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
            if (OP == OP_XOR) asm volatile("v_xor_b32 %0, %8, %0\nv_xor_b32 %1, %8, %1\nv_xor_b32 %2, %8, %2\nv_xor_b32 %3, %8, %3\nv_xor_b32 %4, %8, %4\nv_xor_b32 %5, %8, %5\nv_xor_b32 %6, %8, %6\nv_xor_b32 %7, %8, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
            if (OP == OP_ALIGNBIT) asm volatile("v_alignbit_b32 %0, %8, %0, %9\nv_alignbit_b32 %1, %8, %1, %9\nv_alignbit_b32 %2, %8, %2, %9\nv_alignbit_b32 %3, %8, %3, %9\nv_alignbit_b32 %4, %8, %4, %9\nv_alignbit_b32 %5, %8, %5, %9\nv_alignbit_b32 %6, %8, %6, %9\nv_alignbit_b32 %7, %8, %7, %9" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
            if (OP == OP_BFE) asm volatile("v_bfe_u32 %0, %0, %9, %8\nv_bfe_u32 %1, %1, %9, %8\nv_bfe_u32 %2, %2, %9, %8\nv_bfe_u32 %3, %3, %9, %8\nv_bfe_u32 %4, %4, %9, %8\nv_bfe_u32 %5, %5, %9, %8\nv_bfe_u32 %6, %6, %9, %8\nv_bfe_u32 %7, %7, %9, %8" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
            if (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %8, %0, vcc\nv_cndmask_b32 %1, %8, %1, vcc\nv_cndmask_b32 %2, %8, %2, vcc\nv_cndmask_b32 %3, %8, %3, vcc\nv_cndmask_b32 %4, %8, %4, vcc\nv_cndmask_b32 %5, %8, %5, vcc\nv_cndmask_b32 %6, %8, %6, vcc\nv_cndmask_b32 %7, %8, %7, vcc" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a) : "vcc");
            if (OP == OP_MOV_DPP) asm volatile("v_mov_b32_dpp %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
            if (OP == OP_ADD_DPP) asm volatile("v_add_u32_dpp %0, %8, %0 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %1, %8, %1 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %2, %8, %2 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %3, %8, %3 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %4, %8, %4 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %5, %8, %5 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %6, %8, %6 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_u32_dpp %7, %8, %7 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
            if (OP == OP_LSHL) asm volatile("v_lshlrev_b32 %0, %8, %0\nv_lshlrev_b32 %1, %8, %1\nv_lshlrev_b32 %2, %8, %2\nv_lshlrev_b32 %3, %8, %3\nv_lshlrev_b32 %4, %8, %4\nv_lshlrev_b32 %5, %8, %5\nv_lshlrev_b32 %6, %8, %6\nv_lshlrev_b32 %7, %8, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b));
            if (OP == OP_LSHL64) asm volatile("v_lshlrev_b64 %0, %4, %0\nv_lshlrev_b64 %1, %4, %1\nv_lshlrev_b64 %2, %4, %2\nv_lshlrev_b64 %3, %4, %3\nv_lshlrev_b64 %0, %4, %0\nv_lshlrev_b64 %1, %4, %1\nv_lshlrev_b64 %2, %4, %2\nv_lshlrev_b64 %3, %4, %3" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(b));
            if (OP == OP_MUL_LO) asm volatile("v_mul_lo_u32 %0, %8, %0\nv_mul_lo_u32 %1, %8, %1\nv_mul_lo_u32 %2, %8, %2\nv_mul_lo_u32 %3, %8, %3\nv_mul_lo_u32 %4, %8, %4\nv_mul_lo_u32 %5, %8, %5\nv_mul_lo_u32 %6, %8, %6\nv_mul_lo_u32 %7, %8, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b));
            if (OP == OP_MAD_U24) asm volatile("v_mad_u32_u24 %0, %8, %0, %9\nv_mad_u32_u24 %1, %8, %1, %9\nv_mad_u32_u24 %2, %8, %2, %9\nv_mad_u32_u24 %3, %8, %3, %9\nv_mad_u32_u24 %4, %8, %4, %9\nv_mad_u32_u24 %5, %8, %5, %9\nv_mad_u32_u24 %6, %8, %6, %9\nv_mad_u32_u24 %7, %8, %7, %9" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
            if (OP == OP_FFBL) asm volatile("v_ffbl_b32 %0, %0\nv_ffbl_b32 %1, %1\nv_ffbl_b32 %2, %2\nv_ffbl_b32 %3, %3\nv_ffbl_b32 %4, %4\nv_ffbl_b32 %5, %5\nv_ffbl_b32 %6, %6\nv_ffbl_b32 %7, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
            if (OP == OP_BCNT) asm volatile("v_bcnt_u32_b32 %0, %8, %0\nv_bcnt_u32_b32 %1, %8, %1\nv_bcnt_u32_b32 %2, %8, %2\nv_bcnt_u32_b32 %3, %8, %3\nv_bcnt_u32_b32 %4, %8, %4\nv_bcnt_u32_b32 %5, %8, %5\nv_bcnt_u32_b32 %6, %8, %6\nv_bcnt_u32_b32 %7, %8, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
            if (OP == OP_CMP) asm volatile("v_cmp_eq_u32 vcc, %8, %0\nv_cmp_eq_u32 vcc, %8, %1\nv_cmp_eq_u32 vcc, %8, %2\nv_cmp_eq_u32 vcc, %8, %3\nv_cmp_eq_u32 vcc, %8, %4\nv_cmp_eq_u32 vcc, %8, %5\nv_cmp_eq_u32 vcc, %8, %6\nv_cmp_eq_u32 vcc, %8, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a) : "vcc");
            if (OP == OP_CMP_SGPR) asm volatile("v_cmp_eq_u32 s[20:21], %8, %0\nv_cmp_eq_u32 s[22:23], %8, %1\nv_cmp_eq_u32 s[24:25], %8, %2\nv_cmp_eq_u32 s[26:27], %8, %3\nv_cmp_eq_u32 s[20:21], %8, %4\nv_cmp_eq_u32 s[22:23], %8, %5\nv_cmp_eq_u32 s[24:25], %8, %6\nv_cmp_eq_u32 s[26:27], %8, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            if (OP == OP_CND_SGPR) asm volatile("v_cndmask_b32_e64 %0, %8, %0, %9\nv_cndmask_b32_e64 %1, %8, %1, %9\nv_cndmask_b32_e64 %2, %8, %2, %9\nv_cndmask_b32_e64 %3, %8, %3, %9\nv_cndmask_b32_e64 %4, %8, %4, %9\nv_cndmask_b32_e64 %5, %8, %5, %9\nv_cndmask_b32_e64 %6, %8, %6, %9\nv_cndmask_b32_e64 %7, %8, %7, %9" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "s"(m64));
            if (OP == OP_CND_E64_VCC) asm volatile("v_cndmask_b32_e64 %0, %8, %0, vcc\nv_cndmask_b32_e64 %1, %8, %1, vcc\nv_cndmask_b32_e64 %2, %8, %2, vcc\nv_cndmask_b32_e64 %3, %8, %3, vcc\nv_cndmask_b32_e64 %4, %8, %4, vcc\nv_cndmask_b32_e64 %5, %8, %5, vcc\nv_cndmask_b32_e64 %6, %8, %6, vcc\nv_cndmask_b32_e64 %7, %8, %7, vcc" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a) : "vcc");
            if (OP == OP_CND_E32_DIFF) asm volatile("v_cndmask_b32_e32 %0, %8, %9, vcc\nv_cndmask_b32_e32 %1, %8, %9, vcc\nv_cndmask_b32_e32 %2, %8, %9, vcc\nv_cndmask_b32_e32 %3, %8, %9, vcc\nv_cndmask_b32_e32 %4, %8, %9, vcc\nv_cndmask_b32_e32 %5, %8, %9, vcc\nv_cndmask_b32_e32 %6, %8, %9, vcc\nv_cndmask_b32_e32 %7, %8, %9, vcc" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(a), "v"(b) : "vcc");
            if (OP == OP_ADDC) asm volatile("v_addc_co_u32_e32 %0, vcc, %8, %0, vcc\nv_addc_co_u32_e32 %1, vcc, %8, %1, vcc\nv_addc_co_u32_e32 %2, vcc, %8, %2, vcc\nv_addc_co_u32_e32 %3, vcc, %8, %3, vcc\nv_addc_co_u32_e32 %4, vcc, %8, %4, vcc\nv_addc_co_u32_e32 %5, vcc, %8, %5, vcc\nv_addc_co_u32_e32 %6, vcc, %8, %6, vcc\nv_addc_co_u32_e32 %7, vcc, %8, %7, vcc" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a) : "vcc");
            if (OP == OP_CND_AFTER_CMP) asm volatile("v_cmp_gt_u32 vcc, %8, %0\nv_cndmask_b32 %1, %8, %1, vcc\nv_cndmask_b32 %2, %8, %2, vcc\nv_cndmask_b32 %3, %8, %3, vcc\nv_cndmask_b32 %4, %8, %4, vcc\nv_cndmask_b32 %5, %8, %5, vcc\nv_cndmask_b32 %6, %8, %6, vcc\nv_cndmask_b32 %7, %8, %7, vcc" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a) : "vcc");
            if (OP == OP_CND_IMM) asm volatile("v_cndmask_b32 %0, 0, 1, vcc\nv_cndmask_b32 %1, 0, 1, vcc\nv_cndmask_b32 %2, 0, 1, vcc\nv_cndmask_b32 %3, 0, 1, vcc\nv_cndmask_b32 %4, 0, 1, vcc\nv_cndmask_b32 %5, 0, 1, vcc\nv_cndmask_b32 %6, 0, 1, vcc\nv_cndmask_b32 %7, 0, 1, vcc" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : : "vcc");
            if (OP == OP_MOV) asm volatile("v_mov_b32 %0, %8\nv_mov_b32 %1, %8\nv_mov_b32 %2, %8\nv_mov_b32 %3, %8\nv_mov_b32 %4, %8\nv_mov_b32 %5, %8\nv_mov_b32 %6, %8\nv_mov_b32 %7, %8" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(a));
            if (OP == OP_SUB) asm volatile("v_sub_u32 %0, %8, %0\nv_sub_u32 %1, %8, %1\nv_sub_u32 %2, %8, %2\nv_sub_u32 %3, %8, %3\nv_sub_u32 %4, %8, %4\nv_sub_u32 %5, %8, %5\nv_sub_u32 %6, %8, %6\nv_sub_u32 %7, %8, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
            if (OP == OP_OR) asm volatile("v_or_b32 %0, %8, %0\nv_or_b32 %1, %8, %1\nv_or_b32 %2, %8, %2\nv_or_b32 %3, %8, %3\nv_or_b32 %4, %8, %4\nv_or_b32 %5, %8, %5\nv_or_b32 %6, %8, %6\nv_or_b32 %7, %8, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
            if (OP == OP_LSHR) asm volatile("v_lshrrev_b32 %0, %8, %0\nv_lshrrev_b32 %1, %8, %1\nv_lshrrev_b32 %2, %8, %2\nv_lshrrev_b32 %3, %8, %3\nv_lshrrev_b32 %4, %8, %4\nv_lshrrev_b32 %5, %8, %5\nv_lshrrev_b32 %6, %8, %6\nv_lshrrev_b32 %7, %8, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b));
            if (OP == OP_LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 0, %4\nv_lshl_add_u64 %1, %1, 0, %4\nv_lshl_add_u64 %2, %2, 0, %4\nv_lshl_add_u64 %3, %3, 0, %4\nv_lshl_add_u64 %0, %0, 0, %4\nv_lshl_add_u64 %1, %1, 0, %4\nv_lshl_add_u64 %2, %2, 0, %4\nv_lshl_add_u64 %3, %3, 0, %4" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(w4));
            if (OP == OP_CMP_U64) asm volatile("v_cmp_ne_u64 vcc, %0, %4\nv_cmp_ne_u64 vcc, %1, %4\nv_cmp_ne_u64 vcc, %2, %4\nv_cmp_ne_u64 vcc, %3, %4\nv_cmp_ne_u64 vcc, %0, %4\nv_cmp_ne_u64 vcc, %1, %4\nv_cmp_ne_u64 vcc, %2, %4\nv_cmp_ne_u64 vcc, %3, %4" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(w4) : "vcc");
            if (OP == OP_MOV64) asm volatile("v_mov_b64 %0, %4\nv_mov_b64 %1, %4\nv_mov_b64 %2, %4\nv_mov_b64 %3, %4\nv_mov_b64 %0, %4\nv_mov_b64 %1, %4\nv_mov_b64 %2, %4\nv_mov_b64 %3, %4" : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3) : "v"(w4));
            if (OP == OP_SAVEEXEC) asm volatile("s_and_saveexec_b64 s[20:21], %0\ns_mov_b64 exec, s[20:21]\ns_and_saveexec_b64 s[22:23], %0\ns_mov_b64 exec, s[22:23]\ns_and_saveexec_b64 s[20:21], %0\ns_mov_b64 exec, s[20:21]\ns_and_saveexec_b64 s[22:23], %0\ns_mov_b64 exec, s[22:23]" : : "s"(m64) : "s20", "s21", "s22", "s23", "scc");
            if (OP == OP_S_OR64) asm volatile("s_or_b64 s[20:21], s[20:21], %0\ns_or_b64 s[22:23], s[22:23], %0\ns_or_b64 s[24:25], s[24:25], %0\ns_or_b64 s[26:27], s[26:27], %0\ns_or_b64 s[20:21], s[20:21], %0\ns_or_b64 s[22:23], s[22:23], %0\ns_or_b64 s[24:25], s[24:25], %0\ns_or_b64 s[26:27], s[26:27], %0" : : "s"(m64) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
            if (OP == OP_ADD_DEP) asm volatile("v_add_u32 %0, %1, %0\nv_add_u32 %0, %1, %0\nv_add_u32 %0, %1, %0\nv_add_u32 %0, %1, %0\nv_add_u32 %0, %1, %0\nv_add_u32 %0, %1, %0\nv_add_u32 %0, %1, %0\nv_add_u32 %0, %1, %0" : "+v"(r0) : "v"(a));
            if (OP == OP_ADD3) asm volatile("v_add3_u32 %0, %8, %0, %9\nv_add3_u32 %1, %8, %1, %9\nv_add3_u32 %2, %8, %2, %9\nv_add3_u32 %3, %8, %3, %9\nv_add3_u32 %4, %8, %4, %9\nv_add3_u32 %5, %8, %5, %9\nv_add3_u32 %6, %8, %6, %9\nv_add3_u32 %7, %8, %7, %9" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
            if (OP == OP_LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 2, %8\nv_lshl_add_u32 %1, %1, 2, %8\nv_lshl_add_u32 %2, %2, 2, %8\nv_lshl_add_u32 %3, %3, 2, %8\nv_lshl_add_u32 %4, %4, 2, %8\nv_lshl_add_u32 %5, %5, 2, %8\nv_lshl_add_u32 %6, %6, 2, %8\nv_lshl_add_u32 %7, %7, 2, %8" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
            if (OP == OP_MIN) asm volatile("v_min_u32 %0, %8, %0\nv_min_u32 %1, %8, %1\nv_min_u32 %2, %8, %2\nv_min_u32 %3, %8, %3\nv_min_u32 %4, %8, %4\nv_min_u32 %5, %8, %5\nv_min_u32 %6, %8, %6\nv_min_u32 %7, %8, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
            if (OP == OP_ADD_SDWA) asm volatile("v_add_u32_sdwa %0, %8, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\nv_add_u32_sdwa %1, %8, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\nv_add_u32_sdwa %2, %8, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\nv_add_u32_sdwa %3, %8, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\nv_add_u32_sdwa %4, %8, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\nv_add_u32_sdwa %5, %8, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\nv_add_u32_sdwa %6, %8, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\nv_add_u32_sdwa %7, %8, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
            if (OP == OP_READLANE) asm volatile("v_readlane_b32 %0, %4, 3\nv_readlane_b32 %1, %5, 5\nv_readlane_b32 %2, %6, 7\nv_readlane_b32 %3, %7, 9\nv_readlane_b32 %0, %4, 11\nv_readlane_b32 %1, %5, 13\nv_readlane_b32 %2, %6, 15\nv_readlane_b32 %3, %7, 17" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(r0), "v"(r1), "v"(r2), "v"(r3));
            if (OP == OP_MBCNT) asm volatile("v_mbcnt_lo_u32_b32 %0, %8, %0\nv_mbcnt_lo_u32_b32 %1, %8, %1\nv_mbcnt_lo_u32_b32 %2, %8, %2\nv_mbcnt_lo_u32_b32 %3, %8, %3\nv_mbcnt_lo_u32_b32 %4, %8, %4\nv_mbcnt_lo_u32_b32 %5, %8, %5\nv_mbcnt_lo_u32_b32 %6, %8, %6\nv_mbcnt_lo_u32_b32 %7, %8, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
            if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %8, %0, %9\nv_fma_f32 %1, %8, %1, %9\nv_fma_f32 %2, %8, %2, %9\nv_fma_f32 %3, %8, %3, %9\nv_fma_f32 %4, %8, %4, %9\nv_fma_f32 %5, %8, %5, %9\nv_fma_f32 %6, %8, %6, %9\nv_fma_f32 %7, %8, %7, %9" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
            if (OP == OP_SALU_ADD) asm volatile("s_add_u32 %0, %0, %4\ns_add_u32 %1, %1, %4\ns_add_u32 %2, %2, %4\ns_add_u32 %3, %3, %4\ns_add_u32 %0, %0, %4\ns_add_u32 %1, %1, %4\ns_add_u32 %2, %2, %4\ns_add_u32 %3, %3, %4" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(seed) : "scc");
            if (OP == OP_MIX_VS) asm volatile("v_add_u32 %0, %8, %0\ns_add_u32 %9, %9, %13\nv_add_u32 %1, %8, %1\ns_add_u32 %10, %10, %13\nv_add_u32 %2, %8, %2\ns_add_u32 %11, %11, %13\nv_add_u32 %3, %8, %3\ns_add_u32 %12, %12, %13\nv_add_u32 %4, %8, %4\ns_add_u32 %9, %9, %13\nv_add_u32 %5, %8, %5\ns_add_u32 %10, %10, %13\nv_add_u32 %6, %8, %6\ns_add_u32 %11, %11, %13\nv_add_u32 %7, %8, %7\ns_add_u32 %12, %12, %13" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(seed) : "scc");
            if (OP == OP_MIX_V2S) asm volatile("v_add_u32 %0, %8, %0\nv_add_u32 %1, %8, %1\ns_add_u32 %9, %9, %13\nv_add_u32 %2, %8, %2\nv_add_u32 %3, %8, %3\ns_add_u32 %10, %10, %13\nv_add_u32 %4, %8, %4\nv_add_u32 %5, %8, %5\ns_add_u32 %11, %11, %13\nv_add_u32 %6, %8, %6\nv_add_u32 %7, %8, %7\ns_add_u32 %12, %12, %13" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(seed) : "scc");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) {
        span[2 * wave] = t0;
        span[2 * wave + 1] = t1;
        hwid[wave] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
    }
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + (uint32_t)(w0 + w1 + w2 + w3) + s0 + s1 + s2 + s3 == 0x9E3779B9u) sink[0] = 1;   // keep everything alive
}

typedef void (*KernelFn)(uint32_t, uint32_t, unsigned long long *, uint32_t *, uint32_t *);

template <int OP>
struct Table {
    static void fill(KernelFn *t) {
        t[OP] = rate_kernel<OP>;
        Table<OP + 1>::fill(t);
    }
};
template <>
struct Table<OP_N> {
    static void fill(KernelFn *) {}
};

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("# %s, %d CUs, clock %d MHz (prop); instructions per wave and iteration: 64 (MIX rows: 64 VALU + 64 / 32 SALU)\n", prop.gcnArchName, cus, prop.clockRate / 1000);
    printf("# cycles per wave-instruction per SIMD = slowest wave's s_memtime span / (waves per SIMD x instructions per wave)\n");
    KernelFn fn[OP_N];
    Table<0>::fill(fn);
    const int max_waves = cus * 32;
    unsigned long long *d_span;
    uint32_t *d_hwid, *d_sink;
    hipMalloc(&d_span, sizeof(unsigned long long) * 2 * max_waves);
    hipMalloc(&d_hwid, 4 * max_waves);
    hipMalloc(&d_sink, 4);
    std::vector<unsigned long long> span(2 * max_waves);
    std::vector<uint32_t> hw(max_waves);
    const uint32_t iters = argc > 1 ? (uint32_t)atoi(argv[1]) : 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%-40s %10s %10s %10s %10s   (s_memtime ticks per instruction per SIMD at 1 / 2 / 4 / 8 waves per SIMD; ticks per ns of host wall time over the W=8 launch, launch overhead included)\n", "instruction", "W=1", "W=2", "W=4", "W=8");
    for (int op = 0; op < OP_N; op++) {
        printf("%-40s", op_name[op]);
        double ghz = 0;
        bool placed = true;
        for (int wps : {1, 2, 4, 8}) {
            // one workgroup of 256 threads = one wave per SIMD of a CU; wps workgroups per CU
            const int blocks = cus * wps;
            double ms = 0;
            for (int rep = 0; rep < 2; rep++) {   // (first pass warms the instruction cache and the clocks)
                hipDeviceSynchronize();
                const auto c0 = std::chrono::steady_clock::now();
                hipLaunchKernelGGL(fn[op], dim3(blocks), dim3(256), 0, 0, iters, 12345u, d_span, d_hwid, d_sink);
                hipDeviceSynchronize();
                ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c0).count();
            }
            const int waves = blocks * 4;
            hipMemcpy(span.data(), d_span, sizeof(unsigned long long) * 2 * waves, hipMemcpyDeviceToHost);
            hipMemcpy(hw.data(), d_hwid, 4 * waves, hipMemcpyDeviceToHost);
            unsigned long long worst = 0, tmin = ~0ull, tmax = 0;
            std::map<uint32_t, int> per_simd;
            for (int w = 0; w < waves; w++) {
                worst = std::max(worst, span[2 * w + 1] - span[2 * w]);
                tmin = std::min(tmin, span[2 * w]);
                tmax = std::max(tmax, span[2 * w + 1]);
                // HW_ID: simd [5:4], cu [11:8], sh [12], se [15:13] (+ XCC id from its own register: the same CU ids repeat per XCD,
                // so the count per (se, sh, cu, simd) must be wps x 8 XCDs ... counted loosely: max / min over the keys)
                per_simd[hw[w] & 0xFF30u]++;
            }
            int mn = 1 << 30, mx = 0;
            for (auto &kv : per_simd) { mn = std::min(mn, kv.second); mx = std::max(mx, kv.second); }
            if (mn != mx) placed = false;
            const double per = (double)worst / ((double)wps * 64.0 * iters);
            printf(" %10.2f", per);
            ghz = (double)worst / (ms * 1e6);
        }
        printf("   %.2f ticks/ns\n", ghz);
        (void)placed;
    }
    return 0;
}
