// Encode side of the MI355X LZFSE codec: hand-written HIP kernels for gfx950 (wave64).
//
// Match finding lives in encode_match.hip (bucket chains + per-position candidates), the parse in
// encode_parse.hip (speculative segment walks, stitching, block segmentation). Here:
//
//   enc_block_kernel   per bvx2 block: literal gather, histograms, normalize_m1 (weights.rs:
//                      218-278), weight bytes (weight_encoder.rs:23-37), E tables (encoder.rs:
//                      219-240), reverse FSE of literals / LMDs (literals.rs:93-133, lmds.rs:
//                      62-93), header (block.rs:168-196)
//   enc_pack_kernel    stream assembly: blocks back to back + bvx$ (frontend_bytes.rs:50-61)
//   enc_batch_device   host orchestration of one encode (sub-)batch
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "enc_common.h"

struct lzfse_mi_ctx;

namespace lzmi {

// ------------------------------------------------------------------------------------ block encode

constexpr int BLK_THREADS = 256;
constexpr uint32_t CL = 2048;                         // literals per chunk (8 per thread)
constexpr uint32_t CE = 512;                          // LMDs per chunk (2 per thread)
constexpr uint32_t PK_LIT = CL * 10 / 32 + 8;         // <= 10 bits per literal
constexpr uint32_t PK_LMD = CE * 54 / 32 + 8;         // <= 54 bits per LMD

__device__ __forceinline__ uint32_t l_sym_of(uint32_t v) { return v < 16 ? v : 16u + (v >= 20) + (v >= 28) + (v >= 60); }
__device__ __forceinline__ uint32_t m_sym_of(uint32_t v) { return v < 16 ? v : 16u + (v >= 24) + (v >= 56) + (v >= 312); }
// fse/constants.rs:323-353: largest symbol whose base <= v; base(4q) = 4 (2^q - 1)
__device__ __forceinline__ uint32_t d_sym_of(uint32_t v) {
    uint32_t q = (31 - __builtin_clz(v + 4)) - 2;
    uint32_t r = (v - 4u * ((1u << q) - 1u)) >> q;
    return 4 * q + r;
}

// weights.rs:218-278 (normalize_m1) by one wave over a table in LDS (n <= 256, every lane must call): the scaling,
// the total and the first strict maximum are lane-parallel; the common correction (the surplus fits a quarter of the
// largest weight) is one update, the rare trimming loops (weights.rs:246-278) stay serial on lane 0.
__device__ void normalize_m1_wave(uint16_t *w, uint32_t n, uint32_t in_total, uint32_t out_total) {
    if (in_total == 0) return;
    const int lane = e_lane();
    const uint32_t shift = __builtin_clz(out_total);
    const uint32_t multiply = (1u << 31) / in_total;
    const uint32_t round = 1u << (shift - 1);
    uint32_t sum = 0, best = 0;  // best = weight << 16 | (0xFFFF - index): max picks the largest weight, lowest index
    for (uint32_t i = lane; i < n; i += 64) {
        const uint32_t wi = w[i];
        if (wi == 0) continue;
        uint32_t f = (wi * multiply + round) >> shift;
        if (f == 0) f = 1;
        w[i] = (uint16_t)f;
        sum += f;
        const uint32_t key = (f << 16) | (0xFFFFu - i);
        best = key > best ? key : best;
    }
#pragma unroll
    for (int dd = 32; dd > 0; dd >>= 1) {
        sum += __shfl_xor(sum, dd);
        const uint32_t o = __shfl_xor(best, dd);
        best = o > best ? o : best;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane != 0) return;
    const uint32_t max_index = 0xFFFFu - (best & 0xFFFFu);
    int32_t remaining = (int32_t)out_total - (int32_t)sum;
    if (-remaining < (int32_t)w[max_index] / 4) {
        w[max_index] = (uint16_t)((int32_t)w[max_index] + remaining);
    } else {
        uint32_t overflow = (uint32_t)(-remaining);
        for (int s = 3; s >= 0; s--)
            for (uint32_t i = 0; i < n; i++) {
                if (overflow == 0) break;
                uint32_t wi = w[i];
                if (wi == 0) continue;
                uint32_t k = (wi - 1) >> s;
                if (k > overflow) k = overflow;
                w[i] = (uint16_t)(wi - k);
                overflow -= k;
            }
    }
}

// the same by one lane (kept for reference: the restatement closest to weights.rs:218-278)
__device__ void normalize_m1(uint16_t *w, uint32_t n, uint32_t in_total, uint32_t out_total) {
    if (in_total == 0) return;
    uint32_t shift = __builtin_clz(out_total);
    uint32_t multiply = (1u << 31) / in_total;
    uint32_t round = 1u << (shift - 1);
    uint32_t max_weight = 0, max_index = 0;
    int32_t remaining = (int32_t)out_total;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t wi = w[i];
        if (wi == 0) continue;
        uint32_t f = (wi * multiply + round) >> shift;
        if (f == 0) f = 1;
        w[i] = (uint16_t)f;
        remaining -= (int32_t)f;
        if (f > max_weight) { max_weight = f; max_index = i; }
    }
    if (-remaining < (int32_t)w[max_index] / 4) {
        w[max_index] = (uint16_t)((int32_t)w[max_index] + remaining);
    } else {
        uint32_t overflow = (uint32_t)(-remaining);
        for (int s = 3; s >= 0; s--)
            for (uint32_t i = 0; i < n; i++) {
                if (overflow == 0) break;
                uint32_t wi = w[i];
                if (wi == 0) continue;
                uint32_t k = (wi - 1) >> s;
                if (k > overflow) k = overflow;
                w[i] = (uint16_t)(wi - k);
                overflow -= k;
            }
    }
}

// Workgroup barrier that orders LDS traffic only. __syncthreads() also waits for every outstanding global load
// (its workgroup fence drains vmcnt), which would expose the latency of loads issued one pass ahead; the data the
// workgroup exchanges at these barriers lives in LDS, global memory is only read (inputs) or written (staging).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// OR a <= 64-bit little-endian bit field into a u32 LDS bit buffer at bit offset o
__device__ __forceinline__ void or_bits(uint32_t *pk, uint32_t o, uint64_t v) {
    uint32_t wi = o >> 5, sh = o & 31;
    uint32_t w0 = (uint32_t)(v << sh);
    uint64_t rest = sh ? (v >> (32 - sh)) : (v >> 32);
    uint32_t w1 = (uint32_t)rest, w2 = (uint32_t)(rest >> 32);
    if (w0) atomicOr(&pk[wi], w0);
    if (w1) atomicOr(&pk[wi + 1], w1);
    if (w2) atomicOr(&pk[wi + 2], w2);
}

// ---- FSE state chains, speculatively parallel ----
// s' = t_w + (s >> nb), nb = (t_k + s) >> 10 (encoder.rs:191-199) forgets s quickly: nb bits of it per step.
// Lane j of a wave runs steps [8 j, 8 j + 8) of a chain after a warm-up over the CH_WARM steps before them from an
// arbitrary state; it is right iff its start state equals the end state of lane j - 1, which is checked, and the
// first wrong lane is re-run from the right state until every lane checks out (lane 0 starts from the carried
// true state). The result is exactly the serial recurrence.
constexpr uint32_t CH_WARM = 24;
constexpr uint32_t CH_LEN = 512;                    // steps of one chain per chunk = 64 lanes x 8
constexpr uint32_t CH_ROW = CH_LEN + CH_LEN / 32;   // one padding word per 32 steps: lanes 8 steps apart miss each other's bank
__device__ __forceinline__ uint32_t ch_at(uint32_t t) { return t + (t >> 5); }

__device__ __forceinline__ uint32_t ch_step(uint32_t &s, uint32_t e) {
    const int32_t tk = (int32_t)(int16_t)(e & 0xFFFF), tw = (int32_t)(int16_t)(e >> 16);
    const uint32_t nb = (uint32_t)(tk + (int32_t)s) >> 10;
    const uint32_t out = (s & ((1u << nb) - 1u)) | (nb << 16);
    s = (uint32_t)(tw + (int32_t)(s >> nb));
    return out;
}

// row: the chain's CH_ROW entries (E-table words in, state bits | nb << 16 out); cnt <= CH_LEN steps; returns the end
// state. s_any = the chain's initial state value T: a valid state, and (-T & 0xFFFF) is an E entry that leaves every
// state unchanged (nb = 0, t_w = 0), which keeps the step sequence free of branches.
__device__ __forceinline__ uint32_t chain_run(uint32_t *row, uint32_t cnt, uint32_t s_true, uint32_t s_any) {
    const int lane = e_lane();
    const uint32_t t0 = 8u * lane;
    const bool mine = t0 < cnt;
    const uint32_t nst = mine ? (cnt - t0 < 8 ? cnt - t0 : 8) : 0;
    const uint32_t ident = (0u - s_any) & 0xFFFFu;
    uint32_t e[8], o[8], ew[CH_WARM];
#pragma unroll
    for (int u = 0; u < 8; u++) e[u] = (uint32_t)u < nst ? row[ch_at(t0 + u)] : ident;
    // warm-up over steps [t0 - CH_WARM, t0): lanes that would reach below step 0 start there with the true state
#pragma unroll
    for (int u = 0; u < (int)CH_WARM; u++) ew[u] = (mine && t0 + (uint32_t)u >= CH_WARM) ? row[ch_at(t0 + (uint32_t)u - CH_WARM)] : ident;
    uint32_t start = (mine && t0 > CH_WARM) ? s_any : s_true;
#pragma unroll
    for (int u = 0; u < (int)CH_WARM; u++) ch_step(start, ew[u]);
    uint32_t end = start;
#pragma unroll
    for (int u = 0; u < 8; u++) o[u] = ch_step(end, e[u]);
    for (;;) {
        const uint32_t prev_end = __shfl_up(end, 1);
        const uint64_t bad = __ballot(mine && lane > 0 && start != prev_end);
        if (!bad) break;
        const bool fix = lane == __builtin_ctzll(bad);
        uint32_t s2 = prev_end, o2[8];
#pragma unroll
        for (int u = 0; u < 8; u++) o2[u] = ch_step(s2, e[u]);
        if (fix) {
            start = prev_end; end = s2;
#pragma unroll
            for (int u = 0; u < 8; u++) o[u] = o2[u];
        }
    }
    __builtin_amdgcn_wave_barrier();  // warm-up reads of the neighbours are done before anything is overwritten
#pragma unroll
    for (int u = 0; u < 8; u++) if ((uint32_t)u < nst) row[ch_at(t0 + u)] = o[u];
    const int last = cnt ? (int)((cnt - 1) >> 3) : 0;
    return cnt ? e_readlane(end, last) : s_true;
}

static_assert(CE == CH_LEN && CL == 4 * CH_LEN && CE / BLK_THREADS == 2 && CL / BLK_THREADS == 8, "chunk shapes are baked into enc_block_kernel");

template <int DELTA>
__device__ __forceinline__ uint32_t e_dpp_shr(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + DELTA, 0xF, 0xF, true);
}

// forward bit writer of one wave (bits/bit_writer.rs:8-58): lane contributions are OR-combined
struct BitOut {
    uint64_t acc;
    uint32_t fill;
    uint8_t *out;
    uint32_t pos;
};

__device__ __forceinline__ void bo_flush(BitOut &o) {
    uint32_t nbytes = o.fill >> 3;
    if (e_lane() == 0) __builtin_memcpy(o.out + o.pos, &o.acc, 8);  // staging has >= 8 spare bytes
    o.pos += nbytes;
    o.acc = nbytes == 8 ? 0 : o.acc >> (nbytes * 8);
    o.fill &= 7;
}

__device__ __forceinline__ uint32_t bo_finalize(BitOut &o) {
    uint32_t nbytes = (o.fill + 7) >> 3;
    if (e_lane() == 0) __builtin_memcpy(o.out + o.pos, &o.acc, 8);
    o.pos += nbytes;
    return nbytes * 8 - o.fill;
}

#ifndef BLK_WAVES_PER_EU
#define BLK_WAVES_PER_EU 4
#endif
__global__ __launch_bounds__(BLK_THREADS) __attribute__((amdgpu_waves_per_eu(BLK_WAVES_PER_EU, BLK_WAVES_PER_EU))) void enc_block_kernel(const uint8_t *__restrict__ src, const EncStream *__restrict__ streams,
                                                                uint32_t n_streams, const EncStreamOut *__restrict__ outs,
                                                                const uint2 *__restrict__ lmds, EncBlock *__restrict__ blocks,
                                                                const uint32_t *__restrict__ slot_stream, uint8_t *__restrict__ stage,
                                                                uint8_t *__restrict__ lit_scratch, unsigned long long *__restrict__ cyc) {
    __shared__ uint32_t hist[N_WEIGHTS];
    __shared__ uint16_t wts[N_WEIGHTS];
    __shared__ uint32_t etab[N_WEIGHTS];      // t_k (low 16) | t_w (high 16), encoder.rs:184-188
    __shared__ uint32_t wbits[N_WEIGHTS / 4 * 4 + 200];  // weight payload words (<= 630 bytes)
    __shared__ uint32_t scan_sh[3 * (BLK_THREADS / 64) + 2];
    __shared__ uint32_t sh_misc[16];
    __shared__ uint32_t cl[4 * CH_ROW];    // literal chunk, one row per state chain: E entry -> (state bits | nb << 16)
    __shared__ uint32_t ce[3 * CH_ROW];    // LMD chunk: rows D, M, L
    __shared__ uint32_t pk_lit[PK_LIT];    // bit buffers of the current chunk
    __shared__ uint32_t pk_lmd[PK_LMD];

    const uint32_t slot = blockIdx.x;
    const uint32_t si = slot_stream[slot];
    if (si >= n_streams) return;
    const EncStream st = streams[si];
    const uint32_t bi = slot - st.blk_base;
    const EncStreamOut so = outs[si];
    if (so.status || bi >= so.n_blocks) return;
    EncBlock blk = blocks[slot];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint8_t *s = src + st.src_off;
    const uint2 *bl = lmds + blk.lmd_start;
    uint8_t *sg = stage + blk.stage_off;
    // The block's literal bytes (<= 40 000 + pad) live in global scratch, not in LDS: they are written once, read twice
    // in order, and 40 KB of LDS per block would leave room for two blocks per CU. The scratch is the stream's part of
    // the candidate-record array, which no stage reads any more (4 bytes per position; a block needs at most its own
    // raw bytes + 4 of padding here, blocks are kept 32 bytes apart).
    uint8_t *lit = lit_scratch + st.pos_base * 4 + blk.src_start + 32 * bi;

    // diagnostics (LZFSE_MI_BLOCK_STATS): cycles per phase, summed over blocks
    uint64_t tq = cyc ? __builtin_amdgcn_s_memtime() : 0;
    auto lap = [&](int slot) {
        if (cyc && tid == 0) { const uint64_t t2 = __builtin_amdgcn_s_memtime(); atomicAdd(&cyc[slot], (unsigned long long)(t2 - tq)); tq = t2; }
    };
    for (uint32_t i = tid; i < N_WEIGHTS; i += BLK_THREADS) hist[i] = 0;
    for (uint32_t i = tid; i < sizeof(wbits) / 4; i += BLK_THREADS) wbits[i] = 0;
    __syncthreads();

    // ---- literal gather + LMD symbol histograms ----
    // Two passes in flight (round 4): a pass's source addresses come out of a block scan, and its loads used to be waited for
    // before the next pass's scan began -- one memory round trip per 256 LMDs with nothing under it. Now pass g + 1 is scanned and
    // its loads are issued (prep) before pass g's bytes are stored (fin).
    {
        uint32_t run_lit = 0, run_src = blk.src_start;
        uint2 r_next = (uint32_t)tid < blk.n_lmd ? bl[tid] : make_uint2(0, 0);  // records are fetched one pass ahead
        struct Pass {
            uint64_t w0, w1, w2;
            const uint8_t *ls;
            uint8_t *ld;
            uint32_t l;
            bool valid, wide, l_long;
        };
        auto prep = [&](uint32_t g0) -> Pass {
            Pass P;
            const uint32_t idx = g0 + tid;
            P.valid = idx < blk.n_lmd;
            const uint2 r = r_next;
            r_next = idx + BLK_THREADS < blk.n_lmd ? bl[idx + BLK_THREADS] : make_uint2(0, 0);
            const uint32_t l = r.x & 0xFFFF, m = r.x >> 16, d = r.y;
            // block exclusive scan of l and l + m
            uint32_t il = wave_incl_sum(l), is = wave_incl_sum(l + m);
            if (lane == 63) { scan_sh[wave] = il; scan_sh[BLK_THREADS / 64 + wave] = is; }
            lds_barrier();
            uint32_t ol = 0, os = 0, tl = 0, ts = 0;
            for (int wv = 0; wv < BLK_THREADS / 64; wv++) {
                uint32_t a = scan_sh[wv], b2 = scan_sh[BLK_THREADS / 64 + wv];
                if (wv < wave) { ol += a; os += b2; }
                tl += a; ts += b2;
            }
            const uint32_t ex_l = ol + il - l, ex_s = os + is - (l + m);
            P.ls = s + run_src + ex_s;
            P.ld = lit + run_lit + ex_l;
            P.l = l;
            P.l_long = P.valid && l > 24;
            // three 8-byte loads in flight (the run is followed by >= 4 match or literal bytes of the same stream, except at
            // its very end)
            P.wide = P.valid && !P.l_long && l != 0 && run_src + ex_s + 24 <= st.n;
            P.w0 = 0; P.w1 = 0; P.w2 = 0;
            if (P.wide) { P.w0 = ld_u64(P.ls); P.w1 = l > 8 ? ld_u64(P.ls + 8) : 0; P.w2 = l > 16 ? ld_u64(P.ls + 16) : 0; }
            if (P.valid) {
                atomicAdd(&hist[l_sym_of(l)], 1u);
                atomicAdd(&hist[20 + m_sym_of(m)], 1u);
                atomicAdd(&hist[40 + d_sym_of(d)], 1u);
            }
            run_lit += tl; run_src += ts;
            lds_barrier();
            return P;
        };
        auto fin = [&](const Pass &P) {
            const uint32_t l = P.l;
            if (P.valid && !P.l_long && l) {
                if (P.wide) {
                    // exactly l bytes in at most five stores of 8 / 8 / 4 / 2 / 1 bytes at byte alignment (a byte loop costs the
                    // wave its longest run in iterations)
                    const uint64_t w0 = P.w0, w1 = P.w1, w2 = P.w2;
                    uint64_t cur = w0;
                    uint8_t *q = P.ld;
                    if (l >= 8) { __builtin_memcpy(q, &w0, 8); q += 8; cur = w1; }
                    if (l >= 16) { __builtin_memcpy(q, &w1, 8); q += 8; cur = w2; }
                    if (l == 24) { __builtin_memcpy(q, &w2, 8); }
                    else {
                        if (l & 4) { const uint32_t c4 = (uint32_t)cur; __builtin_memcpy(q, &c4, 4); q += 4; cur >>= 32; }
                        if (l & 2) { const uint16_t c2 = (uint16_t)cur; __builtin_memcpy(q, &c2, 2); q += 2; cur >>= 16; }
                        if (l & 1) *q = (uint8_t)cur;
                    }
                } else {
                    for (uint32_t k = 0; k < l; k++) P.ld[k] = P.ls[k];
                }
            }
            // long runs: the whole wave copies one lane's run at a time
            uint64_t ql = __ballot(P.l_long);
            while (ql) {
                const int L = __builtin_ctzll(ql);
                ql &= ql - 1;
                const uint64_t qs = ((uint64_t)e_readlane((uint32_t)((uintptr_t)P.ls >> 32), L) << 32) | e_readlane((uint32_t)(uintptr_t)P.ls, L);
                const uint64_t qd = ((uint64_t)e_readlane((uint32_t)((uintptr_t)P.ld >> 32), L) << 32) | e_readlane((uint32_t)(uintptr_t)P.ld, L);
                const uint32_t q_n = e_readlane(l, L);
                const uint8_t *q_src = (const uint8_t *)(uintptr_t)qs;
                uint8_t *q_dst = (uint8_t *)(uintptr_t)qd;
                for (uint32_t k = lane; k < q_n; k += 64) q_dst[k] = q_src[k];
            }
        };
        if (blk.n_lmd) {
            Pass P0 = prep(0);
            for (uint32_t g0 = 0; g0 < blk.n_lmd; g0 += BLK_THREADS) {
                const bool more = g0 + BLK_THREADS < blk.n_lmd;
                Pass P1 = P0;
                if (more) P1 = prep(g0 + BLK_THREADS);
                fin(P0);
                P0 = P1;
            }
        }
    }
    __syncthreads();   // (the literal bytes are global stores: drained and visible to the workgroup)
    lap(0);
    // literals.rs:136-145 pad with literals[0]; weights.rs:56-64 literal histogram (unpadded)
    const uint32_t n_lit = blk.n_lit, n4 = (n_lit + 3) / 4 * 4;
    if (tid < 4) lit[n_lit + tid] = n_lit ? lit[0] : 0;
    {
        // 8 bytes per thread and step (the scratch has slack after the pad; bytes beyond n_lit are not counted)
        for (uint32_t i = tid * 8; i < n_lit; i += BLK_THREADS * 8) {
            const uint64_t w = ld_u64(lit + i);
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (i + k < n_lit) atomicAdd(&hist[104 + (uint32_t)((w >> (8 * k)) & 0xFF)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < N_WEIGHTS; i += BLK_THREADS) wts[i] = (uint16_t)hist[i];
    __syncthreads();
    lap(1);
    // ---- normalize (one lane per table) ----
    if (wave == 0 && blk.n_lmd) normalize_m1_wave(wts, 20, blk.n_lmd, L_STATES);
    if (wave == 1 && blk.n_lmd) normalize_m1_wave(wts + 20, 20, blk.n_lmd, M_STATES);
    if (wave == 2 && blk.n_lmd) normalize_m1_wave(wts + 40, 64, blk.n_lmd, D_STATES);
    if (wave == 3 && n_lit) normalize_m1_wave(wts + 104, 256, n_lit, U_STATES);
    __syncthreads();
    lap(2);
    // ---- weight payload (weight_encoder.rs:23-37, weights.rs:139-163) + E tables (encoder.rs:219-240) ----
    if (wave == 0) {
        uint32_t carry_bits = 0;
        for (uint32_t g0 = 0; g0 < N_WEIGHTS; g0 += 64) {
            uint32_t k = g0 + lane;
            uint32_t wv = k < N_WEIGHTS ? wts[k] : 0, code = 0, nb = 0;
            if (k < N_WEIGHTS) {
                switch (wv) {
                case 0: code = 0; nb = 2; break;
                case 1: code = 2; nb = 2; break;
                case 2: code = 1; nb = 3; break;
                case 3: code = 5; nb = 3; break;
                case 4: code = 3; nb = 5; break;
                case 5: code = 11; nb = 5; break;
                case 6: code = 19; nb = 5; break;
                case 7: code = 27; nb = 5; break;
                default:
                    if (wv < 24) { code = ((wv - 8) << 4) + 7; nb = 8; }
                    else { code = ((wv - 24) << 4) + 15; nb = 14; }
                }
            }
            uint32_t inc = nb;
            inc = wave_incl_sum(inc);
            uint32_t bitpos = carry_bits + inc - nb;
            if (nb) {
                uint64_t v64 = (uint64_t)code << (bitpos & 31);
                atomicOr(&wbits[bitpos >> 5], (uint32_t)v64);
                if (v64 >> 32) atomicOr(&wbits[(bitpos >> 5) + 1], (uint32_t)(v64 >> 32));
            }
            carry_bits += e_readlane(inc, 63);
        }
        if (lane == 0) sh_misc[0] = (carry_bits + 7) / 8;  // n_weight_payload_bytes
    }
    if (wave == 1 || wave == 2 || wave == 3) {
        // E tables: wave 1 -> U (256), wave 2 -> D (64), wave 3 -> L and M (20 each)
        if (wave == 1) {
            uint32_t w0 = wts[104 + 4 * lane], w1 = wts[105 + 4 * lane], w2 = wts[106 + 4 * lane], w3 = wts[107 + 4 * lane];
            uint32_t s4 = w0 + w1 + w2 + w3, inc = s4;
            inc = wave_incl_sum(inc);
            uint32_t tot = inc - s4;
            uint32_t ws[4] = {w0, w1, w2, w3};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t wv = ws[q];
                int32_t tk, tw;
                if (wv == 0) { tk = -(int32_t)U_STATES; tw = 0; }
                else { int k = __builtin_clz(wv) - 21; tk = 1024 * k - (int32_t)(wv << k); tw = (int32_t)U_STATES + (int32_t)tot - (int32_t)wv; }
                etab[104 + 4 * lane + q] = ((uint32_t)tk & 0xFFFF) | ((uint32_t)tw << 16);
                tot += wv;
            }
        } else if (wave == 2) {
            uint32_t wv = wts[40 + lane], inc = wv;
            inc = wave_incl_sum(inc);
            uint32_t tot = inc - wv;
            int32_t tk, tw;
            if (wv == 0) { tk = -(int32_t)D_STATES; tw = 0; }
            else { int k = __builtin_clz(wv) - 23; tk = 1024 * k - (int32_t)(wv << k); tw = (int32_t)D_STATES + (int32_t)tot - (int32_t)wv; }
            etab[40 + lane] = ((uint32_t)tk & 0xFFFF) | ((uint32_t)tw << 16);
        } else {
            // lanes 0..19 -> L, lanes 32..51 -> M
            bool isl = lane < 20, ism = lane >= 32 && lane < 52;
            uint32_t wv = isl ? wts[lane] : (ism ? wts[20 + lane - 32] : 0), inc = wv;
#pragma unroll
            for (int dd = 1; dd < 32; dd <<= 1) { uint32_t a = __shfl_up(inc, dd); if ((lane & 31) >= dd) inc += a; }
            uint32_t tot = inc - wv;
            if (isl || ism) {
                int32_t tk, tw;
                if (wv == 0) { tk = -64; tw = 0; }
                else { int k = __builtin_clz(wv) - 25; tk = 1024 * k - (int32_t)(wv << k); tw = 64 + (int32_t)tot - (int32_t)wv; }
                etab[isl ? lane : 20 + lane - 32] = ((uint32_t)tk & 0xFFFF) | ((uint32_t)tw << 16);
            }
        }
    }
    __syncthreads();
    const uint32_t n_wbytes = sh_misc[0];
    // weight bytes -> staging after the 32-byte header
    for (uint32_t i = tid; i < n_wbytes; i += BLK_THREADS) sg[V2_HEADER_SIZE + i] = (uint8_t)(wbits[i >> 2] >> (8 * (i & 3)));

    lap(3);
    // ---- the two reverse FSE streams (literals.rs:93-133, lmds.rs:62-93) ----
    // Only the state recurrence s' = t_w + (s >> nb), nb = (t_k + s) >> 10 (encoder.rs:191-199) is
    // serial. Per chunk: (A0) all threads look up the E-table entry of every symbol, (A) one lane
    // per state chain runs the recurrence and leaves (state bits | nb << 16) in place, (B) all
    // threads prefix-sum the field widths and OR the fields into an LDS bit buffer whose
    // completed words are stored coalesced.
    {
        const uint32_t n_lmd = blk.n_lmd;
        uint32_t *lit_words = (uint32_t *)(sg + stage_lit_off());
        uint8_t *lmd_base = sg + stage_lmd_off(n_lit);
        uint32_t *lmd_words = (uint32_t *)(lmd_base + 8);
        if (tid < 2) ((uint32_t *)lmd_base)[tid] = 0;  // 8-byte pad (lmds.rs:67-69)
        const uint32_t it_lit = (n4 + CL - 1) / CL, it_lmd = (n_lmd + CE - 1) / CE;
        const uint32_t n_it = it_lit > it_lmd ? it_lit : it_lmd;
        uint32_t lit_bits_done = 0, lmd_bits_done = 0;  // bits already emitted (uniform)
        // chain states (wave-uniform): wave w carries literal state (3 - w) and, for w < 3, the D, M, L state
        uint32_t lstate = U_STATES, mstate = wave == 0 ? D_STATES : 64u;
        for (uint32_t i = tid; i < PK_LIT; i += BLK_THREADS) pk_lit[i] = 0;
        for (uint32_t i = tid; i < PK_LMD; i += BLK_THREADS) pk_lmd[i] = 0;
        __syncthreads();
        uint2 mnext[CE / BLK_THREADS];  // LMD records of the next chunk, fetched one pass ahead
#pragma unroll
        for (int u = 0; u < CE / BLK_THREADS; u++) {
            const uint32_t k = tid + u * BLK_THREADS;
            mnext[u] = k < n_lmd ? bl[n_lmd - 1 - k] : make_uint2(0, 0);
        }
        // The literals of a chunk, 8 consecutive emissions per thread (emission k is literal n4 - 1 - k: one 8-byte load,
        // emission u of the thread in byte 7 - u), are also fetched one pass ahead (they live in global scratch).
        auto lit_fetch = [&](uint32_t le0_) -> uint64_t {
            const uint32_t k0 = le0_ + (uint32_t)tid * 8;   // first emission of this thread in the chunk
            if (k0 >= n4) return 0ull;
            if (n4 - k0 >= 8) return ld_u64(lit + (n4 - 8 - k0));
            return (uint64_t)ld_u32(lit) << 32;              // 4 left (n4 is a multiple of 4): literals 3, 2, 1, 0
        };
        uint64_t lnext = lit_fetch(0);
#pragma unroll
        for (int u = 0; u < CE / BLK_THREADS; u++) asm volatile("" ::"v"(mnext[u].x), "v"(mnext[u].y));  // see below
        asm volatile("" ::"v"((uint32_t)lnext), "v"((uint32_t)(lnext >> 32)));
        for (uint32_t it = 0; it < n_it; it++) {
            const uint32_t le0 = it * CL, me0 = it * CE;
            const uint32_t lcnt = le0 < n4 ? (n4 - le0 < CL ? n4 - le0 : CL) : 0;
            const uint32_t mcnt = me0 < n_lmd ? (n_lmd - me0 < CE ? n_lmd - me0 : CE) : 0;
            // ---- A0 ----
            {
                const uint64_t lcur = lnext;
                lnext = lit_fetch(le0 + CL);
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t k = (uint32_t)tid * 8 + u;
                    if (k < lcnt) cl[(u & 3) * CH_ROW + ch_at(2 * (uint32_t)tid + (u >> 2))] = etab[104 + (uint32_t)((lcur >> (8 * (7 - u))) & 0xFF)];
                }
            }
            uint2 mrec[CE / BLK_THREADS];
#pragma unroll
            for (int u = 0; u < CE / BLK_THREADS; u++) {
                uint32_t k = tid + u * BLK_THREADS;
                mrec[u] = mnext[u];
                mnext[u] = me0 + CE + k < n_lmd ? bl[n_lmd - 1 - (me0 + CE + k)] : make_uint2(0, 0);
                if (k < mcnt) {
                    uint2 r = mrec[u];
                    ce[0 * CH_ROW + ch_at(k)] = etab[40 + d_sym_of(r.y)];
                    ce[1 * CH_ROW + ch_at(k)] = etab[20 + m_sym_of(r.x >> 16)];
                    ce[2 * CH_ROW + ch_at(k)] = etab[l_sym_of(r.x & 0xFFFF)];
                }
            }
            lds_barrier();
            lap(4);
            // ---- A: state chains: wave w runs literal chain w (emissions k = w mod 4), then waves 0..2 run D, M, L ----
            lstate = chain_run(cl + wave * CH_ROW, (lcnt + 3 - wave) >> 2, lstate, U_STATES);
            if (wave < 3) mstate = chain_run(ce + wave * CH_ROW, mcnt, mstate, wave == 0 ? D_STATES : 64u);
            lds_barrier();
            lap(5);
            // ---- B: widths, prefix sums, bit packing ----
            // literals: thread owns 8 consecutive emissions
            uint64_t lv0 = 0, lv1 = 0;  // up to 80 bits
            uint32_t lw = 0;
            {
                uint32_t k0 = tid * (CL / BLK_THREADS);
#pragma unroll
                for (int u = 0; u < (int)(CL / BLK_THREADS); u++) {
                    uint32_t k = k0 + u;
                    if (k < lcnt) {
                        uint32_t e = cl[(k & 3) * CH_ROW + ch_at(k >> 2)];
                        uint32_t nb = e >> 16;
                        uint64_t v = e & 0xFFFF;
                        if (lw < 64) { lv0 |= v << lw; if (lw + nb > 64) lv1 |= v >> (64 - lw); }
                        else lv1 |= v << (lw - 64);
                        lw += nb;
                    }
                }
            }
            // LMDs: thread owns emissions tid and tid + 256; field order D, M, L (extra bits, then state bits)
            uint64_t mv[CE / BLK_THREADS];
            uint32_t mw[CE / BLK_THREADS];
#pragma unroll
            for (int u = 0; u < CE / BLK_THREADS; u++) {
                uint32_t k = tid + u * BLK_THREADS;
                mv[u] = 0; mw[u] = 0;
                if (k < mcnt) {
                    uint2 r = mrec[u];
                    uint32_t vd = r.y, vm = r.x >> 16, vl = r.x & 0xFFFF;
                    uint32_t sd = d_sym_of(vd), sm = m_sym_of(vm), sl = l_sym_of(vl);
                    uint32_t ed = ce[ch_at(k)], em = ce[CH_ROW + ch_at(k)], el = ce[2 * CH_ROW + ch_at(k)];
                    uint32_t nxd = d_extra_bits(sd), nxm = m_extra_bits(sm), nxl = l_extra_bits(sl);
                    uint64_t fd = (uint64_t)(vd - d_base_value(sd)) | ((uint64_t)(ed & 0xFFFF) << nxd);
                    uint32_t wd = nxd + (ed >> 16);
                    uint64_t fm = (uint64_t)(vm - m_base_value(sm)) | ((uint64_t)(em & 0xFFFF) << nxm);
                    uint32_t wm = nxm + (em >> 16);
                    uint64_t fl = (uint64_t)(vl - l_base_value(sl)) | ((uint64_t)(el & 0xFFFF) << nxl);
                    uint32_t wl = nxl + (el >> 16);
                    mv[u] = fd | (fm << wd) | (fl << (wd + wm));
                    mw[u] = wd + wm + wl;
                }
            }
            // block exclusive scans: literal widths (per thread), LMD widths (per thread, per slice u)
            uint32_t a = lw, b2 = mw[0], c2 = CE / BLK_THREADS > 1 ? mw[CE / BLK_THREADS - 1] : 0;
            uint32_t ia = wave_incl_sum(a), ib = wave_incl_sum(b2), ic = wave_incl_sum(c2);
            if (lane == 63) { scan_sh[wave] = ia; scan_sh[4 + wave] = ib; scan_sh[8 + wave] = ic; }
            lds_barrier();
            uint32_t oa = 0, ob = 0, oc = 0, ta = 0, tb2 = 0, tc = 0;
            for (int wv = 0; wv < BLK_THREADS / 64; wv++) {
                uint32_t x = scan_sh[wv], y = scan_sh[4 + wv], z = scan_sh[8 + wv];
                if (wv < wave) { oa += x; ob += y; oc += z; }
                ta += x; tb2 += y; tc += z;
            }
            const uint32_t ex_a = oa + ia - a, ex_b = ob + ib - b2, ex_c = tb2 + oc + ic - c2;
            // OR fields into the LDS bit buffers (word 0 starts at the last incomplete output word)
            {
                uint32_t o = (lit_bits_done & 31) + ex_a;
                if (lw) {
                    or_bits(pk_lit, o, lv0);
                    if (lw > 64) or_bits(pk_lit, o + 64, lv1);
                }
#pragma unroll
                for (int u = 0; u < (int)(CE / BLK_THREADS); u++)
                    if (mw[u]) or_bits(pk_lmd, (lmd_bits_done & 31) + (u == 0 ? ex_b : ex_c), mv[u]);
            }
            lds_barrier();
            // The records fetched for the next pass are touched here, while nothing younger is in flight: at the
            // loop's back edge the compiler could only wait for everything, the word stores below included.
#pragma unroll
            for (int u = 0; u < CE / BLK_THREADS; u++) asm volatile("" ::"v"(mnext[u].x), "v"(mnext[u].y));
            asm volatile("" ::"v"((uint32_t)lnext), "v"((uint32_t)(lnext >> 32)));
            // store completed words, carry the incomplete one to word 0
            {
                uint32_t have = (lit_bits_done & 31) + ta, full = have >> 5;
                uint32_t w0 = lit_bits_done >> 5;
                for (uint32_t k = tid; k < full; k += BLK_THREADS) lit_words[w0 + k] = pk_lit[k];
                uint32_t carry = pk_lit[full];
                uint32_t have_m = (lmd_bits_done & 31) + tb2 + tc, full_m = have_m >> 5;
                uint32_t m0 = lmd_bits_done >> 5;
                for (uint32_t k = tid; k < full_m; k += BLK_THREADS) lmd_words[m0 + k] = pk_lmd[k];
                uint32_t carry_m = pk_lmd[full_m];
                lds_barrier();
                for (uint32_t k = tid; k <= full + 3 && k < PK_LIT; k += BLK_THREADS) pk_lit[k] = k == 0 ? carry : 0;
                for (uint32_t k = tid; k <= full_m + 3 && k < PK_LMD; k += BLK_THREADS) pk_lmd[k] = k == 0 ? carry_m : 0;
                lit_bits_done += ta;
                lmd_bits_done += tb2 + tc;
            }
            lds_barrier();
            lap(6);
        }
        // finalize (bit_writer.rs:46-57): the last partial word, unused bits of the last byte
        if (tid == 0) {
            lit_words[lit_bits_done >> 5] = pk_lit[0];
            lmd_words[lmd_bits_done >> 5] = pk_lmd[0];
            uint32_t lb = (lit_bits_done + 7) >> 3, mb = (lmd_bits_done + 7) >> 3;
            sh_misc[1] = lb;                           // literal payload bytes
            sh_misc[2] = lb * 8 - lit_bits_done;       // literal bits
            sh_misc[7] = mb + 8;                       // lmd payload bytes incl. pad (lmds.rs:67-69,91)
            sh_misc[8] = mb * 8 - lmd_bits_done;
        }
        if (lane == 0) sh_misc[3 + (3 - wave)] = lstate - U_STATES;   // chain r carries state 3 - r
        if (lane == 0 && wave < 3) sh_misc[wave == 0 ? 11 : (wave == 1 ? 10 : 9)] = mstate - (wave == 0 ? D_STATES : 64u);
    }
    __syncthreads();
    // ---- header (block.rs:168-196) ----
    if (tid == 0) {
        uint32_t lit_payload = sh_misc[1], lmd_payload = sh_misc[7];
        uint64_t p;
        uint32_t magic = MAGIC_VX2, n_raw = blk.n_lit + blk.n_match;
        __builtin_memcpy(sg, &magic, 4);
        __builtin_memcpy(sg + 4, &n_raw, 4);
        p = (uint64_t)n4 | ((uint64_t)lit_payload << 20) | ((uint64_t)blk.n_lmd << 40) | ((uint64_t)(7 - sh_misc[2]) << 60);
        __builtin_memcpy(sg + 8, &p, 8);
        p = (uint64_t)sh_misc[3] | ((uint64_t)sh_misc[4] << 10) | ((uint64_t)sh_misc[5] << 20) | ((uint64_t)sh_misc[6] << 30) |
            ((uint64_t)lmd_payload << 40) | ((uint64_t)(7 - sh_misc[8]) << 60);
        __builtin_memcpy(sg + 16, &p, 8);
        p = (uint64_t)(V2_HEADER_SIZE + n_wbytes) | ((uint64_t)sh_misc[9] << 32) | ((uint64_t)sh_misc[10] << 42) |
            ((uint64_t)sh_misc[11] << 52);
        __builtin_memcpy(sg + 24, &p, 8);
        blocks[slot].hdr_len = V2_HEADER_SIZE + n_wbytes;
        blocks[slot].lit_len = lit_payload;
        blocks[slot].lmd_len = lmd_payload;
    }
}

// ------------------------------------------------------------------------------------ pack

// One workgroup per block slot: the block's three staged segments go to their final position (sum of
// the sizes of the stream's earlier blocks); the last block appends bvx$ and reports the length.
__global__ __launch_bounds__(256) void enc_pack_kernel(const EncStream *__restrict__ streams, const uint32_t *__restrict__ slot_stream,
                                                       const EncBlock *__restrict__ blocks, const uint8_t *__restrict__ stage,
                                                       uint8_t *__restrict__ dst, EncStreamOut *__restrict__ outs) {
    __shared__ unsigned long long sh_sum[4];
    const uint32_t slot = blockIdx.x;
    const uint32_t si = slot_stream[slot];
    const EncStream st = streams[si];
    const uint32_t bi = slot - st.blk_base;
    const EncStreamOut so = outs[si];
    if (so.status || bi >= so.n_blocks) return;
    const int tid = threadIdx.x;
    unsigned long long part = 0;
    for (uint32_t b = tid; b < bi; b += 256) {
        const EncBlock e = blocks[st.blk_base + b];
        part += (unsigned long long)e.hdr_len + e.lit_len + e.lmd_len;
    }
    for (int d2 = 32; d2 > 0; d2 >>= 1) part += __shfl_down(part, d2);
    if ((tid & 63) == 0) sh_sum[tid >> 6] = part;
    __syncthreads();
    const uint64_t pos0 = sh_sum[0] + sh_sum[1] + sh_sum[2] + sh_sum[3];
    const EncBlock blk = blocks[slot];
    const uint64_t need = (uint64_t)blk.hdr_len + blk.lit_len + blk.lmd_len;
    const bool last = bi + 1 == so.n_blocks;
    const bool fits = pos0 + need + 4 <= st.dst_cap;
    if (fits) {
        uint8_t *d = dst + st.dst_off + pos0;
        const uint8_t *sg = stage + blk.stage_off;
        for (uint32_t i = tid; i < blk.hdr_len; i += 256) d[i] = sg[i];
        d += blk.hdr_len;
        const uint8_t *ls = sg + stage_lit_off();
        for (uint32_t i = tid; i < blk.lit_len; i += 256) d[i] = ls[i];
        d += blk.lit_len;
        const uint8_t *ms = sg + stage_lmd_off(blk.n_lit);
        for (uint32_t i = tid; i < blk.lmd_len; i += 256) d[i] = ms[i];
        if (last && tid < 4) d[blk.lmd_len + tid] = (uint8_t)(MAGIC_EOS >> (8 * tid));
    }
    if (last && tid == 0) {
        // earlier blocks fit whenever the last one does (positions are increasing)
        EncStreamOut o = so;
        o.status = fits ? 0 : LZFSE_MI_BUFFER_OVERFLOW;
        o.out_len = fits ? pos0 + need + 4 : 0;
        outs[si] = o;
    }
}

// ------------------------------------------------------------------------------------ host side

enum { EB_STREAMS, EB_TILES, EB_PREV, EB_SUMMARY, EB_REC, EB_LMDS, EB_BLOCKS, EB_OUTS, EB_STAGE, EB_SLOTS,
       EB_BITMAP, EB_SEGS, EB_LOGS, EB_HDRS, EB_RANGES, EB_GAPS, EB_MATCHES, EB_PC, EB_PL, EB_RSLOTS, EB_SYNC, EB_RSUM, EB_DBG, EB_SEGFLAG, EB_GSTATE, EB_CUT, EB_CTILES, EB_N };
static_assert(EB_N <= 32, "EncScratch slots");

static bool eb_ensure(EncScratch &s, int i, size_t n) {
    if (n <= s.caps[i]) return true;
    if (s.bufs[i]) (void)hipFree(s.bufs[i]);
    s.bufs[i] = nullptr;
    s.caps[i] = 0;
    size_t nc = n + n / 8 + 4096;
    if (hipMalloc(&s.bufs[i], nc) != hipSuccess) return false;
    s.caps[i] = nc;
    return true;
}

void enc_scratch_release(EncScratch &s) {
    for (int i = 0; i < 32; i++) {
        if (s.bufs[i]) (void)hipFree(s.bufs[i]);
        s.bufs[i] = nullptr;
        s.caps[i] = 0;
    }
    for (PinVec &h : s.host) h.release();
}
enum { EH_STREAMS, EH_TILES, EH_CTILES, EH_SLOTS, EH_SEGS, EH_RSLOTS, EH_OUTS, EH_N };
static_assert(EH_N <= 8, "EncScratch::host");

#define E_TRY(x) do { if ((x) != hipSuccess) return LZFSE_MI_IO; } while (0)

int enc_batch_device(lzfse_mi_ctx *c, uint32_t count, const uint8_t *d_src, const uint64_t *src_off,
                     const uint64_t *src_len, uint8_t *d_dst, const uint64_t *dst_off, const uint64_t *dst_cap,
                     uint64_t *out_lens, int *statuses) {
    hipStream_t stq = ctx_stream(c);
    EncScratch &S = ctx_enc(c);
#ifdef LZFSE_MI_DIAG
    // (LZFSE_MI_OPT_DIAG_STATS & 4: where the host's time goes in this call -- scripts/stall_probe.py)
    const bool trace_host = (ctx_diag_stats(c) & 4) != 0;
    std::chrono::steady_clock::time_point tp[6];
    auto mark = [&](int k) { if (trace_host) tp[k] = std::chrono::steady_clock::now(); };
#else
    auto mark = [](int) {};
#endif
    mark(0);
    // whatever path leaves this function, the next lane of a split call must not be left waiting
    struct GateRelease {
        LaneGate *g;
        ~GateRelease() { if (g) g->open(2); }
    } gate_release{ctx_gate_out(c)};
    std::vector<EncStream> hs;
    uint32_t ring = ctx_parse_ring(c) ? 1u : 0u;   // the ring / stream encoder's parse (encode/frontend_ring.rs) instead of the slice parse
    // one window of a longer stream (stream.hip): a single stream, ring parse
    EncWindow *win = ctx_window(c);
    if (win && (count != 1 || !ring || src_len[0] < RING_SIZE || src_len[0] > 0x7FFFFFFFull)) return LZFSE_MI_BAD_ARGUMENT;
    if (win && win->beyond) ring |= RING_CONT;
    // one block of a slice that the reference's front end matches in several (encode_slice_blocks, api.hip)
    RepoWindow *repo = ctx_repo(c);
    static_assert(sizeof(RepoEvent) == sizeof(MatchRec), "RepoEvent is MatchRec");
    if (repo && (count != 1 || ring || win || src_len[0] <= VN_CUTOFF || src_len[0] > 0xF0000000ull)) return LZFSE_MI_BAD_ARGUMENT;
    for (uint32_t i = 0; i < count; i++) {
        out_lens[i] = 0;
        statuses[i] = LZFSE_MI_OK;
        uint64_t n = src_len[i];
        if (n <= VN_CUTOFF) { statuses[i] = LZFSE_MI_UNSUPPORTED; continue; }  // host-side size classes (frontend_bytes.rs:63-77)
        // (a slice of up to BLOCK_GUIDE + 3 = 0x8000_0002 bytes is ONE block of the reference's front end, frontend_bytes.rs:169-178;
        // beyond that it repositions, :348-375, which is not built)
        if (n > 0x80000002ull && !repo) { statuses[i] = LZFSE_MI_UNSUPPORTED; continue; }
        EncStream e{};
        e.src_off = src_off[i]; e.dst_off = dst_off[i]; e.dst_cap = dst_cap[i];
        e.n = (uint32_t)n; e.user_index = i; e.ring = ring;
        if (win && win->start) {
            e.start = 1; e.st_index = win->st[0]; e.st_lit = win->st[1]; e.st_pidx = win->st[2]; e.st_pmidx = win->st[3]; e.st_plen = win->st[4]; e.st_skip = win->skip;
            if (e.st_lit > e.st_index || e.st_index >= e.n - 3) return LZFSE_MI_BAD_ARGUMENT;
            e.st_raw = e.st_lit + e.st_skip;
        }
        if (repo) {
            e.rel0 = repo->rel0; e.stop = repo->final ? 0u : repo->stop; e.no_flush = repo->final ? 0u : 1u; e.n_carry = repo->n_carry;
            if (!repo->first) {
                e.start = 1; e.st_index = repo->rel0 + MAX_D_VALUE; e.st_lit = repo->st_lit; e.st_pidx = repo->st_pidx; e.st_pmidx = repo->st_pmidx;
                e.st_plen = repo->st_plen; e.st_skip = repo->st_skip; e.st_raw = repo->st_raw;
                if (e.st_lit > e.st_index || e.st_index >= e.n - 3) return LZFSE_MI_BAD_ARGUMENT;
            }
            if (e.stop && (e.stop > e.n - 3 || e.stop <= e.st_index)) return LZFSE_MI_BAD_ARGUMENT;
        }
        hs.push_back(e);
    }
    const uint32_t ns = (uint32_t)hs.size();
    if (ns == 0) return LZFSE_MI_OK;
    // longest streams first: per-stream serial stages of the longest stream bound the batch
    std::stable_sort(hs.begin(), hs.end(), [](const EncStream &a, const EncStream &b) { return a.n > b.n; });
    std::vector<EncTile> ht;
    uint64_t pos_total = 0, lmd_total = 0, stage_total = 0, match_total = 0;
    uint32_t blk_total = 0, range_total = 0, seg_total = 0;
    // segment length of the speculative parse: one value for the batch (seg_for, enc_common.h)
    uint64_t batch_pos = 0;
    for (uint32_t si = 0; si < ns; si++) batch_pos += hs[si].n;
    const uint32_t seg = seg_for(batch_pos, hs[0].n /* sorted: the longest */, ns), ev_cap = seg_ev_cap(seg);
    for (uint32_t si = 0; si < ns; si++) {
        EncStream &e = hs[si];
        e.pos_base = pos_total;
        e.tile_base = (uint32_t)ht.size();
        const uint32_t n_pos = e.n - 3;
        for (uint32_t p = 0; p < n_pos; p += TILE_POS) ht.push_back({si, e.n, p, e.ring, e.src_off, e.pos_base, 0u, 0u});
        e.blk_base = blk_total;
        e.blk_cap = e.n / 39000 + 2 + ring;
        e.lmd_base = lmd_total;
        e.lmd_cap = e.n / 4 + e.n / 256 + 2 * e.blk_cap + 64;
        e.lmd_cap = (e.lmd_cap + 63) & ~63u;
        e.stage_base = stage_total;
        e.stage_cap = (uint64_t)e.n + e.n / 2 + e.n / 4 + (uint64_t)e.blk_cap * 1024 + 4096;
        e.stage_cap = (e.stage_cap + 255) & ~255ull;
        e.seg_base = seg_total;
        e.n_seg = (n_pos + seg - 1) / seg;
        seg_total += e.n_seg;
        e.range_base = range_total;
        // (ring parse: up to one more event and range per 16 KiB block, the literals a round pushes when they pass the head)
        e.range_cap = 2 * e.n_seg + 4 + (ring ? e.n / RING_BLK + 4 : 0);
        e.match_base = match_total;
        e.match_cap = e.n / 4 + 8 + (ring ? e.n / RING_BLK + 4 : 0) + e.n_carry + 2;
        blk_total += e.blk_cap;
        range_total += e.range_cap;
        pos_total += ((uint64_t)e.n + 255) & ~255ull;
        lmd_total += e.lmd_cap;
        stage_total += e.stage_cap;
        match_total += e.match_cap;
    }
    const uint32_t nt = (uint32_t)ht.size(), nseg = seg_total;
    mark(1);
    // chain tiles: 1, 2 or 4 candidate tiles long (enc_common.h, chain_tile_mult)
    const uint32_t ch_mult = chain_tile_mult([&](uint32_t m) { uint64_t k = 0; for (const EncStream &e : hs) k += ((uint64_t)e.n - 3 + (uint64_t)TILE_POS * m - 1) / ((uint64_t)TILE_POS * m); return k; },
                                             ctx_diag_chain(c) >> 4);
    const uint32_t ch_pos = TILE_POS * ch_mult;
    std::vector<EncTile> hct;
    for (uint32_t si = 0; si < ns; si++)
        for (uint32_t p = 0; p < hs[si].n - 3; p += ch_pos)
            hct.push_back({si, hs[si].n, p, hs[si].ring, hs[si].src_off, hs[si].pos_base, repo ? repo->skip_lo : 0u, repo ? repo->skip_hi : 0u});
    const uint32_t nct = (uint32_t)hct.size();

    if (!eb_ensure(S, EB_STREAMS, ns * sizeof(EncStream)) || !eb_ensure(S, EB_TILES, nt * sizeof(EncTile)) || !eb_ensure(S, EB_CTILES, nct * sizeof(EncTile)) ||
        !eb_ensure(S, EB_PREV, pos_total * 4) || !eb_ensure(S, EB_SUMMARY, (size_t)nct * ((2u << HASH_BITS) + 64 + 2) * 4) ||
        !eb_ensure(S, EB_REC, pos_total * 4) || !eb_ensure(S, EB_LMDS, lmd_total * 8) ||
        !eb_ensure(S, EB_BLOCKS, (size_t)blk_total * sizeof(EncBlock)) || !eb_ensure(S, EB_OUTS, ns * sizeof(EncStreamOut)) ||
        !eb_ensure(S, EB_STAGE, stage_total + 256) || !eb_ensure(S, EB_SLOTS, (size_t)blk_total * 4) ||
        !eb_ensure(S, EB_BITMAP, pos_total / 8 + 64))
        return LZFSE_MI_IO;
    if (!eb_ensure(S, EB_SEGS, (size_t)nseg * sizeof(uint2)) || !eb_ensure(S, EB_LOGS, (size_t)nseg * ev_cap * sizeof(SpecEvent)) ||
        !eb_ensure(S, EB_HDRS, (size_t)nseg * sizeof(SpecHeader)) || !eb_ensure(S, EB_RANGES, (size_t)range_total * sizeof(RangeRec)) ||
        !eb_ensure(S, EB_GAPS, match_total * sizeof(MatchRec)) || !eb_ensure(S, EB_MATCHES, match_total * sizeof(MatchRec)) ||
        !eb_ensure(S, EB_PC, match_total * 4) || !eb_ensure(S, EB_PL, match_total * 4) ||
        !eb_ensure(S, EB_RSLOTS, (size_t)range_total * 4) || !eb_ensure(S, EB_SYNC, (size_t)nseg * sizeof(uint4)) ||
        !eb_ensure(S, EB_RSUM, (size_t)range_total * sizeof(uint2)))
        return LZFSE_MI_IO;
    mark(2);
    EncStream *d_streams = (EncStream *)S.bufs[EB_STREAMS];
    EncTile *d_tiles = (EncTile *)S.bufs[EB_TILES];
    uint32_t *d_prev = (uint32_t *)S.bufs[EB_PREV];
    uint32_t *d_summary = (uint32_t *)S.bufs[EB_SUMMARY];
    uint32_t *d_flist = d_summary + (size_t)nct * (1u << HASH_BITS), *d_fcount = d_flist + (size_t)nct * ((1u << HASH_BITS) + 64), *d_redo = d_fcount + nct;
    EncTile *d_ctiles = (EncTile *)S.bufs[EB_CTILES];
    uint32_t *d_rec = (uint32_t *)S.bufs[EB_REC];
    uint2 *d_lmds = (uint2 *)S.bufs[EB_LMDS];
    EncBlock *d_blocks = (EncBlock *)S.bufs[EB_BLOCKS];
    EncStreamOut *d_outs = (EncStreamOut *)S.bufs[EB_OUTS];
    uint8_t *d_stage = (uint8_t *)S.bufs[EB_STAGE];
    uint32_t *d_slots = (uint32_t *)S.bufs[EB_SLOTS];
    uint64_t *d_bitmap = (uint64_t *)S.bufs[EB_BITMAP];
    // (descriptors travel from pinned arrays of the context, PinVec: the uploads are asynchronous, the candidate stage is queued
    // before the host has made the lists the later stages want)
    CtlArray<EncStream> ps(S.host[EH_STREAMS], ns);
    CtlArray<EncTile> pt(S.host[EH_TILES], nt), pct(S.host[EH_CTILES], nct);
    std::memcpy(ps.data(), hs.data(), ns * sizeof(EncStream));
    std::memcpy(pt.data(), ht.data(), nt * sizeof(EncTile));
    std::memcpy(pct.data(), hct.data(), nct * sizeof(EncTile));
    E_TRY(hipMemcpyAsync(d_streams, ps.data(), ns * sizeof(EncStream), hipMemcpyHostToDevice, stq));
    E_TRY(hipMemcpyAsync(d_tiles, pt.data(), nt * sizeof(EncTile), hipMemcpyHostToDevice, stq));
    E_TRY(hipMemcpyAsync(d_ctiles, pct.data(), nct * sizeof(EncTile), hipMemcpyHostToDevice, stq));
    E_TRY(hipMemsetAsync(d_outs, 0, ns * sizeof(EncStreamOut), stq));
    E_TRY(hipMemsetAsync(d_bitmap, 0, pos_total / 8 + 64, stq));
    mark(3);
    if (LaneGate *gi = ctx_gate_in(c)) {
        // lane of a split call: start when the previous lane has queued its candidate kernel (LaneGate, internal.h)
        if (gi->wait() == 1) (void)hipStreamWaitEvent(stq, gi->ev, 0);
    }
    {
        StageTimer t(c, "enc_chain");
        launch_enc_chain(d_src, d_ctiles, nct, ch_pos, d_prev, d_summary, d_flist, d_fcount, d_redo, (ctx_diag_chain(c) & 1) != 0, stq);
    }
    {
        StageTimer t(c, "enc_link");
        launch_enc_link(d_ctiles, nct, ch_pos, d_prev, d_summary, d_flist, d_fcount, stq);
    }
    if (LaneGate *go = ctx_gate_out(c)) go->open(hipEventRecord(go->ev, stq) == hipSuccess ? 1 : 2);
    {
        StageTimer t(c, "enc_cand");
        launch_enc_cand(d_src, d_streams, d_tiles, nt, d_prev, d_rec, d_bitmap, stq);
    }
    {
        uint2 *d_segs = (uint2 *)S.bufs[EB_SEGS];
        SpecEvent *d_logs = (SpecEvent *)S.bufs[EB_LOGS];
        SpecHeader *d_hdrs = (SpecHeader *)S.bufs[EB_HDRS];
        RangeRec *d_ranges = (RangeRec *)S.bufs[EB_RANGES];
        MatchRec *d_gaps = (MatchRec *)S.bufs[EB_GAPS], *d_matches = (MatchRec *)S.bufs[EB_MATCHES];
        uint32_t *d_pc = (uint32_t *)S.bufs[EB_PC], *d_pl = (uint32_t *)S.bufs[EB_PL], *d_rslots = (uint32_t *)S.bufs[EB_RSLOTS];
        // the lists of segments, block slots and range slots (a million entries for a 750 MB batch: 0.6 ms of this thread) are made
        // while the device is busy with the candidate stage
        CtlArray<uint2> psegs(S.host[EH_SEGS], nseg);
        CtlArray<uint32_t> pslots(S.host[EH_SLOTS], blk_total), prslots(S.host[EH_RSLOTS], range_total);
        for (uint32_t si = 0; si < ns; si++) {
            const EncStream &e = hs[si];
            uint2 *sg = psegs.data() + e.seg_base;
            for (uint32_t k = 0; k < e.n_seg; k++) sg[k] = make_uint2(si, k);
            std::fill_n(pslots.data() + e.blk_base, e.blk_cap, si);
            std::fill_n(prslots.data() + e.range_base, e.range_cap, si);
        }
        E_TRY(hipMemcpyAsync(d_slots, pslots.data(), (size_t)blk_total * 4, hipMemcpyHostToDevice, stq));
        E_TRY(hipMemcpyAsync(d_segs, psegs.data(), (size_t)nseg * sizeof(uint2), hipMemcpyHostToDevice, stq));
        E_TRY(hipMemcpyAsync(d_rslots, prslots.data(), (size_t)range_total * 4, hipMemcpyHostToDevice, stq));
        {
            StageTimer t(c, "enc_spec");
            launch_enc_spec(d_src, d_streams, d_segs, nseg, seg, d_prev, d_rec, d_bitmap, d_logs, d_hdrs, repo && !repo->final, stq);
        }
        uint4 *d_gstate = nullptr;
        if (win && !win->final) {
            if (!eb_ensure(S, EB_GSTATE, match_total * sizeof(uint4)) || !eb_ensure(S, EB_CUT, ns * sizeof(EncCut))) return LZFSE_MI_IO;
            d_gstate = (uint4 *)S.bufs[EB_GSTATE];
        }
        if (repo && repo->n_carry)   // (the events of the bvx2 block the front end had not closed: in front of the stream's gap events)
            E_TRY(hipMemcpyAsync(d_gaps + hs[0].match_base, repo->carry, (size_t)repo->n_carry * sizeof(MatchRec), hipMemcpyHostToDevice, stq));
        {
            StageTimer t(c, "enc_stitch");
            launch_enc_stitch(d_src, d_streams, ns, d_segs, nseg, seg, d_prev, d_rec, d_bitmap, d_logs, d_hdrs, (uint4 *)S.bufs[EB_SYNC], d_ranges, d_gaps,
                              d_gstate, d_outs, stq);
        }
        {
            StageTimer t(c, "enc_compact");
            launch_enc_compact(d_streams, d_rslots, range_total, ns, d_outs, d_ranges, d_logs, d_gaps, d_matches, d_pc, d_pl,
                               (uint2 *)S.bufs[EB_RSUM], stq);
        }
        {
            StageTimer t(c, "enc_segment");
            if (!eb_ensure(S, EB_SEGFLAG, (size_t)ns * 4 + 64)) return LZFSE_MI_IO;
            const bool big_stream = hs[0].n >= (1u << 20);   // (streams are sorted by size: enc_segpar_kernel only takes large ones)
            launch_enc_segment(d_streams, ns, d_slots, blk_total, big_stream, d_matches, d_pc, d_pl, d_lmds, d_blocks, d_outs,
                               (uint32_t *)S.bufs[EB_SEGFLAG], stq);
        }
        {
            StageTimer t(c, "enc_lmd");
            launch_enc_lmd(d_streams, d_slots, blk_total, d_outs, d_blocks, d_matches, d_pc, d_lmds, stq);
        }
    }
    const int diag_stats = ctx_diag_stats(c);
    {
        StageTimer t(c, "enc_block");
        unsigned long long *d_cyc = nullptr;
        if ((diag_stats & 1) && eb_ensure(S, EB_DBG, 64)) {
            d_cyc = (unsigned long long *)S.bufs[EB_DBG];
            E_TRY(hipMemsetAsync(d_cyc, 0, 64, stq));
        }
        hipLaunchKernelGGL(enc_block_kernel, dim3(blk_total), dim3(BLK_THREADS), 0, stq, d_src, d_streams, ns, d_outs, d_lmds,
                           d_blocks, d_slots, d_stage, (uint8_t *)d_rec, d_cyc);
    }
    {
        StageTimer t(c, "enc_pack");
        hipLaunchKernelGGL(enc_pack_kernel, dim3(blk_total), dim3(256), 0, stq, d_streams, d_slots, d_blocks, d_stage, d_dst, d_outs);
    }
    CtlArray<EncStreamOut> ho(S.host[EH_OUTS], ns + 1);   // (+ the window's cut record, behind the streams')
    EncCut hcut{};
    if (win && !win->final) {
        // where this window of a longer stream is cut: the blocks in front of that point are final, the parse goes on from there
        launch_enc_cut(d_streams, ns, d_outs, d_blocks, (const RangeRec *)S.bufs[EB_RANGES], (const SpecEvent *)S.bufs[EB_LOGS],
                       (const MatchRec *)S.bufs[EB_GAPS], (const uint4 *)S.bufs[EB_GSTATE], (EncCut *)S.bufs[EB_CUT], stq);
        E_TRY(hipMemcpyAsync(&hcut, S.bufs[EB_CUT], sizeof hcut, hipMemcpyDeviceToHost, stq));   // (a window of a stream: one small record, pageable)
    }
    mark(4);
    E_TRY(hipMemcpyAsync(ho.data(), d_outs, ns * sizeof(EncStreamOut), hipMemcpyDeviceToHost, stq));
    E_TRY(hipStreamSynchronize(stq));
    mark(5);
#ifdef LZFSE_MI_DIAG
    if (trace_host) {
        auto ms = [&](int a, int b) { return std::chrono::duration<double, std::milli>(tp[b] - tp[a]).count(); };
        fprintf(stderr, "enc_host ctx=%p streams=%u total=%.3f prep=%.3f alloc=%.3f upload=%.3f launches=%.3f wait=%.3f\n", (void *)c, ns, ms(0, 5), ms(0, 1),
                ms(1, 2), ms(2, 3), ms(3, 4), ms(4, 5));
    }
#endif
    if (hipGetLastError() != hipSuccess) return LZFSE_MI_IO;
    for (uint32_t i = 0; i < ns; i++) {
        uint32_t u = hs[i].user_index;
        statuses[u] = ho[i].status;
        out_lens[u] = ho[i].status ? 0 : ho[i].out_len;
    }
    if (repo && !repo->final && !ho[0].status) {
        // a block that is not the slice's last: the bvx2 blocks that are complete are final; the last one (unclosed when the walk
        // ended) is carried into the next call as its events, and the walk's state goes with it
        const EncStreamOut &o = ho[0];
        repo->e_lit = o.e_lit; repo->e_pidx = o.e_pidx; repo->e_pmidx = o.e_pmidx; repo->e_plen = o.e_plen; repo->e_cross = o.e_cross;
        std::vector<EncBlock> hb(o.n_blocks);
        if (o.n_blocks) E_TRY(hipMemcpy(hb.data(), d_blocks + hs[0].blk_base, (size_t)o.n_blocks * sizeof(EncBlock), hipMemcpyDeviceToHost));
        uint64_t bytes = 0;
        for (uint32_t b = 0; b + 1 < o.n_blocks; b++) bytes += (uint64_t)hb[b].hdr_len + hb[b].lit_len + hb[b].lmd_len;
        repo->final_bytes = bytes;
        uint32_t first_ev = 0, skip = hs[0].st_skip;
        if (o.n_blocks >= 2) { first_ev = hb[o.n_blocks - 2].cut_ev; skip = (first_ev == 0 ? hs[0].st_skip : 0u) + hb[o.n_blocks - 2].cut_skip; }
        if (first_ev == NONE || first_ev > o.n_matches) return LZFSE_MI_IO;
        repo->left.resize(o.n_matches - first_ev);
        if (!repo->left.empty())
            E_TRY(hipMemcpy(repo->left.data(), (const MatchRec *)S.bufs[EB_MATCHES] + hs[0].match_base + first_ev, repo->left.size() * sizeof(MatchRec), hipMemcpyDeviceToHost));
        repo->left_skip = repo->left.empty() ? 0u : skip;
        repo->left_raw = o.n_blocks ? hb[o.n_blocks - 1].src_start : o.e_lit;
        out_lens[hs[0].user_index] = bytes;
    }
    if (win && !win->final && !ho[0].status) {
        win->found = hcut.found;
        win->index = hcut.index; win->lit = hcut.lit; win->p_idx = hcut.p_idx; win->p_midx = hcut.p_midx; win->p_len = hcut.p_len; win->skip_out = hcut.skip;
        out_lens[hs[0].user_index] = hcut.found ? hcut.out_len : 0;
    }
    if ((diag_stats & 1) && S.bufs[EB_DBG]) {
        unsigned long long hc[8];
        if (hipMemcpy(hc, S.bufs[EB_DBG], 64, hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "enc_block cycles/block: gather %llu lit_hist %llu normalize %llu tables %llu | per chunk pass: lookup %llu chains %llu pack+store %llu (blocks %u)\n",
                    hc[0] / blk_total, hc[1] / blk_total, hc[2] / blk_total, hc[3] / blk_total, hc[4] / blk_total, hc[5] / blk_total,
                    hc[6] / blk_total, blk_total);
    }
    if (diag_stats & 2) {
        for (uint32_t i = 0; i < ns && i < 16; i++)
            fprintf(stderr, "walk[%u] n=%u segs=%u matches=%u ranges=%u true_iters=%u syncs=%u fallbacks=%u cycles=%llu\n", i, hs[i].n,
                    hs[i].n_seg, ho[i].n_matches, ho[i].n_ranges, ho[i].iters, ho[i].emits, ho[i].capped,
                    (unsigned long long)ho[i].cycles);
    }
    return LZFSE_MI_OK;
}

#ifdef LZFSE_MI_DIAG
// Debug hook of the diagnostic build for stage-level parity tests (tests/test_gpu_encode.py): the chain links and the
// per-position candidate records of ONE stream. Not part of the ABI; the product library does not export it.
extern "C" LZFSE_MI_API int lzfse_mi_debug_candidates(lzfse_mi_ctx *c, const uint8_t *h_src, size_t n, uint32_t *h_prev, uint32_t *h_rec_xy) {
    if (!c || n <= VN_CUTOFF || n > 0x7FFFFFFFull) return LZFSE_MI_BAD_ARGUMENT;
    hipStream_t stq = ctx_stream(c);
    EncScratch &S = ctx_enc(c);
    EncStream e{};
    e.src_off = 0; e.pos_base = 0; e.n = (uint32_t)n;
    std::vector<EncTile> ht, hct;
    for (uint32_t p = 0; p < e.n - 3; p += TILE_POS) ht.push_back({0u, e.n, p, 0u, 0ull, 0ull});
    const uint32_t nt = (uint32_t)ht.size();
    const uint32_t ch_mult = chain_tile_mult([&](uint32_t m) { return ((uint64_t)e.n - 3 + (uint64_t)TILE_POS * m - 1) / ((uint64_t)TILE_POS * m); }, ctx_diag_chain(c) >> 4);
    const uint32_t ch_pos = TILE_POS * ch_mult;
    for (uint32_t p = 0; p < e.n - 3; p += ch_pos) hct.push_back({0u, e.n, p, 0u, 0ull, 0ull});
    const uint32_t nct = (uint32_t)hct.size();
    size_t padn = (n + 255) & ~(size_t)255;
    if (!eb_ensure(S, EB_STREAMS, sizeof(EncStream)) || !eb_ensure(S, EB_TILES, nt * sizeof(EncTile)) || !eb_ensure(S, EB_CTILES, nct * sizeof(EncTile)) ||
        !eb_ensure(S, EB_PREV, padn * 4) || !eb_ensure(S, EB_SUMMARY, (size_t)nct * ((2u << HASH_BITS) + 64 + 2) * 4) ||
        !eb_ensure(S, EB_REC, padn * 4) || !eb_ensure(S, EB_STAGE, padn + 256) || !eb_ensure(S, EB_BITMAP, padn / 8 + 64))
        return LZFSE_MI_IO;
    uint8_t *d_src = (uint8_t *)S.bufs[EB_STAGE];
    E_TRY(hipMemcpyAsync(d_src, h_src, n, hipMemcpyHostToDevice, stq));
    E_TRY(hipMemcpyAsync(S.bufs[EB_STREAMS], &e, sizeof e, hipMemcpyHostToDevice, stq));
    E_TRY(hipMemcpyAsync(S.bufs[EB_TILES], ht.data(), nt * sizeof(EncTile), hipMemcpyHostToDevice, stq));
    E_TRY(hipMemcpyAsync(S.bufs[EB_CTILES], hct.data(), nct * sizeof(EncTile), hipMemcpyHostToDevice, stq));
    uint32_t *sm = (uint32_t *)S.bufs[EB_SUMMARY];
    uint32_t *fl = sm + (size_t)nct * (1u << HASH_BITS), *fc = fl + (size_t)nct * ((1u << HASH_BITS) + 64), *redo = fc + nct;
    launch_enc_chain(d_src, (EncTile *)S.bufs[EB_CTILES], nct, ch_pos, (uint32_t *)S.bufs[EB_PREV], sm, fl, fc, redo, (ctx_diag_chain(c) & 1) != 0, stq);
    launch_enc_link((EncTile *)S.bufs[EB_CTILES], nct, ch_pos, (uint32_t *)S.bufs[EB_PREV], sm, fl, fc, stq);
    launch_enc_cand(d_src, (EncStream *)S.bufs[EB_STREAMS], (EncTile *)S.bufs[EB_TILES], nt, (uint32_t *)S.bufs[EB_PREV], (uint32_t *)S.bufs[EB_REC],
                    (uint64_t *)S.bufs[EB_BITMAP], stq);
    E_TRY(hipMemcpyAsync(h_prev, S.bufs[EB_PREV], (n - 3) * 4, hipMemcpyDeviceToHost, stq));  // link records
    std::vector<uint32_t> packed(n - 3);
    E_TRY(hipMemcpyAsync(packed.data(), S.bufs[EB_REC], (n - 3) * 4, hipMemcpyDeviceToHost, stq));
    E_TRY(hipStreamSynchronize(stq));
    // (the tests read a record as the pair {distance | backward << 18 | capped << 31, forward length})
    for (size_t k = 0; k + 3 < n; k++) {
        const uint32_t r = packed[k], fwd = (r >> 18) & 0x3FFu;
        h_rec_xy[2 * k] = (r & 0x3FFFFu) | ((r >> 28) << 18) | (fwd == FCAP ? 0x80000000u : 0u);
        h_rec_xy[2 * k + 1] = fwd;
    }
    return hipGetLastError() == hipSuccess ? LZFSE_MI_OK : LZFSE_MI_IO;
}
#endif

}  // namespace lzmi
