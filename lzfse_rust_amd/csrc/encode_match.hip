// Match finding of the MI355X LZFSE encoder: hand-written HIP kernels for gfx950 (wave64).
//
// The reference's encoder is a serial greedy/lazy parse over a 4-way, 2^14-bucket history
// table (encode/frontend_bytes.rs:160-268, encode/history.rs:15-118). Its table state before
// position i is a pure function of src[0..i+3] because every position is inserted exactly once,
// in order (frontend_bytes.rs:187,336-344). That makes the expensive part position-parallel:
//
//   enc_chain_kernel   per chain tile (64 Ki .. 256 Ki positions): link record of every position = distance to the previous
//                      position of its bucket (history.rs:221-224 hash, fse/object.rs:38-43) + check bits of that position,
//                      exact, in order, one LDS exchange per position on a 16 384-entry last-seen table
//                      (enc_chain_ballot_kernel: the same without the exchange, as a checked fallback)
//   enc_link_kernel    first occurrences of a tile: link to earlier tiles (window 262 139)
//   enc_cand_kernel    per position: <= 4 chain entries newest->oldest (<= 3 dependent gathers) with the reference's
//                      gates (frontend_bytes.rs:214-231), forward LCP (match_kit/match_fast.rs:
//                      22-49) and un-gated backward LCS (:61-89), both capped
//
// (A replay of the 4-entry table itself -- bucket-partitioned tables in LDS fed by a stable multi-split of the positions,
// writing every position's four candidates -- was built and measured in round 2: it removes the dependent hops from
// enc_cand_kernel but costs more than the chains it replaces)
#include <type_traits>

#include "enc_common.h"

namespace lzmi {

// ------------------------------------------------------------------------------------ chains

// first-occurrence list of a tile: one slot per bucket + 64 slots where lanes that have nothing to list store
constexpr uint32_t FL_STRIDE = (1u << HASH_BITS) + 64;
#ifndef LZMI_CH_WAVES
#define LZMI_CH_WAVES 8
#endif
#ifndef LZMI_CH_STEPS
#define LZMI_CH_STEPS 16
#endif
#ifndef LZMI_CH_OCC
#define LZMI_CH_OCC 4
#endif
constexpr int CH_WAVES = LZMI_CH_WAVES;         // waves of a chain workgroup: they take the tile's batches in turn
constexpr int CH_STEPS = LZMI_CH_STEPS;         // steps of 64 positions per batch (16: the batch's values fit 128 registers, four waves per SIMD)
constexpr uint32_t CH_POS = 64 * CH_STEPS;     // positions per batch

// One workgroup per chain tile (up to 4 x TILE_POS positions: the host picks the length, chain_tile_mult); positions in order,
// 64 per step. The last-seen table of the tile lives in LDS, one 32-bit entry per bucket, and a step is ONE LDS exchange per
// lane: the entry a lane gets back is its predecessor in the bucket -- an earlier step's, or a lower lane's of the same step,
// because gfx950 serialises the lanes of one ds_wrxchg that hit the same address in ascending lane order (measured;
// scripts/micro/xchg_order.hip). That order is not architectural, so every lane checks what it got (a predecessor must lie
// before it) and a tile that ever sees anything else is redone by enc_chain_ballot_kernel, which assumes nothing.
//
// Round 5: EIGHT waves per tile instead of two, and the exchanges are the only thing they do in turn. The LDS unit takes a
// wave's exchange in ~8 cycles when it is kept busy (scripts/micro/xchg_rate.hip, profiles/r05_xchg_rate.txt) -- rounds 2 to 4
// read the kernel as bound by "the rate of LDS exchanges, ~64 cycles per wave instruction", but what they measured was a lone
// wave per SIMD issuing one instruction every ~5 cycles: hashes, address clamps, ballots, 40-odd instructions per step on either
// side of the exchange. A batch of 32 steps belongs to ONE wave from its loads to its link records; only between "turn == b"
// and "turn = b + 1" (an LDS word) does the order of the batches matter, and that section is the 32 exchanges and their return.
// Everything else -- the loads of the batch after next, the hashes, the records, the first-occurrence list (its slots are
// reserved per batch with one LDS add) -- runs in eight waves side by side.
__global__ __launch_bounds__(64 * CH_WAVES) __attribute__((amdgpu_waves_per_eu(LZMI_CH_OCC, LZMI_CH_OCC))) void enc_chain_kernel(const uint8_t *__restrict__ src, const EncTile *__restrict__ tiles, uint32_t n_tiles,
                                                                  uint32_t tile_pos, uint32_t *__restrict__ prev, uint32_t *__restrict__ summary,
                                                                  uint32_t *__restrict__ flist, uint32_t *__restrict__ fcount,
                                                                  uint32_t *__restrict__ redo, uint32_t force_redo) {
    __shared__ uint32_t last[1u << HASH_BITS];  // seen_make(): (offset in tile + 1) | check bits << 18, 0 = none
    __shared__ uint32_t sh_turn, sh_nfirst, sh_wrong;
    const uint32_t t = blockIdx.x;
    if (t >= n_tiles) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (force_redo) {   // (diagnostic build: every tile goes through the ballot kernel)
        if (tid == 0) redo[t] = 1;
        return;
    }
    const EncTile tl = tiles[t];
    for (uint32_t k = tid; k < (1u << HASH_BITS); k += 64 * CH_WAVES) last[k] = 0;
    if (tid == 0) { sh_turn = 0; sh_nfirst = 0; sh_wrong = 0; }
    const uint8_t *s = src + tl.src_off;
    uint32_t *pv = prev + tl.pos_base;  // link records (enc_common.h)
    const uint32_t n_pos = tl.n - 3;  // positions 0 .. n-4 are hashed (frontend_bytes.rs:166-170)
    const uint32_t t_end = tl.start + tile_pos < n_pos ? tl.start + tile_pos : n_pos;
    uint32_t *fl = flist + (uint64_t)t * FL_STRIDE;
    const uint32_t n_batches = (t_end - tl.start + CH_POS - 1) / CH_POS;
    const uint32_t q_last = t_end - 1;
    // All loads are unconditional (addresses clamped to the tile's last position) and the stores of a full batch too:
    // straight-line code, counted waits. The source values of a wave's next batch are loaded while it works on this one.
    uint32_t nx[CH_STEPS];
#pragma unroll
    for (int j = 0; j < CH_STEPS; j++) {
        const uint32_t q = tl.start + (uint32_t)w * CH_POS + 64 * j + lane;
        nx[j] = ld_u32(s + (q < q_last ? q : q_last));
    }
    uint32_t *pvt = pv + tl.start;   // records of the tile: offsets below 2^18
    bool wrong = false;
    const bool has_skip = tl.skip_hi > tl.skip_lo;
    auto batch = [&](uint32_t b, auto full_tag, auto later_tag) {
        constexpr bool FULL = decltype(full_tag)::value, LATER = decltype(later_tag)::value;
        const uint32_t pb = tl.start + b * CH_POS;
        const uint32_t off0 = pb - tl.start + (uint32_t)lane;   // offset in tile of this lane's position in step 0
        uint32_t key[CH_STEPS], old[CH_STEPS];
        {
            uint32_t ent[CH_STEPS];
#pragma unroll
            for (int j = 0; j < CH_STEPS; j++) {
                key[j] = bucket_of(nx[j]);
                ent[j] = seen_make(off0 + 64 * j + 1, nx[j]);
            }
#pragma unroll
            for (int j = 0; j < CH_STEPS; j++) {   // this wave's next batch
                const uint32_t q = pb + CH_WAVES * CH_POS + 64 * j + lane;
                nx[j] = ld_u32(s + (q < q_last ? q : q_last));
            }
            // ---- this batch's turn at the table: every earlier batch's exchanges have returned to their waves ----
            while (__hip_atomic_load(&sh_turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != b) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            if (has_skip && pb < tl.skip_hi && pb + CH_POS > tl.skip_lo) {
                // (a batch that holds positions the reference never pushed: those make no exchange, and their link records are nobody's business)
#pragma unroll
                for (int j = 0; j < CH_STEPS; j++) {
                    old[j] = 0;
                    const uint32_t q = pb + 64 * j + lane;
                    if ((FULL || q < t_end) && !(q >= tl.skip_lo && q < tl.skip_hi))
                        old[j] = __hip_atomic_exchange(&last[key[j]], ent[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else {
#pragma unroll
            for (int j = 0; j < CH_STEPS; j++) {
                old[j] = 0;
                if (FULL || pb + 64 * j + lane < t_end)
                    old[j] = __hip_atomic_exchange(&last[key[j]], ent[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            }
            // (the LDS alone is waited for: the loads of the next batch and the stores of the last one stay in flight)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __hip_atomic_store(&sh_turn, b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
        }
        // ---- links and first-occurrence list of the batch, from what the exchanges returned ----
        uint32_t n_first = 0;
        if (LATER) {
            // first occurrence of its bucket in the tile: the link into earlier tiles is made by enc_link_kernel from this list
            // (offset | bucket << 18; in any order); the first tile of a stream has nothing before it
            uint32_t cnt = 0;
#pragma unroll
            for (int j = 0; j < CH_STEPS; j++) {
                const uint32_t q = pb + 64 * j + lane;
                // (a position the reference never pushed made no exchange: it is not the first of anything)
                const bool valid = (FULL || q < t_end) && !(has_skip && q >= tl.skip_lo && q < tl.skip_hi);
                cnt += (uint32_t)__popcll(__ballot(valid && (old[j] & 0x3FFFFu) == 0));
            }
            uint32_t base = 0;
            if (lane == 0 && cnt) base = atomicAdd(&sh_nfirst, cnt);
            n_first = e_readlane(base, 0);
        }
#pragma unroll
        for (int j = 0; j < CH_STEPS; j++) {
            if (!FULL && pb + 64 * j >= t_end) continue;  // (not break: the loop must stay fully unrolled, the arrays live in registers)
            const uint32_t off = off0 + 64 * j;
            __builtin_assume(off < (1u << 18));
            const uint32_t qq = pb + 64 * j + lane;
            const bool in_tile = FULL || qq < t_end;
            const bool valid = in_tile && !(has_skip && qq >= tl.skip_lo && qq < tl.skip_hi);
            const uint32_t o = old[j] & 0x3FFFFu;
            wrong = wrong || (valid && o > off);   // a predecessor lies before its position (o is offset + 1)
            if (LATER) {
                const bool first = valid && o == 0;
                const uint64_t fm = __ballot(first);
                const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                fl[first ? n_first + below : (1u << HASH_BITS) + (uint32_t)lane] = off | (key[j] << 18);
                n_first += (uint32_t)__popcll(fm);
            }
            const uint32_t rec = o ? link_make(off + 1 - o, old[j] >> 18) : 0u;  // (inside a tile: < 2^18 and within the match window)
            if (FULL) pvt[off] = rec;
            else if (in_tile) pvt[off] = rec;
        }
    };
    __syncthreads();
    const bool later = tl.start != 0;
    for (uint32_t b = (uint32_t)w; b < n_batches; b += CH_WAVES) {
        const bool full = tl.start + (b + 1) * CH_POS <= t_end;
        if (later) { if (full) batch(b, std::true_type{}, std::true_type{}); else batch(b, std::false_type{}, std::true_type{}); }
        else { if (full) batch(b, std::true_type{}, std::false_type{}); else batch(b, std::false_type{}, std::false_type{}); }
    }
    if (__any(wrong) && lane == 0) sh_wrong = 1;
    __syncthreads();
    if (tid == 0) { fcount[t] = sh_nfirst; redo[t] = sh_wrong; }
    if (sh_wrong) return;
    if (t_end >= n_pos) return;   // the stream's last tile: no later tile looks into its summary (64 KB not written; 3 072 of them in the default batch)
    uint32_t *sm = summary + (uint64_t)t * (1u << HASH_BITS);
    for (uint32_t k = tid; k < (1u << HASH_BITS); k += 64 * CH_WAVES) sm[k] = last[k];
}

// The same links without any assumption about the LDS: the nearest previous position with the same bucket is either a
// lower lane of the same step (found with 14 ballots) or the last-seen entry written by earlier steps. Runs only for
// tiles enc_chain_kernel flagged (never, on the hardware measured); the diagnostic build can send every tile here.
__global__ __launch_bounds__(64) void enc_chain_ballot_kernel(const uint8_t *__restrict__ src, const EncTile *__restrict__ tiles, uint32_t n_tiles,
                                                              uint32_t tile_pos, uint32_t *__restrict__ prev, uint32_t *__restrict__ summary,
                                                              uint32_t *__restrict__ flist, uint32_t *__restrict__ fcount,
                                                              const uint32_t *__restrict__ redo) {
    __shared__ uint32_t last[1u << HASH_BITS];  // offset in tile + 1, 0 = none
    // (a few workgroups look through all the tiles' flags: the kernel is queued behind every enc_chain_kernel and finds nothing to
    // do -- one workgroup per tile made that 55 us of an 11 500-tile launch, profiles/r04_bench_kernel_stats.csv)
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    if (!redo[t]) continue;
    __syncthreads();
    const EncTile tl = tiles[t];
    const int lane = e_lane();
    for (uint32_t k = lane; k < (1u << HASH_BITS); k += 64) last[k] = 0;
    const uint8_t *s = src + tl.src_off;
    uint32_t *pv = prev + tl.pos_base;
    const uint32_t n_pos = tl.n - 3;
    const uint32_t t_end = tl.start + tile_pos < n_pos ? tl.start + tile_pos : n_pos;
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    uint32_t *fl = flist + (uint64_t)t * FL_STRIDE;
    uint32_t n_first = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t p0 = tl.start; p0 < t_end; p0 += 64) {
        const uint32_t p = p0 + lane;
        const bool valid = p < t_end && !(p >= tl.skip_lo && p < tl.skip_hi);   // (positions the reference never pushed: EncTile::skip_lo)
        const uint32_t v = valid ? ld_u32(s + p) : 0u;
        const uint32_t key = bucket_of(v);
        const uint32_t old = last[key];
        const uint32_t mine = p - tl.start + 1;
        uint32_t pr = old ? (tl.start + old - 1) : NONE_TILE;
        uint64_t same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < (int)HASH_BITS; b++) {
            uint64_t bb = __ballot((key >> b) & 1);
            same &= ((key >> b) & 1) ? bb : ~bb;
        }
        const uint64_t lower = same & lt_mask;
        if (lower) pr = p0 + (63 - __builtin_clzll(lower));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (valid && (same >> lane) >> 1 == 0) last[key] = mine;  // newest of its bucket wins
        const bool first = valid && pr == NONE_TILE;
        if (tl.start != 0) {
            const uint64_t fm = __ballot(first);
            if (first) fl[n_first + (uint32_t)__popcll(fm & lt_mask)] = (p - tl.start) | (key << 18);
            n_first += (uint32_t)__popcll(fm);
        }
        if (p < t_end) pv[p] = (!valid || pr == NONE_TILE) ? 0u : link_make(p - pr, chk_of(ld_u32(s + pr)));   // (a position the reference never pushed: no link)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) fcount[t] = n_first;
    uint32_t *sm = summary + (uint64_t)t * (1u << HASH_BITS);
    for (uint32_t k = lane; k < (1u << HASH_BITS); k += 64) {
        const uint32_t o = last[k];
        sm[k] = o ? seen_make(o, ld_u32(s + tl.start + o - 1)) : 0u;
    }
    }
}

// Cross-tile links: a bucket's first occurrence in a tile (listed by the chain kernel) points at the newest
// occurrence in an earlier tile of the same stream, whose summary entry also holds its check bits. A tile further back than
// (MAX_D_VALUE - 1) / tile_pos + 1 lies outside the 262 139-byte window (fse/constants.rs:42) and would end the candidate scan anyway.
__global__ void enc_link_kernel(const EncTile *__restrict__ tiles, uint32_t n_tiles, uint32_t tile_pos,
                                uint32_t *__restrict__ prev, const uint32_t *__restrict__ summary,
                                const uint32_t *__restrict__ flist, const uint32_t *__restrict__ fcount) {
    constexpr uint32_t BPT = (1u << HASH_BITS) / 256;  // workgroups per tile
    // (workgroups -> XCDs round-robin; as in enc_cand_kernel, an XCD takes whole tiles, 8 consecutive ones of every 64:
    // the summaries a tile looks into are those of the tiles before it)
    const uint32_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, g = slot / BPT;
    const uint32_t t = (g >> 3) * 64 + xcd * 8 + (g & 7);
    const uint32_t e = (slot % BPT) * blockDim.x + threadIdx.x;
    if (t >= n_tiles || e >= fcount[t]) return;
    const EncTile tl = tiles[t];
    const uint32_t ent = flist[(uint64_t)t * FL_STRIDE + e];
    const uint32_t p = tl.start + (ent & 0x3FFFFu), key = ent >> 18;
    const uint32_t t_idx = tl.start / tile_pos;  // tile index inside the stream (its tiles are consecutive)
    const uint32_t max_back = (MAX_D_VALUE - 1) / tile_pos + 1;
    for (uint32_t back = 1; back <= max_back && back <= t_idx; back++) {
        const uint32_t sv = summary[(uint64_t)(t - back) * (1u << HASH_BITS) + key];
        if (sv != 0) {
            const uint32_t r = tl.start - back * tile_pos + (sv & 0x3FFFFu) - 1;
            if (p - r <= MAX_D_VALUE) prev[tl.pos_base + p] = link_make(p - r, sv >> 18);   // (the chain kernel left 0 here)
            break;
        }
    }
}

// ------------------------------------------------------------------------------------ candidates

__device__ __forceinline__ uint4 ld_u128(const uint8_t *p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

// forward common length of src[a..] and src[b..] (b < a), starting at `len`, bounded by max
__device__ __forceinline__ uint32_t lcp_fwd(const uint8_t *s, uint32_t a, uint32_t b, uint32_t len, uint32_t max) {
    while (len + 8 <= max) {
        uint64_t x = ld_u64(s + a + len) ^ ld_u64(s + b + len);
        if (x) return len + (uint32_t)(__builtin_ctzll(x) >> 3);
        len += 8;
    }
    while (len < max && s[a + len] == s[b + len]) len++;
    return len;
}

__device__ __forceinline__ uint32_t ffbl_m1(uint32_t x) {   // v_ffbl_b32: index of the lowest set bit, 0xFFFFFFFF for 0
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ uint32_t ffbh_m1(uint32_t x) {   // v_ffbh_u32: number of leading zero bits, 0xFFFFFFFF for 0
    uint32_t r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ uint32_t add_sat(uint32_t a, uint32_t b) {   // unsigned add that stays at 0xFFFFFFFF
    uint32_t r;
    asm("v_add_u32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// index of the first non-zero byte of a 16-byte value (the difference of two: 0x1FFFFFFF if none). v_ffbl_b32 gives -1 for a zero dword, and
// the saturating adds keep it there, so the minimum over the four dwords is the first set bit of the 128-bit difference.
__device__ __forceinline__ uint32_t first_set_byte(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
    const uint32_t b0 = ffbl_m1(x0), b1 = add_sat(ffbl_m1(x1), 32u), b2 = add_sat(ffbl_m1(x2), 64u), b3 = add_sat(ffbl_m1(x3), 96u);
    const uint32_t bit = min(min(b0, b1), min(b2, b3));
    return bit >> 3;
}

// minimum over the aligned group of 8 lanes a lane belongs to, in all of them (three DPP steps: the pairs of a quad, the
// halves of a quad, the two quads of a half row)
__device__ __forceinline__ uint32_t group8_min(uint32_t v) {
    // (one v_min_u32_dpp per step; a DPP source written by the instruction before needs two wait states, which the compiler
    // cannot see inside an asm: the s_nop)
    asm("s_nop 1\n"
        "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf"
        : "+v"(v));
    return v;
}

__device__ __forceinline__ uint32_t group4_min(uint32_t v) {   // the same over the aligned group of 4 lanes
    asm("s_nop 1\n"
        "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "+v"(v));
    return v;
}

// (after group4_min: the quads of a half row, then the half rows of a row, exchange their minima)
__device__ __forceinline__ uint32_t half_row_min(uint32_t v) {
    asm("s_nop 1\n"
        "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ uint32_t row_min(uint32_t v) {
    asm("s_nop 1\n"
        "v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
        : "+v"(v));
    return v;
}

// rec[i] = rec_make(dist, fwd_len, bwd) (enc_common.h); 0: no match at i
//
// One lane per position, hops in lock-step. A lane compares at most CAND_C1 bytes on its own. Lanes
// that are still equal there are grouped into runs of consecutive positions with the same distance
// (the inside of one long match): only the head of a run is extended, by the whole wave, and the
// followers derive LCP(i + t, c + t) = LCP(i, c) - t. Repetitive data costs O(1) per position.
constexpr uint32_t CAND_C1 = 12;   // a lane compares 12 bytes on its own: what one dword-aligned 16-byte load holds of a candidate
constexpr int CAND_GL = 4;  // lanes per group of the long-match work list (each takes 32 bytes of a step: 16 heads at a time, 128 bytes per head)
constexpr uint32_t CAND_BPT = (TILE_POS + 255) / 256;  // workgroups per tile
constexpr uint32_t CAND_WIN_DW = (32 + 256 + FCAP + 64 + 36 + 3) / 4 + 1;   // dwords of the workgroup's source window
// (scripts/cand_phases.sh: instruction counts per phase -- the kernel is cut short after phase CAND_ABL with what it has computed kept
// alive by an empty asm, and records "no match" everywhere: a valid all-literal parse; never defined in a library build)
#ifdef CAND_STATS
// (scripts/cand_phases.sh stats: what phase 3 meets -- printed per launch on stderr; never defined in a library build)
__device__ unsigned long long g_cand_stats[32];
#define CAND_STAT(k, v) do { if (lane == 0) atomicAdd(&g_cand_stats[k], (unsigned long long)(v)); } while (0)
#else
#define CAND_STAT(k, v)
#endif
#ifdef CAND_ABL
#define CAND_ABL_EXIT(n, expr)                                                                      \
    if (CAND_ABL == (n)) {                                                                          \
        asm volatile("" ::"v"(expr));                                                               \
        if (valid) rec[tl.pos_base + i] = 0;                                                        \
        if (lane == 0 && i < tl.start + TILE_POS && i < n_pos) bitmap[(tl.pos_base + i) >> 6] = 0;  \
        return;                                                                                     \
    }
#else
#define CAND_ABL_EXIT(n, expr)
#endif

__global__ __launch_bounds__(256) void enc_cand_kernel(const uint8_t *__restrict__ src, const EncStream *__restrict__ streams,
                                                       const EncTile *__restrict__ tiles, uint32_t n_tiles, const uint32_t *__restrict__ prev,
                                                       uint32_t *__restrict__ rec, uint64_t *__restrict__ bitmap) {
    // Workgroups are handed to the 8 XCDs round-robin. All workgroups of one tile go to the same XCD, so the
    // link records and source bytes a tile gathers from (its own 0.5 MB + the 2 MB window before it) stay in
    // that XCD's 4 MB L2 instead of being spread over all eight.
    // (LDS is kept small on purpose: this kernel runs next to another lane's chain kernel, whose tiles need 64 KiB each)
    __shared__ uint16_t q_id[4][256], q_res[4][256];   // per-wave work list of phase 3: slot * 64 + lane of a head; its length
    __shared__ uint32_t q_dist[4][256], q_tab[4][64];  // distance of a head (by slot * 64 + lane); first head per distance hash
    __shared__ uint32_t s_win[CAND_WIN_DW];            // source bytes [i_wg - 32, i_wg + 256 + FCAP + 64 + 36 ..) around the workgroup's 256 positions
    const uint32_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    // g: the g-th tile this XCD works on. An XCD takes 8 CONSECUTIVE tiles of every 64: a tile gathers from the up to
    // four tiles before it (the 262 139-byte window), and those are then the tiles the same L2 has just worked on.
    const uint32_t g = slot / CAND_BPT, bx = slot % CAND_BPT;
    const uint32_t t = (g >> 3) * 64 + xcd * 8 + (g & 7);
    if (t >= n_tiles) return;
    const EncTile tl = tiles[t];
    const uint32_t i = tl.start + bx * blockDim.x + threadIdx.x;
    const uint32_t n = tl.n, n_pos = n - 3;
    if (tl.start + bx * blockDim.x >= n_pos) return;  // block-uniform
    const bool valid = i < n_pos && i < tl.start + TILE_POS;
    const uint8_t *s = src + tl.src_off;
    const uint32_t *pv = prev + tl.pos_base;
    const int lane = e_lane();
    const uint32_t self = valid ? pv[i] : 0u;
    const uint32_t max_total = valid ? n - i : 0;
    const uint32_t cap_total = max_total < FCAP ? max_total : FCAP;
    const uint32_t c1 = cap_total < CAND_C1 ? cap_total : CAND_C1;
    // ---- source window of the workgroup: bytes [i_wg - 32, i_wg + 256 + FCAP + 64 + 36) of the stream (i_wg = position of
    // thread 0; zeros before the stream's first and behind its last byte) go to LDS with coalesced loads. Every lane's OWN
    // side of all byte compares is read from there -- forward up to FCAP + 64 (phase 3), backward up to 32 -- so that the
    // texture addresser, which is what bounds this kernel (TA_BUSY 77 %, profiles/r04_cand_phases_tcp.txt), only sees the
    // candidates' side ----
    {
        const int64_t w0 = (int64_t)(tl.start + bx * blockDim.x) - 32;
        for (uint32_t k = threadIdx.x; k < CAND_WIN_DW; k += 256) {
            const int64_t pos = w0 + 4 * (int64_t)k;
            uint32_t wv4 = 0;
            if (pos >= 0 && pos + 4 <= (int64_t)n) wv4 = ld_u32(s + pos);
            else
                for (int t = 0; t < 4; t++)
                    if (pos + t >= 0 && pos + t < (int64_t)n) wv4 |= (uint32_t)s[pos + t] << (8 * t);
            s_win[k] = wv4;
        }
        __syncthreads();
    }
    const uint32_t *win = s_win + 16 * (threadIdx.x >> 6);   // the wave's view: byte 32 + lane is its lane's position
    // ---- phase 1: follow the chain (<= 3 dependent 4-byte gathers: a link record holds the distance to the next entry and
    // 14 check bits of THAT entry's 4 bytes). Equal check bits = candidate; the byte compare starts at byte 0, so the
    // rare entry whose check bits agree by chance (2^-14) is dropped there: history.rs Item.val == val, exactly. ----
    uint32_t cc[4] = {NONE, NONE, NONE, NONE};
    uint32_t ln[4] = {0, 0, 0, 0};
    {
        const uint32_t wo = 32u + (uint32_t)lane, wq = wo >> 2;
        const uint32_t my_chk = chk_of(__builtin_amdgcn_alignbit(win[wq + 1], win[wq], (wo & 3) * 8));
        bool alive = valid;
        uint32_t c = i, rc = self;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (alive) {
                const uint32_t d = link_dist(rc);
                if (d == 0 || i - (c - d) > MAX_D_VALUE) alive = false;  // frontend_bytes.rs:222-224: stop, not skip
                else {
                    c -= d;
                    if (link_chk(rc) == my_chk) { cc[q] = c; ln[q] = 4; }   // (4: provisional until phase 2 has looked)
                    if (q < 3) rc = pv[c];
                }
            }
        }
    }
    CAND_ABL_EXIT(1, cc[0] ^ cc[1] ^ cc[2] ^ cc[3] ^ ln[0] ^ (ln[1] << 4) ^ (ln[2] << 8) ^ (ln[3] << 12))
    // ---- runs: consecutive positions inside one match see the same distance in the same chain slot, and
    // LCP(i + t, c + t) = LCP(i, c) - t. Only the first lane of such a run (its head) compares bytes; the
    // followers derive their length from the head's. The kernel is bound by the number of cache lines its
    // divergent loads touch, and on compressible data most equal candidates are followers. ----
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    bool fol[4], prov[4];   // prov: candidate by its check bits (before the bytes were looked at)
#pragma unroll
    for (int k = 0; k < 4; k++) prov[k] = ln[k] != 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t dk = ln[k] ? i - cc[k] : NONE;
        const uint32_t dlo = dpp_take<0x138, 0xF>(NONE, dk);   // wave_shr:1 -- one vector instruction; __shfl_up is a trip through the LDS crossbar
        fol[k] = ln[k] != 0 && lane > 0 && dlo == dk;
    }
    CAND_ABL_EXIT(2, cc[0] ^ cc[1] ^ cc[2] ^ cc[3] ^ ln[0] ^ (ln[1] << 4) ^ (ln[2] << 8) ^ (ln[3] << 12) ^ ((uint32_t)fol[0] << 20) ^ ((uint32_t)fol[1] << 21) ^ ((uint32_t)fol[2] << 22) ^ ((uint32_t)fol[3] << 23))
    // ---- phase 2: the first CAND_C1 bytes of all heads together. A candidate's bytes come from ONE dword-aligned 16-byte
    // load (the 16 bytes from c & ~3 hold s[c .. c + 12] at any byte offset) shifted into place: a byte-misaligned vector
    // load costs the texture addresser about four times the cycles of an aligned one, and the addresser is what bounds
    // this kernel (TA_BUSY 77 %). The loads of the four slots are in flight at the same time. ----
    {
        bool tail[4], go[4];
        const bool room = 16 <= max_total;
#pragma unroll
        for (int k = 0; k < 4; k++) { const bool act = ln[k] != 0 && !fol[k]; go[k] = act && room; tail[k] = act && !room; }
        const uint4 zero4 = make_uint4(0, 0, 0, 0);
        uint4 bq[4] = {zero4, zero4, zero4, zero4};
        uint32_t a0 = 0, a1 = 0, a2 = 0;
        if (go[0] || go[1] || go[2] || go[3]) {
            const uint32_t wo = 32u + (uint32_t)lane, q = wo >> 2, sh = (wo & 3) * 8;
            const uint32_t d0 = win[q], d1 = win[q + 1], d2 = win[q + 2], d3 = win[q + 3];
            a0 = __builtin_amdgcn_alignbit(d1, d0, sh); a1 = __builtin_amdgcn_alignbit(d2, d1, sh); a2 = __builtin_amdgcn_alignbit(d3, d2, sh);
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (go[k]) bq[k] = *reinterpret_cast<const uint4 *>((uintptr_t)(s + cc[k]) & ~(uintptr_t)3);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t sh = ((uint32_t)(uintptr_t)(s + cc[k]) & 3u) * 8;
            const uint32_t x0 = a0 ^ __builtin_amdgcn_alignbit(bq[k].y, bq[k].x, sh);
            const uint32_t x1 = a1 ^ __builtin_amdgcn_alignbit(bq[k].z, bq[k].y, sh);
            const uint32_t x2 = a2 ^ __builtin_amdgcn_alignbit(bq[k].w, bq[k].z, sh);
            const uint32_t m2 = x2 ? 8u + ((uint32_t)__builtin_ctz(x2) >> 3) : 12u;
            const uint32_t m1 = x1 ? 4u + ((uint32_t)__builtin_ctz(x1) >> 3) : m2;
            const uint32_t m = x0 ? 0u : m1;   // the entry's 4 bytes differ: its check bits agreed by chance
            ln[k] = go[k] ? (m < c1 ? m : c1) : ln[k];
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (tail[k]) { const uint32_t m = lcp_fwd(s, i, cc[k], 0, c1); ln[k] = m >= 4 ? m : 0u; }  // within 16 bytes of the stream's end
    }
    CAND_ABL_EXIT(3, cc[0] ^ cc[1] ^ cc[2] ^ cc[3] ^ ln[0] ^ (ln[1] << 4) ^ (ln[2] << 8) ^ (ln[3] << 12) ^ ((uint32_t)fol[0] << 20) ^ ((uint32_t)fol[1] << 21) ^ ((uint32_t)fol[2] << 22) ^ ((uint32_t)fol[3] << 23))
    // ---- phase 3: heads still equal after CAND_C1 bytes go to a per-wave work list and are extended by groups of
    // CAND_GL lanes, 32 bytes per lane and step, 16 heads at a time (up to FCAP + 64, so that 63 followers stay exact up to
    // FCAP); a group that has finished its head takes the next one from the list while the others go on with theirs (on the
    // Snappy files a wave has 13 such heads on average, 15 % of them longer than the 128 bytes of a step, 9 % longer than the
    // cap: profiles/r04_cand_stats.txt); then the followers take head - t ----
    {
        const int wv = threadIdx.x >> 6;
        bool more[4], dep[4];
        uint32_t lead[4];
        uint64_t any_more = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
#ifdef CAND_LOWCAP
            // (experiment, experiments/README.md: no phase 3 at all -- a candidate that is still equal after CAND_C1 bytes, and every
            // follower of one, is recorded as "at least": the segment walkers measure it exactly if and when they stand there)
            more[k] = false;
#else
            more[k] = ln[k] == CAND_C1 && !fol[k] && CAND_C1 < cap_total;
#endif
            dep[k] = false; lead[k] = 0;
            any_more |= __ballot(more[k]);
        }
        uint32_t total = 0;
        CAND_STAT(0, 1);
        CAND_STAT(1, any_more != 0);
        if (any_more) {
#ifndef CAND_NO_DEDUP
            // Heads of the wave with the same distance that are < CAND_C1 positions apart lie inside one match (each is
            // >= CAND_C1 long): LCP(i2, i2 - d) = LCP(i0, i0 - d) + i0 - i2. One of them is measured.
            q_tab[wv][lane] = 0xFFFFFFFFu;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (more[k]) {
                    const uint32_t dk = i - cc[k];
                    q_dist[wv][k * 64 + lane] = dk;
                    atomicMin(&q_tab[wv][(dk * 0x9E3779B1u) >> 26], (uint32_t)(k * 64 + lane));
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (more[k]) {
                    const uint32_t dk = i - cc[k];
                    lead[k] = q_tab[wv][(dk * 0x9E3779B1u) >> 26];
                    // (the two heads must lie within the CAND_C1 bytes one of them has verified to be inside one match)
                    const int gap = (int)(lead[k] & 63) - lane;
                    dep[k] = lead[k] != (uint32_t)(k * 64 + lane) && q_dist[wv][lead[k]] == dk &&
                             (uint32_t)(gap < 0 ? -gap : gap) < CAND_C1;
                }
#else
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (more[k]) q_dist[wv][k * 64 + lane] = i - cc[k];
#endif
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool ind = more[k] && !dep[k];
                const uint64_t mk = __ballot(ind);
                if (ind) q_id[wv][total + (uint32_t)__popcll(mk & lt_mask)] = (uint16_t)(k * 64 + lane);
                total += (uint32_t)__popcll(mk);
            }
        }
#ifdef CAND_STATS
        {
            uint32_t nm = 0, nd = 0;
            for (int k = 0; k < 4; k++) { nm += (uint32_t)__popcll(__ballot(more[k])); nd += (uint32_t)__popcll(__ballot(dep[k])); }
            CAND_STAT(2, nm); CAND_STAT(3, nd); CAND_STAT(4, total);
            CAND_STAT(8 + (total == 0 ? 0 : total == 1 ? 1 : total == 2 ? 2 : total <= 4 ? 3 : total <= 8 ? 4 : total <= 16 ? 5 : total <= 32 ? 6 : 7), 1);
        }
        uint32_t n_iter = 0;
#endif
        if (total) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // A step of a group: 128 bytes of its head. The CANDIDATE's bytes come as 16-byte ALIGNED chunks, two per lane
            // (lanes on 128 contiguous bytes cost the texture addresser 0.58 clk per lane and chunk when they are aligned, 1.06
            // when they are not: profiles/r04_gather_bench.txt; round 3 loaded both sides byte-misaligned, 43 % of the kernel's
            // L1 accesses); the position's own bytes are read from the workgroup's window in LDS at whatever byte offset that
            // makes. The first chunk may begin up to 3 bytes before the head's position (those bytes are masked) or inside
            // the CAND_C1 bytes phase 2 has compared (equal again); behind the end of the stream the window holds zeros and
            // whatever differs there lies beyond the limit.
            // A wave with few heads gives each of them more lanes (4 lanes = 128 bytes a step, up to 16 lanes = 512): inside a long run
            // every wave has just its lane 0 as a head, in up to four slots, each of them 1 087 bytes long -- 9 steps with 4 lanes,
            // 3 with 16 (zeros, long periods: profiles/r04_repetitive.txt)
            const uint32_t gl = total <= 4 ? 16u : total <= 8 ? 8u : (uint32_t)CAND_GL, gl_sh = total <= 4 ? 4u : total <= 8 ? 3u : 2u;
            const uint32_t i0 = i - (uint32_t)lane, sub = (uint32_t)lane & (gl - 1);
            uint32_t next = 0, it_lim = 0, it_id = 0, it_w = 0;
            int32_t it_t = 0;           // offset (from the head's position) of the byte the group's NEXT chunk 0 begins with
            const uint4 *it_p = nullptr;   // that chunk
            bool busy = false;
            for (;;) {
                const uint64_t idle = __ballot(!busy);   // (a group is idle or busy as a whole)
                if (next < total && idle) {
                    // the idle groups take the next heads of the list, in group order: a lane counts the idle LANES below it
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                    const uint32_t q = next + (below >> gl_sh);
                    if (!busy && q < total) {
                        busy = true;
                        it_id = q_id[wv][q];
                        const uint32_t hl = it_id & 63, hi = i0 + hl;
                        const uint8_t *cp = s + (hi - q_dist[wv][it_id]);
                        const uintptr_t pa = ((uintptr_t)cp + CAND_C1) & ~(uintptr_t)15;
                        it_p = reinterpret_cast<const uint4 *>(pa);
                        it_t = (int32_t)(uint32_t)(pa - (uintptr_t)cp);   // -3 .. CAND_C1
                        it_w = 32u + 64u * (uint32_t)wv + hl;
                        const uint32_t room = n - hi;
                        it_lim = room < FCAP + 64 ? room : FCAP + 64;
                    }
                    next += (uint32_t)__popcll(idle) >> gl_sh;
                }
                if (!__any(busy)) break;
#ifdef CAND_STATS
                n_iter++;
#endif
                const int32_t t = it_t + 32 * (int32_t)sub;
                uint32_t r = 0xFFFFu;   // first byte found to differ (none: beyond every limit)
                if (busy && t < (int32_t)it_lim) {
                    const uint4 *cp = it_p + 2 * sub;
                    const bool two = t + 16 < (int32_t)it_lim;   // (a chunk that begins at or behind the limit may lie behind the buffer: not touched)
                    const uint4 cb = cp[0];
                    uint4 cb2 = make_uint4(0, 0, 0, 0);
                    if (two) cb2 = cp[1];
                    const uint32_t wo = it_w + (uint32_t)t, q = wo >> 2, sh = (wo & 3) * 8;
                    const uint32_t d0 = s_win[q], d1 = s_win[q + 1], d2 = s_win[q + 2], d3 = s_win[q + 3], d4 = s_win[q + 4];
                    const uint32_t d5 = s_win[q + 5], d6 = s_win[q + 6], d7 = s_win[q + 7], d8 = s_win[q + 8];
                    uint32_t x0 = cb.x ^ __builtin_amdgcn_alignbit(d1, d0, sh);
                    const uint32_t x1 = cb.y ^ __builtin_amdgcn_alignbit(d2, d1, sh);
                    const uint32_t x2 = cb.z ^ __builtin_amdgcn_alignbit(d3, d2, sh);
                    const uint32_t x3 = cb.w ^ __builtin_amdgcn_alignbit(d4, d3, sh);
                    if (t < 0) x0 &= 0xFFFFFFFFu << (8 * (uint32_t)(-t));   // (bytes before the head's position)
                    const uint32_t m1 = first_set_byte(x0, x1, x2, x3);
                    const uint32_t y0 = cb2.x ^ __builtin_amdgcn_alignbit(d5, d4, sh), y1 = cb2.y ^ __builtin_amdgcn_alignbit(d6, d5, sh);
                    const uint32_t y2 = cb2.z ^ __builtin_amdgcn_alignbit(d7, d6, sh), y3 = cb2.w ^ __builtin_amdgcn_alignbit(d8, d7, sh);
                    const uint32_t m2 = two ? 16u + first_set_byte(y0, y1, y2, y3) : 0x1FFFFFFFu;
                    r = (uint32_t)t + (m1 < m2 ? m1 : m2);
                }
                r = group4_min(r);
                if (gl >= 8) r = half_row_min(r);
                if (gl >= 16) r = row_min(r);
                if (busy) {
                    it_t += (int32_t)(32u << gl_sh);
                    it_p += 2u << gl_sh;
                    if (r < 0xFFFFu || it_t >= (int32_t)it_lim) {   // (no difference found: 0xFFFF, or 0x1FFFFFFF and more)
                        if (sub == 0) q_res[wv][it_id] = (uint16_t)(r < it_lim ? r : it_lim);
                        busy = false;
                    }
                }
            }
#ifdef CAND_STATS
            CAND_STAT(5, n_iter);
            CAND_STAT(16 + (n_iter <= 1 ? 1 : n_iter == 2 ? 2 : n_iter <= 4 ? 3 : n_iter <= 8 ? 4 : n_iter <= 16 ? 5 : 6), 1);
#endif
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (more[k]) ln[k] = dep[k] ? q_res[wv][lead[k]] + (lead[k] & 63) - (uint32_t)lane : q_res[wv][k * 64 + lane];
        }
    }
    CAND_ABL_EXIT(4, cc[0] ^ cc[1] ^ cc[2] ^ cc[3] ^ ln[0] ^ (ln[1] << 4) ^ (ln[2] << 8) ^ (ln[3] << 12) ^ ((uint32_t)fol[0] << 20) ^ ((uint32_t)fol[1] << 21) ^ ((uint32_t)fol[2] << 22) ^ ((uint32_t)fol[3] << 23))
    uint32_t best_len = 0, best_idx = 0;
    bool capped = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t len = ln[k];
        const uint32_t c = cc[k];
        const bool head = prov[k] && !fol[k];   // (a head whose check bits agreed by chance has len 0: its followers measure themselves)
        const uint64_t hm = __ballot(head);
        if (__any(fol[k])) {
            const uint64_t below = hm & lt_mask;
            const int h = below ? 63 - __builtin_clzll(below) : 0;
            const uint32_t hl = __shfl(len, h);
            // LCP(i + t, c + t) = LCP(i, c) - t holds while t < LCP(i, c). Every lane between the head and this one has
            // equal check bits; should the head's turn out a chance hit (< 4 + t bytes), this lane measures for itself.
            if (fol[k]) {
                const uint32_t tt = (uint32_t)(lane - h);
#ifdef CAND_LOWCAP
                if (hl >= CAND_C1) { len = CAND_C1; capped = true; }
                else
#endif
                if (hl >= tt + 4) len = hl - tt;
                else { const uint32_t m = lcp_fwd(s, i, c, 0, cap_total); len = m >= 4 ? m : 0u; }
            }
        }
        if (len) {
            if (len > cap_total) len = cap_total;
            // (ring parse: a candidate that runs to the end of the input is compared past it by the reference's coarse compare,
            // ring/object.rs:39-84, and which candidate wins depends on that: the walkers evaluate such a position exactly)
            if (len == cap_total && (cap_total < max_total || tl.ring)) capped = true;
#ifdef CAND_LOWCAP
            if (len >= CAND_C1 && CAND_C1 < cap_total) capped = true;
#endif
            if (len > best_len) { best_len = len; best_idx = c; }  // ties keep the newest (:226)
        }
    }
    CAND_ABL_EXIT(5, best_len ^ (best_idx << 10) ^ ((uint32_t)capped << 31))
    // backward extension: the same run structure, LCS(i + t, c + t) = LCS(i, c) + t (up to the cap)
    uint32_t r = 0;
    {
        const uint32_t bd = best_len ? i - best_idx : NONE;
        const uint32_t bd_lo = dpp_take<0x138, 0xF>(NONE, bd);
        const bool bfol = best_len != 0 && lane > 0 && bd_lo == bd;
        const uint32_t bmax = best_idx < BCAP ? best_idx : BCAP;   // (BCAP = "at least BCAP": the walkers ask for more)
        uint32_t bw = 0;
        if (best_len && !bfol) {
            if (best_idx >= 16) {
                // The BCAP = 13 bytes in front of the candidate come with ONE dword-aligned 16-byte load that ends at or behind it
                // (round 3: up to two dependent byte-misaligned 8-byte loads in a loop); the position's own 16 bytes, shifted to
                // lie the same way, come from the window in LDS. Byte j of the chunk is byte e - 1 - j back from the candidate
                // (e = 13 .. 16 bytes of the chunk lie in front of it, the rest is masked), so the backward length is the number
                // of equal bytes counted from the chunk's top: v_ffbh_u32 gives -1 for a zero dword and the saturating adds keep it.
                const uintptr_t cpa = ((uintptr_t)(s + best_idx) - BCAP) & ~(uintptr_t)3;
                const uint32_t e = (uint32_t)((uintptr_t)(s + best_idx) - cpa);
                const uint4 cb = *reinterpret_cast<const uint4 *>(cpa);
                const uint32_t wo = 32u + (uint32_t)lane - e, q = wo >> 2, sh = (wo & 3) * 8;
                const uint32_t d0 = win[q], d1 = win[q + 1], d2 = win[q + 2], d3 = win[q + 3], d4 = win[q + 4];
                const uint32_t x0 = cb.x ^ __builtin_amdgcn_alignbit(d1, d0, sh), x1 = cb.y ^ __builtin_amdgcn_alignbit(d2, d1, sh);
                const uint32_t x2 = cb.z ^ __builtin_amdgcn_alignbit(d3, d2, sh);
                uint32_t x3 = cb.w ^ __builtin_amdgcn_alignbit(d4, d3, sh);
                x3 &= 0xFFFFFFFFu >> (8 * (16 - e));   // (e >= 13: at most three bytes of the last dword lie behind the candidate's position)
                const uint32_t lead = min(min(ffbh_m1(x3), add_sat(ffbh_m1(x2), 32u)), min(add_sat(ffbh_m1(x1), 64u), add_sat(ffbh_m1(x0), 96u)));
                bw = (lead >> 3) - (16 - e);         // equal bytes from the chunk's top, less the masked ones (all equal: huge)
            } else {
                while (bw < bmax && s[i - bw - 1] == s[best_idx - bw - 1]) bw++;   // (a candidate in the stream's first 16 bytes)
            }
            bw = bw < bmax ? bw : bmax;
        }
        const uint64_t hm = __ballot(best_len != 0 && !bfol);
        if (__any(bfol)) {
            const uint64_t below = hm & lt_mask;
            const int h = below ? 63 - __builtin_clzll(below) : 0;
            const uint32_t hb = __shfl(bw, h);
            if (bfol) { bw = hb + (uint32_t)(lane - h); if (bw > bmax) bw = bmax; }
        }
        if (valid) {
            // (a length that stopped at the cap is stored as FCAP; one that ends at FCAP exactly is asked for again: harmless)
            if (best_len) r = rec_make(i - best_idx, capped ? FCAP : best_len, bw < BCAP ? bw : BCAP);
            rec[tl.pos_base + i] = r;
        }
    }
#ifdef CAND_PAD_VALU
    // (sensitivity experiment, scripts/cand_pad.sh: CAND_PAD_VALU extra vector instructions per wave -- 1xx: full-rate v_add_u32, 2xx: half-rate v_alignbit_b32)
    {
        uint32_t pz = r;
#pragma unroll
        for (int u = 0; u < CAND_PAD_VALU % 1000; u++) {
            if (CAND_PAD_VALU / 1000 == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(pz) : "v"(best_len));
            else asm volatile("v_alignbit_b32 %0, %0, %1, %1" : "+v"(pz) : "v"(best_len));
        }
        asm volatile("" ::"v"(pz));
    }
#endif
    // has-match bitmap: tile starts are multiples of 64, so a wave covers exactly one word
    const uint64_t bits = __ballot(valid && r != 0);
    // (words past the stream's last position belong to the next stream: never touch them)
    if (lane == 0 && i < tl.start + TILE_POS && i < n_pos) bitmap[(tl.pos_base + i) >> 6] = bits;
}

// ------------------------------------------------------------------------------------ launchers

void launch_enc_chain(const uint8_t *src, const EncTile *tiles, uint32_t n_tiles, uint32_t tile_pos, uint32_t *prev, uint32_t *summary,
                      uint32_t *flist, uint32_t *fcount, uint32_t *redo, bool force_redo, hipStream_t st) {
    if (!n_tiles) return;
    hipLaunchKernelGGL(enc_chain_kernel, dim3(n_tiles), dim3(64 * CH_WAVES), 0, st, src, tiles, n_tiles, tile_pos, prev, summary, flist, fcount, redo,
                       force_redo ? 1u : 0u);
    hipLaunchKernelGGL(enc_chain_ballot_kernel, dim3(n_tiles < 512u ? n_tiles : 512u), dim3(64), 0, st, src, tiles, n_tiles, tile_pos, prev, summary, flist, fcount, redo);
}

void launch_enc_link(const EncTile *tiles, uint32_t n_tiles, uint32_t tile_pos, uint32_t *prev, const uint32_t *summary,
                     const uint32_t *flist, const uint32_t *fcount, hipStream_t st) {
    if (!n_tiles) return;
    hipLaunchKernelGGL(enc_link_kernel, dim3(((n_tiles + 63) / 64) * 64 * ((1u << HASH_BITS) / 256)), dim3(256), 0, st, tiles, n_tiles, tile_pos, prev, summary, flist,
                       fcount);
}

void launch_enc_cand(const uint8_t *src, const EncStream *streams, const EncTile *tiles, uint32_t n_tiles, const uint32_t *prev, uint32_t *rec,
                     uint64_t *bitmap, hipStream_t st) {
    if (!n_tiles) return;
    hipLaunchKernelGGL(enc_cand_kernel, dim3(((n_tiles + 63) / 64) * 64 * CAND_BPT), dim3(256), 0, st, src, streams, tiles, n_tiles, prev, rec, bitmap);
#ifdef CAND_STATS
    {
        unsigned long long h[32], z[32] = {};
        (void)hipStreamSynchronize(st);
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cand_stats), sizeof h) == hipSuccess && hipMemcpyToSymbol(HIP_SYMBOL(g_cand_stats), z, sizeof z) == hipSuccess)
            fprintf(stderr, "cand_stats tiles=%u waves=%llu with_more=%llu more=%llu dep=%llu heads=%llu iters=%llu busy_groups=%llu | heads/wave 0,1,2,3-4,5-8,9-16,17-32,33+: %llu %llu %llu %llu %llu %llu %llu %llu | iters/wave 1,2,3-4,5-8,9-16,17+: %llu %llu %llu %llu %llu %llu\n",
                    n_tiles, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15], h[17], h[18], h[19], h[20], h[21], h[22]);
    }
#endif
}

}  // namespace lzmi
