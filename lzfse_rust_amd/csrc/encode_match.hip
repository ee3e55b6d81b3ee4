// Match finding of the MI355X LZFSE encoder: hand-written HIP kernels for gfx950 (wave64).
//
// The reference keeps a history table of 2^14 buckets x 4 entries, newest first (encode/history.rs:15-118). At
// position i it copies the row of bucket(src[i..i+4]) (the candidate queue), shift-inserts i, and scans the queue
// newest -> oldest (encode/frontend_bytes.rs:183-244). Every position is inserted exactly once and in order
// (frontend_bytes.rs:187,336-344), so the queue a position sees is a pure function of src[0..i+3]:
//
//   enc_table_kernel   replays the table itself. The buckets are cut into 16 partitions; one wave per partition
//                      and span keeps its 1 024 rows (16 KiB) in LDS, scans the span's positions in order, picks
//                      those that hash into its partition and writes, for each, the row it found: the position's
//                      four candidates, 16 bytes. No chains, no dependent hops later on.
//   enc_cand_kernel    one lane per position: distance gate with the reference's *break* (frontend_bytes.rs:
//                      222-224), forward length of every candidate (match_kit/match_fast.rs:22-49), best on forward
//                      length with ties to the newest (:226), backward length of the winner (:61-89), both capped
#include "enc_common.h"

namespace lzmi {

// ------------------------------------------------------------------------------------ history table

// A span is a run of positions of one stream whose rows one set of TB_PARTS waves produces. A long stream is cut into
// several spans so that the serial scan of each stays short; every span but the first of a stream replays the
// TB_WARM positions before it without output. That reproduces the row entries within the match window exactly
// (any candidate a position of the span may use lies < 262 140 positions back, fse/constants.rs:42); older entries
// are missing, which changes nothing: the candidate scan stops at the first entry beyond the window, and an empty
// slot stops it as well (history.rs:72-84: the reference's sentinel fails the same distance test).
constexpr uint32_t TB_ROWS = 1u << (HASH_BITS - TB_PART_BITS);   // rows of one partition
constexpr uint32_t TB_WARM = 262144;
static_assert(TB_WARM >= MAX_D_VALUE && TB_WARM % 64 == 0, "warm-up covers the match window");
constexpr int TB_BATCH = 16;  // 64-position steps whose source values are fetched together

__global__ __launch_bounds__(64) void enc_table_kernel(const uint8_t *__restrict__ src, const EncStream *__restrict__ streams,
                                                       const EncSpan *__restrict__ spans, uint32_t n_spans,
                                                       uint4 *__restrict__ cand4) {
    __shared__ uint4 table[TB_ROWS];       // row = the bucket's four newest positions, newest first
    __shared__ uint32_t ring_pos[128];     // positions of this partition waiting for their table step
    __shared__ uint16_t ring_key[128];     // ... and their row index
    // all partitions of a span on one XCD (workgroups are dealt to the 8 XCDs round-robin): the span's source bytes are
    // fetched into one L2 instead of eight
    const uint32_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const uint32_t span = (slot >> TB_PART_BITS) * 8 + xcd, part = slot & (TB_PARTS - 1);
    if (span >= n_spans) return;
    const EncSpan sp = spans[span];
    const EncStream st = streams[sp.stream];
    const uint8_t *s = src + st.src_off;
    uint4 *out = cand4 + st.pos_base;
    const int lane = e_lane();
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (uint32_t k = lane; k < TB_ROWS; k += 64) table[k] = make_uint4(NONE, NONE, NONE, NONE);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // one table step: the next `cnt` (<= 64) waiting positions, one per lane, in position order
    auto table_step = [&](uint32_t first, uint32_t cnt) {
        const bool have = (uint32_t)lane < cnt;
        const uint32_t e = (first + (uint32_t)lane) & 127u;
        const uint32_t p = ring_pos[e], k = ring_key[e] & (TB_ROWS - 1);
        const uint4 row = table[k];
        // lanes of this step that share my row: they see each other's insertions (history.rs:110-118 push = shift)
        uint64_t same = __ballot(have);
#pragma unroll
        for (int b = 0; b < (int)(HASH_BITS - TB_PART_BITS); b++) {
            const uint64_t bb = __ballot((k >> b) & 1);
            same &= ((k >> b) & 1) ? bb : ~bb;
        }
        const uint64_t lower = same & lt_mask;
        uint4 c = row;
        if (__ballot(have && lower != 0)) {
            // the r lower lanes of my row were inserted before me: they are my newest candidates, nearest lane first
            uint64_t rem = lower;
            uint32_t pp[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int lj = rem ? 63 - __builtin_clzll(rem) : lane;
                pp[j] = __shfl(p, lj);
                if (!rem) pp[j] = NONE;
                rem &= ~(1ull << lj);
            }
            const uint32_t r = (uint32_t)__popcll(lower);
            if (r == 1) c = make_uint4(pp[0], row.x, row.y, row.z);
            else if (r == 2) c = make_uint4(pp[0], pp[1], row.x, row.y);
            else if (r == 3) c = make_uint4(pp[0], pp[1], pp[2], row.x);
            else if (r >= 4) c = make_uint4(pp[0], pp[1], pp[2], pp[3]);
        }
        if (have && p >= sp.begin) out[p] = c;
        if (have && ((same >> lane) >> 1) == 0) table[k] = make_uint4(p, c.x, c.y, c.z);  // the row's last lane writes it back
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    uint32_t n_in = 0, n_done = 0;  // ring counters (uniform): appended, processed
    for (uint32_t pb = sp.first; pb < sp.end; pb += 64 * TB_BATCH) {
        uint32_t vv[TB_BATCH];
#pragma unroll
        for (int j = 0; j < TB_BATCH; j++) {
            const uint32_t q = pb + 64 * j + lane;
            vv[j] = q < sp.end ? ld_u32(s + q) : 0u;
        }
#pragma unroll
        for (int j = 0; j < TB_BATCH; j++) {
            const uint32_t p = pb + 64 * j + lane;
            const uint32_t key = bucket_of(vv[j]);
            const bool own = p < sp.end && (key >> (HASH_BITS - TB_PART_BITS)) == part;
            const uint64_t m = __ballot(own);
            if (m) {
                if (own) {
                    const uint32_t e = (n_in + (uint32_t)__popcll(m & lt_mask)) & 127u;
                    ring_pos[e] = p;
                    ring_key[e] = (uint16_t)key;
                }
                n_in += (uint32_t)__popcll(m);
                if (n_in - n_done >= 64) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    table_step(n_done, 64);
                    n_done += 64;
                }
            }
        }
    }
    if (n_in > n_done) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        table_step(n_done, n_in - n_done);
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

// rec[i] = { dist | bwd << 18 | capped << 31 , fwd_len }  ; fwd_len == 0: no match at i
//
// One lane per position. Step 0 compares the first 16 bytes of every candidate (a candidate is a match iff its first
// four bytes are equal: history.rs Item.val == val). Lanes whose candidates are still equal after that are grouped
// into runs of consecutive positions with the same distance in the same slot (the inside of one long match): only the
// head of a run goes on comparing, the followers derive LCP(i + t, c + t) = LCP(i, c) - t. Heads still equal after
// CAND_C1 bytes are extended by groups of CAND_GL lanes. Repetitive data costs O(1) per position.
constexpr uint32_t CAND_C1 = 64;
constexpr int CAND_GL = 8;  // lanes per group of the long-match work list

__global__ __launch_bounds__(256) void enc_cand_kernel(const uint8_t *__restrict__ src, const EncStream *__restrict__ streams,
                                                       const EncTile *__restrict__ tiles, uint32_t n_tiles, const uint4 *__restrict__ cand4,
                                                       uint2 *__restrict__ rec, uint64_t *__restrict__ bitmap) {
    // Workgroups are handed to the 8 XCDs round-robin. All workgroups of one tile go to the same XCD, so the
    // source bytes a tile gathers from (its own 64 KB + the 256 KB window before it) stay in that XCD's 4 MB L2.
    __shared__ uint32_t q_i[4][256], q_c[4][256];  // per-wave work list of phase 3: position, candidate
    __shared__ uint16_t q_id[4][256], q_res[4][256];
    __shared__ uint32_t q_dist[4][256], q_tab[4][64];  // distance of a listed head; first head per distance hash
    __shared__ uint32_t s_win[4][52];                  // per wave: 192 source bytes around its 64 positions (+ read slack)
    const uint32_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const uint32_t t = (slot / CAND_BPT) * 8 + xcd, bx = slot % CAND_BPT;
    if (t >= n_tiles) return;
    const EncTile tl = tiles[t];
    const EncStream st = streams[tl.stream];
    const uint32_t i = tl.start + bx * blockDim.x + threadIdx.x;
    const uint32_t n = st.n, n_pos = n - 3;
    if (tl.start + bx * blockDim.x >= n_pos) return;  // block-uniform
    const bool valid = i < n_pos && i < tl.start + TILE_POS;
    const uint8_t *s = src + st.src_off;
    const int lane = e_lane();
    const uint4 self = valid ? cand4[st.pos_base + i] : make_uint4(NONE, NONE, NONE, NONE);
    const uint32_t max_total = valid ? n - i : 0;
    const uint32_t cap_total = max_total < FCAP ? max_total : FCAP;
    const uint32_t c1 = cap_total < CAND_C1 ? cap_total : CAND_C1;
    // ---- source window of the wave: bytes [i0 - 32, i0 + 160) of the stream (i0 = position of lane 0) go to LDS
    // with one coalesced load; every lane's own side of the byte compares (forward up to 64 + 16 bytes, backward
    // up to 32) is read from there instead of 64 separate unaligned loads per step ----
    {
        const uint32_t i0 = i - (uint32_t)lane;
        if (lane < 48) {
            const int64_t pos = (int64_t)i0 - 32 + 4 * lane;
            uint32_t wv4 = 0;
            if (pos >= 0 && pos + 4 <= (int64_t)n) wv4 = ld_u32(s + pos);
            else
                for (int k = 0; k < 4; k++)
                    if (pos + k >= 0 && pos + k < (int64_t)n) wv4 |= (uint32_t)s[pos + k] << (8 * k);
            s_win[threadIdx.x >> 6][lane] = wv4;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const uint32_t *win = s_win[threadIdx.x >> 6];
    // ---- phase 1: the candidate queue, newest -> oldest; the scan STOPS at the first entry beyond the window
    // (frontend_bytes.rs:222-224: break, not skip) or at an empty slot ----
    uint32_t cc[4] = {NONE, NONE, NONE, NONE};
    uint32_t ln[4] = {0, 0, 0, 0};
    {
        const uint32_t cq[4] = {self.x, self.y, self.z, self.w};
        bool alive = valid;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (alive && (cq[q] == NONE || i - cq[q] > MAX_D_VALUE)) alive = false;
            if (alive) cc[q] = cq[q];
        }
    }
    // ---- step 0: the first 16 bytes of every candidate, all loads of the step in flight together (each sits alone in
    // its branch and is consumed branch-free afterwards; a load next to its use inside a branch is waited for there).
    // Fewer than 4 equal bytes: not a match (the table entry's value differs). Consecutive positions inside one match
    // see the same distance in the same slot, so their loads fall into the same cache lines. ----
    bool tail[4] = {false, false, false, false};
    {
        const bool room = 16 <= max_total;
        const uint4 zero4 = make_uint4(0, 0, 0, 0);
        uint4 a = zero4, bq[4] = {zero4, zero4, zero4, zero4};
        bool go[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { go[k] = cc[k] != NONE && room; tail[k] = cc[k] != NONE && !room; }
        if (go[0] || go[1] || go[2] || go[3]) {
            const uint32_t wo = 32u + (uint32_t)lane, q = wo >> 2, sh = (wo & 3) * 8;
            const uint32_t d0 = win[q], d1 = win[q + 1], d2 = win[q + 2], d3 = win[q + 3], d4 = win[q + 4];
            a = make_uint4(__builtin_amdgcn_alignbit(d1, d0, sh), __builtin_amdgcn_alignbit(d2, d1, sh),
                           __builtin_amdgcn_alignbit(d3, d2, sh), __builtin_amdgcn_alignbit(d4, d3, sh));
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (go[k]) bq[k] = ld_u128(s + cc[k]);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint64_t lo = ((uint64_t)(a.y ^ bq[k].y) << 32) | (a.x ^ bq[k].x);
            const uint64_t hi = ((uint64_t)(a.w ^ bq[k].w) << 32) | (a.z ^ bq[k].z);
            const uint32_t mlo = lo ? (uint32_t)(__builtin_ctzll(lo | (1ull << 63)) >> 3) : 16u;
            const uint32_t mhi = hi ? 8 + (uint32_t)(__builtin_ctzll(hi | (1ull << 63)) >> 3) : 16u;
            uint32_t m = mlo < mhi ? mlo : mhi;
            m = m < c1 ? m : c1;
            ln[k] = (go[k] && m >= 4) ? m : 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (tail[k]) {  // within 16 bytes of the stream's end
                const uint32_t m = lcp_fwd(s, i, cc[k], 0, c1);
                ln[k] = m >= 4 ? m : 0u;
            }
    }
    // ---- runs: LCP(i + t, c + t) = LCP(i, c) - t as long as every position in between is itself a match with that
    // distance in that slot. Only the first lane of such a run (its head) compares further bytes; the followers derive
    // their length from the head's. ----
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    bool fol[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t dk = ln[k] ? i - cc[k] : NONE;
        const uint32_t dlo = __shfl_up(dk, 1);
        fol[k] = ln[k] != 0 && lane > 0 && dlo == dk;
    }
    // ---- phase 2: heads that matched all 16 bytes go on, 16 bytes per candidate and step, up to CAND_C1 ----
    {
        bool act[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { act[k] = ln[k] == 16 && !fol[k] && c1 > 16 && !tail[k]; tail[k] = false; }
#pragma unroll 1
        for (uint32_t off = 16; off < CAND_C1; off += 16) {
            if (!__any(act[0] || act[1] || act[2] || act[3])) break;
            const bool room = off + 16 <= max_total;
            bool go[4];
#pragma unroll
            for (int k = 0; k < 4; k++) go[k] = act[k] && room;
            const uint4 zero4 = make_uint4(0, 0, 0, 0);
            uint4 a = zero4, bq[4] = {zero4, zero4, zero4, zero4};
            if (go[0] || go[1] || go[2] || go[3]) {
                const uint32_t wo = 32u + (uint32_t)lane + off, q = wo >> 2, sh = (wo & 3) * 8;  // wo + 16 <= 32 + 63 + 48 + 16 < 192
                const uint32_t d0 = win[q], d1 = win[q + 1], d2 = win[q + 2], d3 = win[q + 3], d4 = win[q + 4];
                a = make_uint4(__builtin_amdgcn_alignbit(d1, d0, sh), __builtin_amdgcn_alignbit(d2, d1, sh),
                               __builtin_amdgcn_alignbit(d3, d2, sh), __builtin_amdgcn_alignbit(d4, d3, sh));
            }
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (go[k]) bq[k] = ld_u128(s + cc[k] + off);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint64_t lo = ((uint64_t)(a.y ^ bq[k].y) << 32) | (a.x ^ bq[k].x);
                const uint64_t hi = ((uint64_t)(a.w ^ bq[k].w) << 32) | (a.z ^ bq[k].z);
                // both halves are always consumed, so that neither load can be deferred into a branch
                const uint32_t mlo = lo ? (uint32_t)(__builtin_ctzll(lo | (1ull << 63)) >> 3) : 16u;
                const uint32_t mhi = hi ? 8 + (uint32_t)(__builtin_ctzll(hi | (1ull << 63)) >> 3) : 16u;
                const uint32_t m = mlo < mhi ? mlo : mhi;
                uint32_t nl = off + m;
                const bool stop = m < 16 || nl >= c1;
                nl = nl < c1 ? nl : c1;
                tail[k] = tail[k] || (act[k] && !room);
                ln[k] = go[k] ? nl : ln[k];
                act[k] = go[k] && !stop;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (tail[k]) ln[k] = lcp_fwd(s, i, cc[k], ln[k], c1);  // within 80 bytes of the stream's end
    }
    // ---- phase 3: heads still equal after CAND_C1 bytes go to a per-wave work list and are extended by groups
    // of CAND_GL lanes, 16 bytes per lane and step, several heads at a time (up to FCAP + 64, so that 63
    // followers stay exact up to FCAP); then the followers take head - t ----
    {
        const int wv = threadIdx.x >> 6;
        bool more[4], dep[4];
        uint32_t lead[4];
        uint64_t any_more = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            more[k] = ln[k] == CAND_C1 && !fol[k] && CAND_C1 < cap_total;
            dep[k] = false; lead[k] = 0;
            any_more |= __ballot(more[k]);
        }
        uint32_t total = 0;
        if (any_more) {
            // Heads of the wave with the same distance lie inside one match (they are < 64 positions apart and
            // each is >= 64 long): LCP(i2, i2 - d) = LCP(i0, i0 - d) + i0 - i2. One of them is measured.
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
                    dep[k] = lead[k] != (uint32_t)(k * 64 + lane) && q_dist[wv][lead[k]] == dk;
                }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool ind = more[k] && !dep[k];
                const uint64_t mk = __ballot(ind);
                if (ind) {
                    const uint32_t q = total + (uint32_t)__popcll(mk & lt_mask);
                    q_i[wv][q] = i; q_c[wv][q] = cc[k]; q_id[wv][q] = (uint16_t)(k * 64 + lane);
                }
                total += (uint32_t)__popcll(mk);
            }
        }
        if (total) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int g = lane / CAND_GL, sub = lane % CAND_GL;
            const uint64_t gmask = ((1ull << CAND_GL) - 1) << (g * CAND_GL);
            const uint64_t before = (1ull << (g * CAND_GL)) - 1;
            uint32_t next = 0, it_i = 0, it_c = 0, it_lim = 0, it_off = 0, it_id = 0;
            bool busy = false;
            for (;;) {
                const uint64_t idle = __ballot(!busy && sub == 0);
                if (next < total && idle) {
                    const uint32_t q = next + (uint32_t)__popcll(idle & before);
                    if (!busy && q < total) {
                        busy = true;
                        it_i = q_i[wv][q]; it_c = q_c[wv][q]; it_id = q_id[wv][q];
                        it_off = CAND_C1;
                        const uint32_t maxh = n - it_i;
                        it_lim = maxh < FCAP + 64 ? maxh : FCAP + 64;
                    }
                    next += (uint32_t)__popcll(idle);
                }
                if (!__any(busy)) break;
                const uint32_t o = it_off + 16 * sub;
                uint64_t xl = 0, xh = 0;
                if (busy && o < it_lim) {
                    if (it_i + o + 16 <= n) {
                        const uint4 a = ld_u128(s + it_i + o), bq = ld_u128(s + it_c + o);
                        xl = ((uint64_t)(a.y ^ bq.y) << 32) | (a.x ^ bq.x);
                        xh = ((uint64_t)(a.w ^ bq.w) << 32) | (a.z ^ bq.z);
                    } else {
                        for (uint32_t tt = 0; it_i + o + tt < n; tt++) {
                            const uint64_t x = (uint64_t)(s[it_i + o + tt] ^ s[it_c + o + tt]);
                            if (tt < 8) xl |= x << (8 * tt); else xh |= x << (8 * (tt - 8));
                        }
                    }
                }
                const bool bad = (xl | xh) != 0;
                const uint32_t r = o + (xl ? (uint32_t)(__builtin_ctzll(xl) >> 3) : 8 + (uint32_t)(__builtin_ctzll(xh | (1ull << 63)) >> 3));
                const uint64_t badm = __ballot(bad) & gmask;
                const uint32_t rr = __shfl(r, badm ? __builtin_ctzll(badm) : lane);
                if (busy) {
                    uint32_t res = 0;
                    bool done = false;
                    if (badm) { res = rr < it_lim ? rr : it_lim; done = true; }
                    else {
                        it_off += 16 * CAND_GL;
                        if (it_off >= it_lim) { res = it_lim; done = true; }
                    }
                    if (done) {
                        if (sub == 0) q_res[wv][it_id] = (uint16_t)res;
                        busy = false;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (more[k]) ln[k] = dep[k] ? q_res[wv][lead[k]] + (lead[k] & 63) - (uint32_t)lane : q_res[wv][k * 64 + lane];
        }
    }
    uint32_t best_len = 0, best_idx = 0;
    bool capped = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t len = ln[k];
        const uint32_t c = cc[k];
        const bool head = len != 0 && !fol[k];
        const uint64_t hm = __ballot(head);
        if (__any(fol[k])) {
            const uint64_t below = hm & lt_mask;
            const int h = below ? 63 - __builtin_clzll(below) : 0;
            const uint32_t hl = __shfl(len, h);
            // (every lane between the head and this one is a match with this distance: hl >= lane - h + 4)
            if (fol[k]) len = hl - (uint32_t)(lane - h);
        }
        if (len) {
            if (len > cap_total) len = cap_total;
            if (len == cap_total && cap_total < max_total) capped = true;
            if (len > best_len) { best_len = len; best_idx = c; }  // ties keep the newest (:226)
        }
    }
    // backward extension: the same run structure, LCS(i + t, c + t) = LCS(i, c) + t (up to the cap)
    uint2 r = make_uint2(0, 0);
    {
        const uint32_t bd = best_len ? i - best_idx : NONE;
        const uint32_t bd_lo = __shfl_up(bd, 1);
        const bool bfol = best_len != 0 && lane > 0 && bd_lo == bd;
        const uint32_t bmax = best_idx < BCAP ? best_idx : BCAP;
        uint32_t bw = 0;
        if (best_len && !bfol) {
            // common suffix with this position's side read from the window (8 bytes per step, at most BCAP = 32 back)
            uint32_t len = 0;
            bool open = true;
            while (open && len + 8 <= bmax) {
                const uint32_t wo = 32u + (uint32_t)lane - len - 8, q = wo >> 2, sh = (wo & 3) * 8;
                const uint32_t d0 = win[q], d1 = win[q + 1], d2 = win[q + 2];
                const uint64_t av = (uint64_t)__builtin_amdgcn_alignbit(d1, d0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, sh) << 32);
                const uint64_t x = av ^ ld_u64(s + best_idx - len - 8);
                if (x) { len += (uint32_t)(__builtin_clzll(x) >> 3); open = false; }
                else len += 8;
            }
            while (open && len < bmax && s[i - len - 1] == s[best_idx - len - 1]) len++;
            bw = len;
        }
        const uint64_t hm = __ballot(best_len != 0 && !bfol);
        if (__any(bfol)) {
            const uint64_t below = hm & lt_mask;
            const int h = below ? 63 - __builtin_clzll(below) : 0;
            const uint32_t hb = __shfl(bw, h);
            if (bfol) { bw = hb + (uint32_t)(lane - h); if (bw > bmax) bw = bmax; }
        }
        if (valid) {
            if (best_len) {
                r.x = (i - best_idx) | (bw << 18) | (capped ? REC_CAPPED : 0u);
                r.y = best_len;
            }
            rec[st.pos_base + i] = r;
        }
    }
    // has-match bitmap: tile starts are multiples of 64, so a wave covers exactly one word
    const uint64_t bits = __ballot(valid && r.y != 0);
    // (words past the stream's last position belong to the next stream: never touch them)
    if (lane == 0 && i < tl.start + TILE_POS && i < n_pos) bitmap[(st.pos_base + i) >> 6] = bits;
}

// ------------------------------------------------------------------------------------ launchers

void launch_enc_table(const uint8_t *src, const EncStream *streams, const EncSpan *spans, uint32_t n_spans, uint4 *cand4, hipStream_t st) {
    if (!n_spans) return;
    hipLaunchKernelGGL(enc_table_kernel, dim3(((n_spans + 7) / 8) * 8 * TB_PARTS), dim3(64), 0, st, src, streams, spans, n_spans, cand4);
}

void launch_enc_cand(const uint8_t *src, const EncStream *streams, const EncTile *tiles, uint32_t n_tiles, const uint4 *cand4, uint2 *rec,
                     uint64_t *bitmap, hipStream_t st) {
    if (!n_tiles) return;
    hipLaunchKernelGGL(enc_cand_kernel, dim3(((n_tiles + 7) / 8) * 8 * CAND_BPT), dim3(256), 0, st, src, streams, tiles, n_tiles, cand4, rec, bitmap);
}

}  // namespace lzmi
