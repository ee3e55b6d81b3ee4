// Parallel form of the reference's serial parse (encode/frontend_bytes.rs:160-211 + match_object.rs
// + fse/buffer.rs). The parse is a deterministic state machine over positions with state
//     (index, literal_index, pending match)
// driven by the per-position candidate records of enc_cand_kernel. Two walkers that reach the
// same state evolve identically, so:
//
//   enc_spec_kernel     one LANE per 2 KiB segment walks its segment (plus a 512-byte overrun) from
//                       the guessed state (index = lit = segment start, no pending) and logs every
//                       emitted match together with the state after it
//   enc_stitch_kernel   one wave per stream: the log of segment 0 is exact; for every boundary the
//                       first pair of events of log k / log k+1 with identical state-after is the
//                       hand-over point. If the logs do not meet (or a log aborted on a capped
//                       record) the true walk continues here, scalar, until they do
//   enc_compact_kernel  adopted log ranges + gap events -> ordered match list (lit_pos, L, M, D)
//   enc_segment_kernel  prefix sums of LMD / literal counts; serial only per bvx2 block: binary
//                       search of the last event that fits (10 000 LMDs / 40 000 literals,
//                       fse/buffer.rs:45-97), exact simulation of the boundary event
//   enc_segpar_kernel   the same cut for all blocks of a large stream at once when only the LMD limit closes blocks
//   enc_lmd_kernel      per block: every event writes its LMDs (L > 315 and M > 2 359 splits,
//                       D zeroing against the previous LMD, buffer.rs:99-117)
#include "enc_common.h"

namespace lzmi {

// ------------------------------------------------------------------------------------ shared step

struct WState {
    uint32_t index, lit, p_idx, p_midx, p_len;
};

__device__ __forceinline__ uint32_t next_has(const uint64_t *bm, uint32_t index, uint32_t stop) {
    uint32_t w = index >> 6;
    uint64_t bits = bm[w] & (~0ull << (index & 63));
    while (!bits) {
        w++;
        if ((w << 6) >= stop) return stop;
        bits = bm[w];
    }
    return (w << 6) + (uint32_t)__builtin_ctzll(bits);
}

// the same scan by a whole wave (uniform arguments and result): 64 bitmap words per step, for the stitcher's true
// walk, which crosses long stretches without any candidate on incompressible data
__device__ __forceinline__ uint32_t next_has_wave(const uint64_t *bm, uint32_t index, uint32_t stop) {
    const int lane = e_lane();
    uint32_t w = index >> 6;
    const uint64_t first = bm[w] & (~0ull << (index & 63));
    if (first) return (w << 6) + (uint32_t)__builtin_ctzll(first);
    for (w++; (w << 6) < stop; w += 64) {
        const uint32_t wl = w + (uint32_t)lane;
        const uint64_t bits = (wl << 6) < stop ? bm[wl] : 0ull;
        const uint64_t nz = __ballot(bits != 0);
        if (nz) {
            const int L = __builtin_ctzll(nz);
            const uint32_t lo = e_readlane((uint32_t)bits, L), hi = e_readlane((uint32_t)(bits >> 32), L);
            const uint64_t b = (uint64_t)lo | ((uint64_t)hi << 32);
            return ((w + (uint32_t)L) << 6) + (uint32_t)__builtin_ctzll(b);
        }
    }
    return stop;
}

// Match::select::<40> (match_object.rs:12-33) on an incoming match; returns true on emit
// (without branches: the segment walkers run it for 64 lanes in lockstep, where every branch is a pair of exec-mask instructions
// whichever lanes take it, and a lone walker wave pays ~5 cycles for any instruction)
__device__ __forceinline__ bool select40(WState &st, uint32_t i_idx, uint32_t i_midx, uint32_t i_len,
                                         uint32_t &e_idx, uint32_t &e_midx, uint32_t &e_len) {
    const bool good = i_len >= GOOD_MATCH_LEN;                 // the incoming match is emitted as it is
    const bool none = st.p_len == 0;                           // nothing pending: the incoming one waits
    const bool apart = st.p_idx + st.p_len <= i_idx;           // the pending one ends before the incoming one: emit it, keep the new one
    const bool longer = i_len > st.p_len;                      // they overlap: the longer one is emitted, the other dropped
    const bool emit = good || !none;
    const bool emit_in = good || (!none && !apart && longer);
    const bool keep_in = !good && (none || apart);
    e_idx = emit_in ? i_idx : st.p_idx; e_midx = emit_in ? i_midx : st.p_midx; e_len = emit_in ? i_len : st.p_len;
    st.p_idx = keep_in ? i_idx : st.p_idx; st.p_midx = keep_in ? i_midx : st.p_midx; st.p_len = keep_in ? i_len : 0u;
    return emit;
}

// ------------------------------------------------------------------------------------ speculative walk

// exact forward / backward lengths by the whole wave (512 / 64 bytes per step); every lane must call
__device__ uint32_t st_wave_lcp_fwd(const uint8_t *s, uint32_t a, uint32_t b, uint32_t len, uint32_t max) {
    const int lane = e_lane();
    // bulk: 16 bytes per lane, two steps in flight = 2 KiB per round trip (matches of 100 KB and more exist)
    while (len + 2048 <= max) {
        const uint32_t o0 = len + 16 * lane, o1 = o0 + 1024;
        uint4 a0, b0, a1, b1;
        __builtin_memcpy(&a0, s + a + o0, 16); __builtin_memcpy(&b0, s + b + o0, 16);
        __builtin_memcpy(&a1, s + a + o1, 16); __builtin_memcpy(&b1, s + b + o1, 16);
        const bool bad0 = ((a0.x ^ b0.x) | (a0.y ^ b0.y) | (a0.z ^ b0.z) | (a0.w ^ b0.w)) != 0;
        const bool bad1 = ((a1.x ^ b1.x) | (a1.y ^ b1.y) | (a1.z ^ b1.z) | (a1.w ^ b1.w)) != 0;
        const uint64_t m0 = __ballot(bad0), m1 = __ballot(bad1);
        if (m0) { len += 16 * (uint32_t)__builtin_ctzll(m0); break; }       // the 8-byte loop below finds the byte
        if (m1) { len += 1024 + 16 * (uint32_t)__builtin_ctzll(m1); break; }
        len += 2048;
    }
    while (len < max) {
        uint32_t off = len + 8 * lane;
        uint64_t x = 0;
        if (off + 8 <= max) x = ld_u64(s + a + off) ^ ld_u64(s + b + off);
        else
            for (uint32_t k = 0; off + k < max && k < 8; k++)
                x |= (uint64_t)(s[a + off + k] ^ s[b + off + k]) << (8 * k);
        uint64_t bad = __ballot(x != 0);
        if (bad) {
            int bl = __builtin_ctzll(bad);
            uint32_t xl = e_readlane((uint32_t)x, bl), xh = e_readlane((uint32_t)(x >> 32), bl);
            uint64_t xx = (uint64_t)xl | ((uint64_t)xh << 32);
            uint32_t r = len + 8 * bl + (uint32_t)(__builtin_ctzll(xx) >> 3);
            return r < max ? r : max;
        }
        len += 512;
    }
    return max;
}

__device__ uint32_t st_wave_lcs_bwd(const uint8_t *s, uint32_t a, uint32_t b, uint32_t max) {
    const int lane = e_lane();
    uint32_t len = 0;
    while (len < max) {
        uint32_t off = len + lane;
        bool bad = off < max && s[a - off - 1] != s[b - off - 1];
        uint64_t bm = __ballot(bad);
        if (bm) {
            uint32_t r = len + (uint32_t)__builtin_ctzll(bm);
            return r < max ? r : max;
        }
        len += 64;
    }
    return max;
}

// Forward length of candidate c at position p as the reference's RING encoder measures it, by the whole wave (uniform
// arguments): match_inc_coarse::<4> (ring/object.rs:39-84) looks at 8 bytes, then at 32 bytes per step, and gives up with
// `max` once a step that began at or beyond `max` found nothing -- so it returns the true common length while that is
// below thr = 12 + 32 (K + 1), K = ceil((max - 12) / 32), and `max` otherwise; and it reads the ring, not the input: from
// `tail` on it sees what the ring still holds there, the byte one ring earlier (zeros while the ring has not wrapped: a
// fresh RingBox is zeroed, ring/ring_box.rs:9-17). find_match picks the candidate by THESE lengths (:460-471) and only
// match_short cuts the winner back to max (:479-481).
__device__ uint32_t st_ring_fwd(const uint8_t *s, uint32_t p, uint32_t c, const RingGeo g, uint32_t max) {
    const uint32_t K = max > 12 ? (max - 12 + 31) / 32 : 0;
    const uint32_t thr = 12 + 32 * (K + 1);
    const uint32_t real = g.tail - p;                 // bytes from p on that are the input's own
    const uint32_t lim = thr < real ? thr : real;
    uint32_t len = st_wave_lcp_fwd(s, p, c, 4, lim);
    if (len == lim && lim < thr) {
        // past the tail (thr - lim <= 63 bytes: one lane each)
        const uint32_t q = lim + (uint32_t)e_lane();
        bool bad = false;
        if (q < thr) {
            const uint32_t xa = p + q, xb = c + q;
            const uint8_t va = xa < g.tail ? s[xa] : ((g.wrapped || xa >= RING_SIZE) ? s[xa - RING_SIZE] : (uint8_t)0);
            const uint8_t vb = xb < g.tail ? s[xb] : ((g.wrapped || xb >= RING_SIZE) ? s[xb - RING_SIZE] : (uint8_t)0);
            bad = va != vb;
        }
        const uint64_t bm = __ballot(bad);
        len = bm ? lim + (uint32_t)__builtin_ctzll(bm) : thr;
    }
    return len < thr ? len : max;
}

// find_match's forward part for the ring parse (frontend_ring.rs:453-481) at position p, exact, by the whole wave
__device__ void st_ring_find(const uint8_t *s, const uint32_t *pv, uint32_t ring, uint32_t n, uint32_t p, uint32_t &best_len, uint32_t &best_idx) {
    const RingGeo g = ring_geo(ring, n, p);
    const uint32_t max = g.is_short ? n - p : RING_LONG_MATCH;
    const uint32_t v = ld_u32(s + p);
    uint32_t c = p, d = link_dist(pv[p]);
    best_len = 0; best_idx = 0;
    for (int q = 0; q < 4 && d != 0; q++) {
        c -= d;
        if (p - c > MAX_D_VALUE) break;
        if (ld_u32(s + c) == v) {
            const uint32_t len = st_ring_fwd(s, p, c, g, max);
            if (len > best_len) { best_len = len; best_idx = c; }
        }
        d = link_dist(pv[c]);
    }
    if (g.is_short && best_len > max) best_len = max;
}

// All 64 lanes stay in the loop until every segment of the wave is finished, so that a lane whose
// record is capped can have its exact lengths computed by the whole wave (one request at a time).
// (REPO: the call holds a block of a long slice that is not the last, EncStream::stop -- its walkers leave the match that crosses the
// block's limit to the stitcher; a kernel of its own, so that every other call's walkers carry nothing of it)
// At most three walker waves per SIMD (round 5): a large launch is bound by the rate of its record misses, not by how many walkers
// are resident -- 2.53 ms exclusive at 3 as at 8 waves per SIMD, 2.9 at 2, 4.75 at 1 (profiles/r05_ab_spec_occ.txt) -- and the slots it
// leaves are what another lane's candidate kernel runs in: encode 35.0 -> 35.5 GB/s in the three-lane schedule of the default batch.
#ifndef LZMI_SPEC_OCC
#define LZMI_SPEC_OCC 3
#endif
#define LZMI_SPEC_ATTR __attribute__((amdgpu_waves_per_eu(1, LZMI_SPEC_OCC)))
template <bool STAGED, bool REPO>
__global__ __launch_bounds__(64) LZMI_SPEC_ATTR void enc_spec_kernel(const uint8_t *__restrict__ src, const EncStream *__restrict__ streams,
                                                      const uint2 *__restrict__ segs, uint32_t n_segs, uint32_t seg,
                                                      const uint32_t *__restrict__ prev, const uint32_t *__restrict__ rec,
                                                      const uint64_t *__restrict__ bitmap, SpecEvent *__restrict__ logs,
                                                      SpecHeader *__restrict__ hdrs) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const bool exists = g < n_segs;
    const uint2 sg = exists ? segs[g] : make_uint2(0, 0);
    const EncStream &es = streams[sg.x];
    const uint32_t end = REPO ? walk_end(es) : es.n - 3;
    const uint32_t rel0 = REPO ? es.rel0 : 0u;   // (read once: the loop below stores events, and the compiler would fetch it again at every step)
    const bool to_limit = REPO && es.stop != 0;
    const uint32_t ring = exists ? es.ring : 0u, n_own = es.n;
    const uint32_t S = sg.y * seg, ev_cap = seg_ev_cap(seg);
    const uint32_t stop = exists ? ((S + seg + OVER < end) ? S + seg + OVER : end) : 0;
    const uint32_t *r = rec + es.pos_base;
    const uint64_t *bm = bitmap + (es.pos_base >> 6);
    SpecEvent *ev = logs + (uint64_t)g * ev_cap;
    const int lane = e_lane();
    WState st;
    st.index = exists ? S : 0; st.lit = S; st.p_idx = 0; st.p_midx = 0; st.p_len = 0;
    uint32_t nev = 0, status = 0;
    // Event log through LDS: a lane's events are 16 bytes each at its own place in memory, so storing them one by one
    // costs a write request per event and lane, and the store rate is what this kernel pays for most after the
    // record gathers. A lane collects SPEC_STAGE events in LDS (slot-major: the lanes' events of one slot are adjacent,
    // no bank conflicts); when its stage is full the wave writes those bytes as whole lines, one lane per event.
    // (STAGED = false, small batches: every event is stored at once; such a launch is one round of resident waves whose
    // time is the longest walk, and the flush loops would only lengthen its steps)
#ifndef LZMI_SPEC_STAGE
#define LZMI_SPEC_STAGE 8
#endif
    constexpr uint32_t SPEC_STAGE = LZMI_SPEC_STAGE;
    __shared__ SpecEvent s_stage[STAGED ? SPEC_STAGE : 1][STAGED ? 64 : 1];
    uint32_t n_staged = 0;      // events of this lane in LDS (they follow the nev - n_staged events already in memory)
    auto flush_full = [&](uint64_t who) {
        // every lane in `who` has its stage written out: lanes 0 .. count - 1 move one event each
        while (who) {
            const int L = __builtin_ctzll(who);
            who &= who - 1;
            const uint32_t cnt = e_readlane(n_staged, L), done = e_readlane(nev, L) - cnt;
            const uint64_t eb = ((uint64_t)e_readlane((uint32_t)((uintptr_t)ev >> 32), L) << 32) | e_readlane((uint32_t)(uintptr_t)ev, L);
            SpecEvent *dst = (SpecEvent *)(uintptr_t)eb + done;
            if ((uint32_t)lane < cnt) dst[lane] = s_stage[lane][L];
        }
    };
    uint32_t cw = NONE - 1;  // word index of w0 (w1 is the following word); nothing cached yet
    uint64_t w0 = 0, w1 = 0;
    bool running = exists && st.index < stop;
    while (__any(running)) {
        uint32_t p = 0;
        uint32_t rr = 0;
        bool have = false;
        if (running) {
            // the two bitmap words at the walker's position are kept in registers and refreshed together with
            // the record load, so a step normally costs one memory round trip
            const uint32_t w = st.index >> 6;
            const uint64_t from = ~0ull << (st.index & 63);
            bool hit = false;
            uint32_t slow = st.index;
            if (w == cw) {
                if (w0 & from) { p = (w << 6) + (uint32_t)__builtin_ctzll(w0 & from); hit = true; }
                else if (w1) { p = ((w + 1) << 6) + (uint32_t)__builtin_ctzll(w1); hit = true; }
                else slow = (w + 2) << 6;
            } else if (w == cw + 1) {
                if (w1 & from) { p = (w << 6) + (uint32_t)__builtin_ctzll(w1 & from); hit = true; }
                else slow = (w + 1) << 6;
            }
            if (!hit) p = slow < stop ? next_has(bm, slow, stop) : stop;
            if (p >= stop) { st.index = stop; running = false; }
            else {
                st.index = p; rr = r[p]; have = true;
                if ((p >> 6) != cw) { cw = p >> 6; w0 = bm[cw]; w1 = bm[cw + 1]; }
            }
        }
        uint32_t dist = rec_dist(rr), bw = rec_bwd(rr), fwd = rec_fwd(rr);
        uint32_t midx = p - dist;
        // ---- exact re-evaluation of capped records, one requesting lane at a time, by the whole wave ----
        uint64_t req = __ballot(have && fwd == FCAP);
        while (req) {
            const int L = __builtin_ctzll(req);
            req &= req - 1;
            const uint32_t q_stream = e_readlane(sg.x, L), q_p = e_readlane(p, L);
            const EncStream &qs = streams[q_stream];
            const uint8_t *s = src + qs.src_off;
            const uint32_t *pv = prev + qs.pos_base;
            const uint32_t maxl = qs.n - q_p;
            const uint32_t lim = maxl < XCAP ? maxl : XCAP;
            const uint32_t v = ld_u32(s + q_p);
            uint32_t best_len = 0, best_idx = 0, c = q_p, d = link_dist(pv[q_p]);
            bool over = false;
            for (int q = 0; q < 4 && d != 0; q++) {  // frontend_bytes.rs:214-231
                c -= d;
                if (q_p - c > MAX_D_VALUE) break;
                if (ld_u32(s + c) == v) {
                    uint32_t len = st_wave_lcp_fwd(s, q_p, c, 4, lim);
                    // (ring parse: a candidate that runs to the end of the input is measured past it, st_ring_fwd: the stitcher's)
                    if (len == lim && (lim < maxl || qs.ring)) over = true;
                    if (len > best_len) { best_len = len; best_idx = c; }
                }
                d = link_dist(pv[c]);
            }
            const uint32_t q_hr = best_idx - parse_head(qs.ring, qs.n, q_p, REPO ? qs.rel0 : 0u, best_idx);
            const uint32_t bl = st_wave_lcs_bwd(s, q_p, best_idx, q_hr < BCAP ? q_hr : BCAP);
            if (lane == L) {
                if (over) { status = 1; running = false; have = false; }  // longer than XCAP: left to the stitcher
                else { fwd = best_len; midx = best_idx; dist = q_p - best_idx; bw = bl; }
            }
        }
        // ---- exact backward length when the capped one may be too short (frontend_bytes.rs:259-268) ----
        // backward room: the literals before p, and the bytes between the candidate and the start of the input -- or, for
        // the ring parse, the ring's head (frontend_ring.rs:482)
        const uint32_t hroom = midx - parse_head(ring, n_own, p, rel0, midx);
        const uint32_t room = (p - st.lit) < hroom ? p - st.lit : hroom;
        uint32_t b = bw < room ? bw : room;
        uint64_t reqb = __ballot(have && bw == BCAP && room > BCAP);
        while (reqb) {
            const int L = __builtin_ctzll(reqb);
            reqb &= reqb - 1;
            const uint32_t q_stream = e_readlane(sg.x, L), q_p = e_readlane(p, L), q_m = e_readlane(midx, L), q_room = e_readlane(room, L);
            const uint8_t *s = src + streams[q_stream].src_off;
            const uint32_t bl = st_wave_lcs_bwd(s, q_p, q_m, q_room);
            if (lane == L) b = bl;
        }
        if (have) {
            {
                uint32_t e_idx = 0, e_midx = 0, e_len = 0;
                const uint32_t lit_before = st.lit;
                // a block of a slice that is not the last (EncStream::stop): the match that carries the literal index past the
                // block's limit is the stitcher's to make -- the position it is found at decides which positions the reference
                // never pushed (EncTile::skip_lo), and an event does not say it. Whatever this step could emit that reaches the
                // limit -- the incoming match or the pending one -- ends the walker here, before its state changes.
                if (to_limit && (p + fwd >= end || (st.p_len && st.p_idx + st.p_len >= end))) {
                    st.index = p;
                    status = 1; running = false;
                } else if (select40(st, p - b, midx - b, fwd + b, e_idx, e_midx, e_len)) {
                    {
                    st.lit = e_idx + e_len;
                    st.index = (p + 1 > st.lit) ? p + 1 : st.lit;
                    if (nev < ev_cap) {
                        const SpecEvent h = ev_pack(e_idx, e_len, e_idx - e_midx, lit_before, st.index, st.p_idx, st.p_midx, st.p_len);
                        if (STAGED) s_stage[n_staged++][lane] = h;
                        else ev[nev] = h;
                        nev++;
                    } else nev = ev_cap + 1;   // (log overflow: cannot happen, reported below)
                    }
                } else {
                    st.index = p + 1;
                }
                if (st.index >= stop) running = false;
            }
        }
        if (STAGED) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint64_t full = __ballot(n_staged == SPEC_STAGE);
            if (full) {
                flush_full(full);
                if (n_staged == SPEC_STAGE) n_staged = 0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    if (STAGED) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint64_t rest = __ballot(n_staged != 0);
        if (rest) flush_full(rest);
    }
    if (!exists) return;
    SpecHeader h;
    h.n_events = nev < ev_cap ? nev : ev_cap;
    h.status = nev > ev_cap ? 2u : status;  // 2: log overflow (cannot happen: every emit advances lit by >= 4)
    h.f_index = st.index; h.f_lit = st.lit;
    h.f_pidx = st.p_len ? st.p_idx : 0; h.f_pmidx = st.p_len ? st.p_midx : 0; h.f_plen = st.p_len;
    h.pad = 0;
    hdrs[g] = h;
}

// ------------------------------------------------------------------------------------ stitch

// the walker state after two events is the same (index, literal_index, pending match; with equal indices the pending
// matches are equal when their packed fields are)
__device__ __forceinline__ bool ev_state_eq(const SpecEvent &a, const SpecEvent &b) {
    return ev_index_after(a) == ev_index_after(b) && ev_lit_after(a) == ev_lit_after(b) && a.z == b.z &&
           ((a.y ^ b.y) >> 30) == 0 && ((a.w ^ b.w) >> 27) == 0;
}
// ... and the state after an event is this one
__device__ __forceinline__ bool ev_state_is(const SpecEvent &e, const WState &T) {
    return ev_index_after(e) == T.index && ev_lit_after(e) == T.lit && ev_plen(e) == T.p_len &&
           (T.p_len == 0 || (ev_pidx(e) == T.p_idx && ev_pmidx(e) == T.p_midx));
}

// first event of a log with index_after >= key
__device__ __forceinline__ uint32_t ev_lower_bound(const SpecEvent *ev, uint32_t n, uint32_t key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (ev_index_after(ev[mid]) < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// One thread per segment boundary: the first pair of events of log k (from the start of segment
// k + 1 on) and log k + 1 whose state-after is identical. Independent of every other boundary.
__global__ __launch_bounds__(64) void enc_sync_kernel(const EncStream *__restrict__ streams, const uint2 *__restrict__ segs,
                                                      uint32_t n_segs, uint32_t seg, const SpecEvent *__restrict__ logs,
                                                      const SpecHeader *__restrict__ hdrs, uint4 *__restrict__ sync) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_segs) return;
    const uint2 sg = segs[g];
    uint4 out = make_uint4(0, 0, 0, 0);
    if (sg.y + 1 < streams[sg.x].n_seg) {
        const uint32_t nk = hdrs[g].n_events, nk1 = hdrs[g + 1].n_events;
        const SpecEvent *Lk = logs + (uint64_t)g * seg_ev_cap(seg), *Lk1 = Lk + seg_ev_cap(seg);
        uint32_t i = ev_lower_bound(Lk, nk, (sg.y + 1) * seg), j = 0;
        while (i < nk && j < nk1) {
            const SpecEvent ei = Lk[i], ej = Lk1[j];
            const uint32_t ia = ev_index_after(ei), ja = ev_index_after(ej);
            if (ia < ja) i++;
            else if (ia > ja) j++;
            else if (ev_state_eq(ei, ej)) { out = make_uint4(1, i, j, 0); break; }
            else { i++; j++; }
        }
    }
    sync[g] = out;
}

constexpr uint32_t GS_NO_RESUME = 0x80000000u;   // (positions are 31 bits)
struct Stitch {
    RangeRec *ranges;
    uint32_t n_ranges, range_cap;
    MatchRec *gaps;
    uint4 *gstate;       // per gap event: the walk's state after it (index, pending); GS_NO_RESUME in x: the walk cannot be taken up there
    uint32_t g_marked;   // gap events [g_marked, n_gaps) have no state yet
    uint32_t n_gaps, gap_cap;
    uint32_t out_count;  // matches emitted so far
    uint32_t gap_open;   // first gap event of the currently open gap range
    int status;
    bool writer;         // this wave stores what the functions below record (wave 0 of the stream's workgroup)
};

__device__ __forceinline__ void sx_add_range(Stitch &x, uint32_t kind, uint64_t begin, uint32_t count) {
    if (!count) return;
    if (x.n_ranges >= x.range_cap) { x.status = LZFSE_MI_IO; return; }
    if (e_lane() == 0 && x.writer) {
        RangeRec r;
        r.begin = begin; r.count = count; r.out_off = x.out_count; r.kind = kind;
        x.ranges[x.n_ranges] = r;
    }
    x.n_ranges++;
    x.out_count += count;
}

__device__ __forceinline__ void sx_close_gap(Stitch &x) {
    if (x.n_gaps > x.gap_open) sx_add_range(x, 1, x.gap_open, x.n_gaps - x.gap_open);
    x.gap_open = x.n_gaps;
}

__device__ __forceinline__ void sx_gap_event(Stitch &x, uint32_t lit_before, uint32_t e_idx, uint32_t e_len, uint32_t dist) {
    if (x.n_gaps >= x.gap_cap) { x.status = LZFSE_MI_IO; return; }
    if (e_lane() == 0) {
        MatchRec m;
        m.lit_pos = lit_before; m.l = e_idx - lit_before; m.m = e_len; m.d = dist;
        x.gaps[x.n_gaps] = m;
    }
    x.n_gaps++;
}

// end of a step of the true walk: the last event the step made is one the walk can be resumed behind (a window of a
// stream may be cut there, enc_cut_kernel); an event that the same step followed with another one is not, and keeps
// only where the walk stood when the step was over (not before that)
__device__ __forceinline__ void sx_mark_state(Stitch &x, const WState &T) {
    if (x.g_marked == x.n_gaps || !x.gstate) return;
    if (e_lane() == 0) {
        for (uint32_t g = x.g_marked; g + 1 < x.n_gaps; g++) x.gstate[g] = make_uint4(GS_NO_RESUME | T.index, 0, 0, 0);
        x.gstate[x.n_gaps - 1] = make_uint4(T.index, T.p_len ? T.p_idx : 0, T.p_len ? T.p_midx : 0, T.p_len);
    }
    x.g_marked = x.n_gaps;
}
// the literals a round end pushed (the event just made): the walk stands at the round end B, nothing pending
__device__ __forceinline__ void sx_mark_round(Stitch &x, uint32_t B) {
    if (!x.gstate || x.status) return;
    if (e_lane() == 0) {
        for (uint32_t g = x.g_marked; g + 1 < x.n_gaps; g++) x.gstate[g] = make_uint4(GS_NO_RESUME | B, 0, 0, 0);
        x.gstate[x.n_gaps - 1] = make_uint4(B, 0, 0, 0);
    }
    x.g_marked = x.n_gaps;
}

// One wave per stream; control flow and values are wave-uniform. W > 1 (a call of few large streams, launch_enc_stitch): W waves
// per stream. Following mode then takes 64 * W boundaries a step -- the max-plus maps compose across the waves as they do across
// the lanes, every wave keeps its own copy of the (uniform) walk variables and all copies move alike -- and the true walk, which
// is serial, is wave 0's alone: the others wait for what it leaves. With 512-position segments for such calls (seg_for) the
// walkers of one 64 MiB stream are 2 048 waves instead of 512, each a fifth as long: enc_spec 0.62 -> 0.25 ms, and the four
// times as many boundaries cost the stitcher less than the 32 768 did.
template <int W>
__global__ __launch_bounds__(64 * W) void enc_stitch_kernel(const uint8_t *__restrict__ src, const EncStream *__restrict__ streams,
                                                            uint32_t n_streams, uint32_t seg, const uint32_t *__restrict__ prev,
                                                            const uint32_t *__restrict__ rec, const uint64_t *__restrict__ bitmap,
                                                            const SpecEvent *__restrict__ logs, const SpecHeader *__restrict__ hdrs,
                                                            const uint4 *__restrict__ sync, RangeRec *__restrict__ ranges,
                                                            MatchRec *__restrict__ gaps, uint4 *__restrict__ gstate, EncStreamOut *__restrict__ outs) {
    __shared__ int32_t s_cA[W], s_cB[W];          // a wave's composite map over the boundaries it may take
    __shared__ uint32_t s_nb[W], s_nn[W], s_tot[W], s_st[8];
    const uint32_t si = blockIdx.x;
    if (si >= n_streams) return;
    const int wv = W > 1 ? (int)(threadIdx.x >> 6) : 0;
    const EncStream &es = streams[si];
    const uint8_t *s = src + es.src_off;
    const uint32_t *pv = prev + es.pos_base;
    const uint32_t *r = rec + es.pos_base;
    const uint64_t *bm = bitmap + (es.pos_base >> 6);
    const uint32_t n = es.n, end = walk_end(es), K = es.n_seg;
    // ring parse: the rounds of match_long end at the multiples of RING_BLK in [RING_FIRST_END, t_last]; a round that ends
    // with literals older than the new head pushes them as they are and drops the pending match (frontend_ring.rs:250-272)
    const uint32_t ring = es.ring;
    const bool rounds = ring_rounds(ring, n);
    const uint32_t t_last = rounds ? ring_t_last(n) : 0u;
    const uint32_t ev_cap = seg_ev_cap(seg);
    const SpecEvent *L0 = logs + (uint64_t)es.seg_base * ev_cap;
    const SpecHeader *H0 = hdrs + es.seg_base;
    Stitch x;
    x.ranges = ranges + es.range_base; x.n_ranges = 0; x.range_cap = es.range_cap;
    x.gaps = gaps + es.match_base; x.n_gaps = 0; x.gap_cap = es.match_cap;
    x.gstate = gstate ? gstate + es.match_base : nullptr; x.g_marked = 0;
    x.out_count = 0; x.gap_open = 0; x.status = 0;
    x.writer = wv == 0;
    if (es.n_carry) {
        // the events of the block the front end had not closed when the block before ended (a repo window, EncStream::n_carry): the
        // host has put them in front of the stream's gap events, and they lead its match list
        x.n_gaps = es.n_carry; x.g_marked = es.n_carry;
        sx_close_gap(x);
    }
    uint32_t cross = 0;   // 1 + the position the walk's last match was found at, when that match took the literal index past the end
    uint32_t st_iters = 0, st_syncs = 0, st_fallbacks = 0;
    const uint64_t t_begin = __builtin_amdgcn_s_memtime();

    uint32_t k = 0, a = 0;     // following log k; events [a, ..) of it are true and not yet adopted
    WState T;                  // valid in walking mode
    T.index = 0; T.lit = 0; T.p_idx = 0; T.p_midx = 0; T.p_len = 0;
    bool walking = false, done = false;
    if (es.start) {
        // a window that continues a stream: the true walk goes on from the state the window before was cut at, until a log agrees
        T.index = es.st_index; T.lit = es.st_lit; T.p_idx = es.st_pidx; T.p_midx = es.st_pmidx; T.p_len = es.st_plen;
        walking = true;
        k = T.index / seg < K ? T.index / seg : K - 1;
    }
    // Following mode, 64 boundaries per wave and step (one per lane). With a = first not yet adopted event of the log
    // being followed, a boundary synced at (i, j) hands over at i_eff = max(i, a - 1) and leaves
    // a' = j + (i_eff - i) + 1 = max(j + 1, a + (j - i)): a max-plus map (A, B) = (j + 1, j - i). Such maps
    // compose associatively, (A1, B1) then (A2, B2) = (max(A2, A1 + B2), B1 + B2), so the serial hand-over
    // chain is a wave prefix scan and every lane writes its own range record.
    auto follow = [&]() {
        const int lane = e_lane();
        const uint64_t lt = lane ? (~0ull >> (64 - lane)) : 0ull;
        const uint32_t kw = k + 64u * (uint32_t)wv;       // this wave's first boundary
        uint4 sy_l = make_uint4(0, 0, 0, 0);
        if (kw + lane + 1 < K) sy_l = sync[es.seg_base + kw + lane];  // enc_sync_kernel: (found, i, j)
        const uint64_t nf = __ballot(sy_l.x == 0);
        int nb = nf ? __builtin_ctzll(nf) : 64;  // boundaries before the first one without a sync point
        const bool act0 = lane < nb;
        const int32_t ii = (int32_t)sy_l.y, jj = (int32_t)sy_l.z;
        // (identity of the composition: A = -inf, B = 0; the same six DPP steps as wave_incl_sum, earlier lanes on the left)
        constexpr int32_t NEG = INT32_MIN / 2;
        int32_t A = act0 ? jj + 1 : NEG, B = act0 ? jj - ii : 0;
#define LZMI_MAXPLUS_STEP(CTRL, MASK)                                                                                   \
        {                                                                                                                \
            const int32_t Al = (int32_t)dpp_take<CTRL, MASK>((uint32_t)NEG, (uint32_t)A), Bl = (int32_t)dpp_take<CTRL, MASK>(0u, (uint32_t)B); \
            const int32_t t = Al + B;                                                                                    \
            A = A > t ? A : t; B = Bl + B;                                                                               \
        }
        LZMI_MAXPLUS_STEP(0x111, 0xF) LZMI_MAXPLUS_STEP(0x112, 0xF) LZMI_MAXPLUS_STEP(0x114, 0xF) LZMI_MAXPLUS_STEP(0x118, 0xF)
        LZMI_MAXPLUS_STEP(0x142, 0xA) LZMI_MAXPLUS_STEP(0x143, 0xC)
#undef LZMI_MAXPLUS_STEP
        int32_t a0 = (int32_t)a;       // first not yet adopted event of the log this wave's first boundary hands over from
        uint32_t consumed = (uint32_t)nb;
        bool fail_here = nb < 64;      // the chain stops at a boundary of this step
        if (W > 1) {
            // the waves' composites, in order: a wave takes its boundaries only if every wave before it took all 64 of its own
            if (lane == 0) {
                s_cA[wv] = nb ? (int32_t)e_readlane((uint32_t)A, nb ? nb - 1 : 0) : NEG;
                s_cB[wv] = nb ? (int32_t)e_readlane((uint32_t)B, nb ? nb - 1 : 0) : 0;
                s_nb[wv] = (uint32_t)nb;
            }
            __syncthreads();
            int32_t ac = (int32_t)a;
            bool open = true;          // all waves before this one took 64
            consumed = 0; fail_here = false;
            for (int w = 0; w < W; w++) {
                if (w == wv) { a0 = ac; if (!open) nb = 0; }
                if (!open) continue;
                const uint32_t nbw = s_nb[w];
                if (nbw) { const int32_t t2 = ac + s_cB[w]; ac = s_cA[w] > t2 ? s_cA[w] : t2; }
                consumed += nbw;
                if (nbw < 64) { open = false; fail_here = true; }
            }
            a = (uint32_t)ac;          // (taken over below unless the step is refused)
        }
        const bool act = lane < nb;
        const int32_t a_out = A > a0 + B ? A : a0 + B;
        const int32_t a_prev = (int32_t)dpp_take<0x138, 0xF>(0u, (uint32_t)a_out);   // wave_shr:1
        const int32_t a_in = lane ? a_prev : a0;
        const int32_t i_eff = (a_in > 0 && ii + 1 < a_in) ? a_in - 1 : ii;
        const uint32_t cnt = (act && i_eff >= a_in) ? (uint32_t)(i_eff - a_in + 1) : 0u;
        const uint32_t inc = wave_incl_sum(cnt);
        const uint64_t hm = __ballot(cnt != 0);
        uint32_t n_new = (uint32_t)__popcll(hm), total = e_readlane(inc, 63);
        uint32_t r_base = x.n_ranges, o_base = x.out_count;
        if (W > 1) {
            if (lane == 0) { s_nn[wv] = n_new; s_tot[wv] = total; }
            __syncthreads();
            n_new = 0; total = 0;
            for (int w = 0; w < W; w++) {
                if (w == wv) { r_base += n_new; o_base += total; }
                n_new += s_nn[w]; total += s_tot[w];
            }
            __syncthreads();           // (the arrays are written again by the next step)
        } else if (nb > 0) a = e_readlane((uint32_t)a_out, nb - 1);
        if (consumed > 0) {
            if (x.n_ranges + n_new > x.range_cap) x.status = LZFSE_MI_IO;
            else {
                if (cnt) {
                    RangeRec rr;
                    rr.begin = (uint64_t)(es.seg_base + kw + lane) * ev_cap + (uint32_t)a_in;
                    rr.count = cnt; rr.out_off = o_base + inc - cnt; rr.kind = 0;
                    x.ranges[r_base + (uint32_t)__popcll(hm & lt)] = rr;
                }
                x.n_ranges += n_new;
                x.out_count += total;
                k += consumed;
                st_syncs += consumed;
            }
        }
        if (fail_here && !x.status) {
            // no sync point at boundary k: adopt the rest of log k and continue from its final state
            const SpecHeader hk = H0[k];
            if (hk.n_events > a) sx_add_range(x, 0, (uint64_t)(es.seg_base + k) * ev_cap + a, hk.n_events - a);
            T.index = hk.f_index; T.lit = hk.f_lit; T.p_idx = hk.f_pidx; T.p_midx = hk.f_pmidx; T.p_len = hk.f_plen;
            walking = true;
            st_fallbacks++;
            x.gap_open = x.n_gaps;
        }
    };
    // ---- one step of the true walk (exact, scalar): frontend_bytes.rs:183-208 ----
    auto walk_step = [&]() {
        if (T.index >= end) { done = true; return; }
        uint32_t p = next_has_wave(bm, T.index, end);
        if (rounds) {
            // positions without a candidate are single steps of match_long: every round end B in (T.index, p] is reached
            // with idx == B, the head moves to B - RING/2, and literals below it are pushed (pending dropped)
            const uint32_t pe = p < end ? p : end;
            uint32_t B = ring_round_floor(ring, (T.index / RING_BLK + 1) * RING_BLK);
            for (; B <= pe && B <= t_last && !x.status; B += RING_BLK)
                if (T.lit < B - RING_HALF) {
                    T.p_len = 0;
                    sx_gap_event(x, T.lit, B - RING_HALF, 0, 1);
                    T.lit = B - RING_HALF;
                    sx_mark_round(x, B);
                }
        }
        if (p >= end) { T.index = end; done = true; return; }
        st_iters++;
        T.index = p;
        const uint32_t rr = r[p];
        uint32_t dist = rec_dist(rr), bw = rec_bwd(rr), fwd = rec_fwd(rr);
        uint32_t midx = p - dist;
        const uint32_t head = parse_head(ring, n, p, es.rel0, midx);
        if (fwd == FCAP) {
            // exact forward part of find_match (frontend_bytes.rs:214-231; ring parse: frontend_ring.rs:453-481) by the whole wave
            uint32_t best_len = 0, best_idx = 0;
            if (ring) st_ring_find(s, pv, ring, n, p, best_len, best_idx);
            else {
                uint32_t v = ld_u32(s + p), c = p, d = link_dist(pv[p]);
                for (int q = 0; q < 4 && d != 0; q++) {
                    c -= d;
                    if (p - c > MAX_D_VALUE) break;
                    if (ld_u32(s + c) == v) {
                        uint32_t len = st_wave_lcp_fwd(s, p, c, 4, n - p);
                        if (len > best_len) { best_len = len; best_idx = c; }
                    }
                    d = link_dist(pv[c]);
                }
            }
            fwd = best_len; midx = best_idx; dist = p - midx;
            bw = st_wave_lcs_bwd(s, p, midx, midx - head < BCAP ? midx - head : BCAP);
        }
        const uint32_t hroom = midx - head;
        const uint32_t room = (p - T.lit) < hroom ? p - T.lit : hroom;
        uint32_t b = bw < room ? bw : room;
        if (bw == BCAP && room > BCAP) b = st_wave_lcs_bwd(s, p, midx, room);
        uint32_t e_idx = 0, e_midx = 0, e_len = 0;
        const uint32_t lit_before = T.lit;
        const bool emitted = select40(T, p - b, midx - b, fwd + b, e_idx, e_midx, e_len);
        if (emitted) {
            sx_gap_event(x, lit_before, e_idx, e_len, e_idx - e_midx);
            T.lit = e_idx + e_len;
            if (T.lit >= end) { T.index = end; cross = p + 1; done = true; return; }
            T.index = (p + 1 > T.lit) ? p + 1 : T.lit;
        } else {
            T.index = p + 1;
        }
        if (rounds) {
            // a step (or a match's skip) that carried idx over a round end: the round ends with this idx (:382-392)
            const uint32_t B = ring_round_floor(ring, (p / RING_BLK + 1) * RING_BLK);
            if (B <= T.index && B <= t_last) {
                const uint32_t nh = (T.index & ~(RING_BLK - 1)) - RING_HALF;
                if (T.lit < nh) {
                    T.p_len = 0;
                    sx_gap_event(x, T.lit, nh, 0, 1);
                    T.lit = nh;
                }
            }
        }
        sx_mark_state(x, T);
        if (emitted) {
            // does the log of the segment we are in agree with this state?
            uint32_t kk = T.index / seg;
            if (kk >= K) kk = K - 1;
            if (kk > k) {
                const SpecHeader hh = H0[kk];
                const SpecEvent *Lkk = L0 + (uint64_t)kk * ev_cap;
                uint32_t j = ev_lower_bound(Lkk, hh.n_events, T.index);
                if (j < hh.n_events) {
                    if (ev_state_is(Lkk[j], T)) {
                        sx_close_gap(x);
                        k = kk; a = j + 1;
                        walking = false;
                        st_syncs++;
                    }
                }
            }
        }
    };
    while (!done && !x.status) {
        if (!walking) { follow(); continue; }
        if (W == 1) { walk_step(); continue; }
        // the true walk is wave 0's; the others take over what it leaves
        if (wv == 0) {
            do walk_step(); while (walking && !done && !x.status);
            if (e_lane() == 0) {
                s_st[0] = k; s_st[1] = a; s_st[2] = x.n_ranges; s_st[3] = x.out_count; s_st[4] = (uint32_t)x.status;
                s_st[5] = (walking ? 1u : 0u) | (done ? 2u : 0u);
            }
        }
        __syncthreads();
        k = s_st[0]; a = s_st[1]; x.n_ranges = s_st[2]; x.out_count = s_st[3]; x.status = (int)s_st[4];
        walking = (s_st[5] & 1u) != 0; done = (s_st[5] & 2u) != 0;
        __syncthreads();
    }
    if (wv != 0) return;
    if (!x.status) {
        if (!walking) {
            // left the loop while following: cannot happen (the last segment always ends in walking mode)
            x.status = LZFSE_MI_IO;
        } else if (es.no_flush) {
            // the input goes on (a block of a slice that is not its last): reposition (frontend_bytes.rs:356-367) -- self.index = the
            // limit, or the literal index when a match ran past it; literals that lie more than MAX_MATCH_DISTANCE below it have
            // passed the next block's head: they go as they are and the pending match is dropped. The walk's state leaves with
            // the stream's result below.
            const uint32_t idx_end = T.lit > end ? T.lit : end, nb = idx_end - MAX_D_VALUE;
            if (T.lit < nb) {
                T.p_len = 0;
                sx_gap_event(x, T.lit, nb, 0, 1);
                T.lit = nb;
            }
            sx_close_gap(x);
        } else {
            // flush_pending (frontend_bytes.rs:271-285), then flush_literals (:304-317)
            if (T.p_len != 0) {
                sx_gap_event(x, T.lit, T.p_idx, T.p_len, T.p_idx - T.p_midx);
                T.lit = T.p_idx + T.p_len;
                T.p_len = 0;
            }
            if (n - T.lit) sx_gap_event(x, T.lit, n, 0, 1);
            sx_close_gap(x);
        }
    }
    if (x.gstate && e_lane() == 0)
        for (uint32_t g = x.g_marked; g < x.n_gaps; g++) x.gstate[g] = make_uint4(GS_NO_RESUME | end, 0, 0, 0);   // (the end of the input: nothing to resume)
    if (e_lane() == 0) {
        EncStreamOut o;
        o.n_blocks = 0; o.status = x.status; o.out_len = 0;
        o.n_matches = x.out_count; o.n_ranges = x.n_ranges;
        o.e_lit = T.lit; o.e_pidx = T.p_len ? T.p_idx : 0; o.e_pmidx = T.p_len ? T.p_midx : 0; o.e_plen = T.p_len; o.e_cross = cross; o.e_pad = 0;
        o.iters = st_iters; o.emits = st_syncs; o.capped = st_fallbacks; o.refills = 0;
        o.cycles = __builtin_amdgcn_s_memtime() - t_begin;
        outs[si] = o;
    }
}

// ------------------------------------------------------------------------------------ compaction

// number of LMDs an event expands to when no block limit interferes (fse/buffer.rs:56-97)
__device__ __forceinline__ uint32_t lmd_count_of(uint32_t l, uint32_t m) {
    uint32_t a = l ? (l - 1) / MAX_L_VALUE : 0;
    uint32_t b = m ? (m + MAX_M_VALUE - 1) / MAX_M_VALUE : 1;
    return a + b;
}

// one wave per range slot: copies the range's events into the ordered match list and leaves the
// range-local inclusive prefix sums of (LMD count, literal count) in pc / pl plus the range totals
__global__ __launch_bounds__(64) void enc_compact_kernel(const EncStream *__restrict__ streams, const uint32_t *__restrict__ slot_stream,
                                                         const EncStreamOut *__restrict__ outs, const RangeRec *__restrict__ ranges,
                                                         const SpecEvent *__restrict__ logs, const MatchRec *__restrict__ gaps,
                                                         MatchRec *__restrict__ matches, uint32_t *__restrict__ pc,
                                                         uint32_t *__restrict__ pl, uint2 *__restrict__ rsum) {
    const uint32_t slot = blockIdx.x;
    const uint32_t si = slot_stream[slot];
    const EncStream &es = streams[si];
    const uint32_t ri = slot - es.range_base;
    const EncStreamOut so = outs[si];
    if (so.status || ri >= so.n_ranges) return;
    const RangeRec rg = ranges[slot];
    MatchRec *out = matches + es.match_base + rg.out_off;
    uint32_t *PC = pc + es.match_base + rg.out_off, *PL = pl + es.match_base + rg.out_off;
    const int lane = e_lane();
    const SpecEvent *ev = logs + rg.begin;
    const MatchRec *g = gaps + es.match_base + rg.begin;
    const uint32_t st_skip = es.st_skip;   // (read once: the loop stores, and the compiler would fetch the descriptor again every time round)
    uint32_t carry_c = 0, carry_l = 0;
    for (uint32_t q0 = 0; q0 < rg.count; q0 += 64) {
        const uint32_t q = q0 + lane;
        MatchRec m = {0, 0, 0, 0};
        uint32_t c = 0;
        if (q < rg.count) {
            if (rg.kind == 0) {
                const SpecEvent e = ev[q];
                m.lit_pos = ev_lit(e); m.l = ev_idx(e) - m.lit_pos; m.m = ev_len(e); m.d = ev_dist(e);
            } else {
                m = g[q];
            }
            if (st_skip && rg.out_off + q == 0) {
                // a window that continues a stream: the front of its first event left with the window before
                const uint32_t sk = st_skip, sl = sk < m.l ? sk : m.l;
                m.lit_pos += sk; m.l -= sl; m.m -= sk - sl;
            }
            out[q] = m;
            c = lmd_count_of(m.l, m.m);
        }
        uint32_t ic = wave_incl_sum(c), il = wave_incl_sum(m.l);
        if (q < rg.count) { PC[q] = carry_c + ic; PL[q] = carry_l + il; }
        carry_c += e_readlane(ic, 63);
        carry_l += e_readlane(il, 63);
    }
    if (lane == 0) rsum[slot] = make_uint2(carry_c, carry_l);
}

// one workgroup per stream: exclusive scan of the range totals (a 256-thread workgroup finds room on a busy chip
// much sooner than a 1024-thread one; a stream has one range per 2 KiB segment or fewer)
constexpr int RSCAN_THREADS = 256;
__global__ __launch_bounds__(RSCAN_THREADS) void enc_rscan_kernel(const EncStream *__restrict__ streams, uint32_t n_streams,
                                                                  const EncStreamOut *__restrict__ outs, uint2 *__restrict__ rsum) {
    constexpr int NWV = RSCAN_THREADS / 64;
    __shared__ uint32_t sh[2 * NWV + 2];
    const uint32_t si = blockIdx.x;
    if (si >= n_streams) return;
    const EncStream &es = streams[si];
    const EncStreamOut so = outs[si];
    if (so.status) return;
    uint2 *rs = rsum + es.range_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t carry_c = 0, carry_l = 0;
    for (uint32_t g0 = 0; g0 < so.n_ranges; g0 += RSCAN_THREADS) {
        const uint32_t j = g0 + tid;
        const uint2 v = j < so.n_ranges ? rs[j] : make_uint2(0, 0);
        uint32_t ic = wave_incl_sum(v.x), il = wave_incl_sum(v.y);
        if (lane == 63) { sh[wave] = ic; sh[NWV + wave] = il; }
        __syncthreads();
        uint32_t oc = 0, ol = 0, tc = 0, tl = 0;
        for (int w = 0; w < NWV; w++) {
            uint32_t xx = sh[w], yy = sh[NWV + w];
            if (w < wave) { oc += xx; ol += yy; }
            tc += xx; tl += yy;
        }
        if (j < so.n_ranges) rs[j] = make_uint2(carry_c + oc + ic - v.x, carry_l + ol + il - v.y);  // exclusive
        carry_c += tc; carry_l += tl;
        __syncthreads();
    }
}

// one wave per range slot: range-local prefix sums -> stream-wide prefix sums
__global__ __launch_bounds__(64) void enc_papply_kernel(const EncStream *__restrict__ streams, const uint32_t *__restrict__ slot_stream,
                                                        const EncStreamOut *__restrict__ outs, const RangeRec *__restrict__ ranges,
                                                        const uint2 *__restrict__ rsum, uint32_t *__restrict__ pc, uint32_t *__restrict__ pl) {
    const uint32_t slot = blockIdx.x;
    const uint32_t si = slot_stream[slot];
    const EncStream &es = streams[si];
    const uint32_t ri = slot - es.range_base;
    const EncStreamOut so = outs[si];
    if (so.status || ri >= so.n_ranges) return;
    const RangeRec rg = ranges[slot];
    const uint2 off = rsum[slot];
    uint32_t *PC = pc + es.match_base + rg.out_off, *PL = pl + es.match_base + rg.out_off;
    for (uint32_t q = e_lane(); q < rg.count; q += 64) { PC[q] += off.x; PL[q] += off.y; }
}

// ------------------------------------------------------------------------------------ block segmentation

constexpr int SEGM_THREADS = 64;

struct Emit {  // serial LMD emitter used for the (rare) events that straddle a block boundary
    uint2 *lmds;
    uint32_t lmd_cap, lmd_count;
    uint32_t n_lmd, n_lit, n_match, prev_d;
    int status;
};

__device__ __forceinline__ void em_put(Emit &w, uint32_t l, uint32_t m, uint32_t d) {
    if (w.lmd_count >= w.lmd_cap) { w.status = LZFSE_MI_IO; return; }
    if (e_lane() == 0) w.lmds[w.lmd_count] = make_uint2(l | (m << 16), d);
    w.lmd_count++;
    w.n_lmd++;
}
__device__ __forceinline__ void em_push_l(Emit &w, uint32_t l) { w.prev_d = 1; em_put(w, l, 0, 1); }
__device__ __forceinline__ void em_push_lmd(Emit &w, uint32_t l, uint32_t m, uint32_t d) {
    uint32_t ds = (w.prev_d == d) ? 0u : d;
    w.prev_d = d;
    em_put(w, l, m, ds);
    w.n_match += m;
}
// fse/buffer.rs:45-97
__device__ bool em_buffer_push(Emit &w, uint32_t &n_lit, uint32_t &match_len, uint32_t d) {
    while (n_lit > MAX_L_VALUE) {
        if (w.n_lmd == LMDS_PER_BLOCK) return false;
        uint32_t limit = LITERALS_PER_BLOCK - w.n_lit;
        if (MAX_L_VALUE <= limit) { w.n_lit += MAX_L_VALUE; n_lit -= MAX_L_VALUE; em_push_l(w, MAX_L_VALUE); }
        else if (limit != 0) { w.n_lit += limit; n_lit -= limit; em_push_l(w, limit); return false; }
        else return false;
    }
    if (w.n_lmd == LMDS_PER_BLOCK) return false;
    uint32_t literal_len = n_lit;
    uint32_t limit = LITERALS_PER_BLOCK - w.n_lit;
    if (literal_len <= limit) { w.n_lit += literal_len; n_lit = 0; }
    else if (limit != 0) { w.n_lit += limit; n_lit -= limit; em_push_l(w, limit); return false; }
    else return false;
    while (match_len > MAX_M_VALUE) {
        em_push_lmd(w, literal_len, MAX_M_VALUE, d);
        match_len -= MAX_M_VALUE;
        literal_len = 0;
        if (w.n_lmd == LMDS_PER_BLOCK) return false;
    }
    em_push_lmd(w, literal_len, match_len, d);
    match_len = 0;
    return true;
}

// One wave per stream cuts the event list into bvx2 blocks, using the stream-wide inclusive prefix sums of
// per-event LMD and literal counts (enc_compact / enc_rscan / enc_papply). Blocks follow each other serially (a
// block starts where the previous one ended, remainder of the boundary event included); all values are
// wave-uniform, the lanes only share the search for the last event that fits (64 probes per step).
__global__ __launch_bounds__(SEGM_THREADS) void enc_segment_kernel(const EncStream *__restrict__ streams, uint32_t n_streams,
                                                                  const MatchRec *__restrict__ matches, const uint32_t *__restrict__ pc,
                                                                  const uint32_t *__restrict__ pl, uint2 *__restrict__ lmds,
                                                                  EncBlock *__restrict__ blocks, EncStreamOut *__restrict__ outs,
                                                                  const uint32_t *__restrict__ flags) {
    const uint32_t si = blockIdx.x;
    if (si >= n_streams) return;
    if (flags[si] == 2u) return;   // cut by enc_segpar_kernel / enc_segfin_kernel
    const int lane = e_lane();
    const EncStream &es = streams[si];
    EncStreamOut so = outs[si];
    if (so.status) return;
    const uint32_t E = so.n_matches;
    const MatchRec *mt = matches + es.match_base;
    const uint32_t *PC = pc + es.match_base, *PL = pl + es.match_base;  // inclusive prefix sums
    // ---- serial over blocks ----
    Emit w;
    w.lmds = lmds + es.lmd_base; w.lmd_cap = es.lmd_cap; w.lmd_count = 0;
    w.n_lmd = 0; w.n_lit = 0; w.n_match = 0; w.prev_d = 0; w.status = 0;
    EncBlock *bk = blocks + es.blk_base;
    uint32_t n_blk = 0;
    uint64_t stage_used = 0;
    uint32_t j = 0;             // next unprocessed event
    bool rem = false;           // remainder of a boundary event pending
    uint32_t rem_l = 0, rem_m = 0, rem_d = 0;
    uint32_t raw_pos = es.start ? es.st_raw : 0u;   // first raw byte of the current block (a window that continues a stream: where it takes the stream up)
    uint32_t rem_ev = 0, rem_all = 0;   // the event `rem` is what is left of, and its l + m
    bool more = true;
    // A block normally costs ONE memory round trip: the 64 lanes that probe the prefix sums just below the expected end
    // of the block also fetch the events there; the last complete event, the boundary event, the first event of the next
    // block and the prefix sums in front of it are then picked out of registers (carried into the next iteration).
    bool c_base = false, c_start = false;          // carried: PC / PL at j - 1, lit_pos of event j
    uint32_t c_base_c = 0, c_base_l = 0, c_start_pos = 0;
    while (more && !w.status) {
        const uint32_t blk_lmd_start = w.lmd_count;
        w.n_lmd = 0; w.n_lit = 0; w.n_match = 0; w.prev_d = 0;
        bool full = false;
        if (rem) {
            if (em_buffer_push(w, rem_l, rem_m, rem_d)) rem = false; else full = true;
        }
        uint32_t ev_begin = j, ev_end = j, head_lmds = w.n_lmd, head_prev_d = w.prev_d;
        uint32_t cut_ev = NONE, cut_skip = 0;     // the event the next block begins with, and how much of it is behind
        if (full) { cut_ev = rem_ev; cut_skip = rem_all - (rem_l + rem_m); }
        if (!full) {
            // events [j, j2) fit completely: n_lmd + sum c <= 10 000 and n_lit + sum l <= 40 000
            const uint32_t base_c = j ? (c_base ? c_base_c : PC[j - 1]) : 0, base_l = j ? (c_base ? c_base_l : PL[j - 1]) : 0;
            const uint32_t j_first = j;
            const bool had_start = c_start;
            const uint32_t start_carried = c_start_pos;
            c_base = false; c_start = false;
            uint32_t w0 = 0;                // first event of the probed window
            bool probed = false;
            uint32_t pr_c = 0, pr_l = 0;    // PC / PL of this lane's probe
            MatchRec pr_m;                  // ... and its event
            pr_m.lit_pos = 0; pr_m.l = 0; pr_m.m = 0; pr_m.d = 0;
            const uint32_t room_c = LMDS_PER_BLOCK - w.n_lmd, room_l = LITERALS_PER_BLOCK - w.n_lit;
            // first event that does NOT fit: in [j, hi] with hi <= j + 10 001 (every event has at least one LMD);
            // fits(e) is monotone, 64 probes per step narrow [lo, hi] by a factor of 65
            uint32_t lo = j, hi = E - j > LMDS_PER_BLOCK + 1 ? j + LMDS_PER_BLOCK + 1 : E;
            // nearly every event is one LMD, so the answer is usually just below j + room_c: one dense probe of the
            // 64 events below that point settles most blocks in a single round trip
            if (hi - lo > 128) {
                const uint32_t top = (j + room_c + 1 < hi) ? j + room_c + 1 : hi;  // events >= top cannot fit (top <= hi)
                if (top - lo > 64) {
                    const uint32_t e = top - 64 + (uint32_t)lane;
                    w0 = top - 64; probed = true;
                    pr_c = PC[e]; pr_l = PL[e]; pr_m = mt[e];
                    const bool fits = pr_c - base_c <= room_c && pr_l - base_l <= room_l;
                    const uint32_t nfit = (uint32_t)__popcll(__ballot(fits));  // monotone: the first nfit probes fit
                    if (nfit == 0) hi = top - 64;                               // answer below the probed range
                    else { lo = top - 64 + nfit; hi = nfit < 64 ? lo : top; }   // exact when a probe failed
                }
            }
            while (lo < hi) {
                const uint32_t span = hi - lo, stp = span / 65 + 1;
                const uint32_t e = lo + stp * (uint32_t)lane + (stp - 1);  // probes lo+stp-1, lo+2stp-1, ...
                const bool in = e < hi;
                const bool fits = in && PC[e] - base_c <= room_c && PL[e] - base_l <= room_l;
                const uint64_t fm = __ballot(fits), im = __ballot(in);
                const uint32_t nfit = (uint32_t)__popcll(fm);  // monotone: the fitting probes are the first nfit
                // all events up to probe nfit-1 fit; probe nfit (if any) does not
                const uint32_t new_lo = lo + stp * nfit;
                const uint32_t new_hi = nfit < (uint32_t)__popcll(im) ? lo + stp * nfit + (stp - 1) : hi;
                lo = new_lo; hi = new_hi;
            }
            const uint32_t j2 = lo;
            ev_end = j2;
            // events j2 - 1, j2 (and j2 + 1) in the probed window: taken from the lanes that fetched them
            const bool in_win = probed && j2 > w0 && j2 < w0 + 64;
            const int la = in_win ? (int)(j2 - 1 - w0) : 0;
            if (j2 > j) {
                MatchRec last;
                uint32_t pc_last, pl_last;
                if (in_win) {
                    pc_last = e_readlane(pr_c, la); pl_last = e_readlane(pr_l, la);
                    last.lit_pos = e_readlane(pr_m.lit_pos, la); last.l = e_readlane(pr_m.l, la);
                    last.m = e_readlane(pr_m.m, la); last.d = e_readlane(pr_m.d, la);
                } else { pc_last = PC[j2 - 1]; pl_last = PL[j2 - 1]; last = mt[j2 - 1]; }
                const uint32_t dc = pc_last - base_c, dl = pl_last - base_l;
                const uint32_t end_pos = last.lit_pos + last.l + last.m;
                // raw bytes of the full events = end_pos - (start of event j's literals)
                const uint32_t start_pos = had_start ? start_carried : mt[j_first].lit_pos;
                w.n_lmd += dc; w.n_lit += dl; w.n_match += (end_pos - start_pos) - dl;
                w.lmd_count += dc;  // written by enc_lmd_kernel
                w.prev_d = last.d;  // every event ends with push_lmd(.., d)
                if (w.lmd_count > w.lmd_cap) { w.status = LZFSE_MI_IO; break; }
            }
            j = j2;
            if (j < E) {
                // boundary event: fills the block (by construction it cannot complete)
                MatchRec m;
                if (in_win) {
                    m.lit_pos = 0; m.l = e_readlane(pr_m.l, la + 1); m.m = e_readlane(pr_m.m, la + 1); m.d = e_readlane(pr_m.d, la + 1);
                    // the next block starts behind it: prefix sums at the boundary event, position of the event after it
                    c_base = true; c_base_c = e_readlane(pr_c, la + 1); c_base_l = e_readlane(pr_l, la + 1);
                    if (la + 2 < 64) { c_start = true; c_start_pos = e_readlane(pr_m.lit_pos, la + 2); }
                } else m = mt[j];
                rem_l = m.l; rem_m = m.m; rem_d = m.d;
                j++;
                if (em_buffer_push(w, rem_l, rem_m, rem_d)) rem = false;
                else {
                    rem = true; full = true;
                    rem_ev = j - 1; rem_all = m.l + m.m;
                    cut_ev = rem_ev; cut_skip = rem_all - (rem_l + rem_m);
                }
            }
        }
        if (!full && j >= E && !rem) more = false;  // final block (fse/backend.rs:92-95)
        // close the block
        if (n_blk >= es.blk_cap) { w.status = LZFSE_MI_IO; break; }
        const uint32_t need = stage_need(w.n_lit, w.n_lmd);
        if (stage_used + need > es.stage_cap) { w.status = LZFSE_MI_IO; break; }
        EncBlock b;
        b.lmd_start = es.lmd_base + blk_lmd_start;
        b.stage_off = es.stage_base + stage_used;
        b.src_start = raw_pos;
        b.n_lmd = w.n_lmd; b.n_lit = w.n_lit; b.n_match = w.n_match;
        b.hdr_len = 0; b.lit_len = 0; b.lmd_len = 0; b.cut_ev = cut_ev; b.cut_skip = cut_skip; b.pad = 0;
        b.ev_begin = ev_begin; b.ev_end = ev_end; b.head_lmds = head_lmds; b.head_prev_d = head_prev_d;
        if (lane == 0) bk[n_blk] = b;
        n_blk++;
        stage_used += need;
        raw_pos += w.n_lit + w.n_match;
    }
    so.n_blocks = n_blk;
    so.status = w.status;
    if (lane == 0) outs[si] = so;
}

// ------------------------------------------------------------------------------------ block segmentation, all blocks at once
//
// The serial kernel above spends about a microsecond per block, which is a third of the encoder's time for ONE large stream
// (860 blocks in 64 MiB of text). When the 10 000-LMD limit is what closes every block -- the 40 000-literal limit never binds
// -- the cut is no chain at all: Buffer::push (fse/buffer.rs:45-97) hands out the LMDs of the events in order, so block k is
// the LMDs [10 000 k, 10 000 k + 10 000) of the stream-wide LMD sequence, and everything the serial walk would leave in the
// block record follows from the prefix sums by two binary searches: the event that holds the block's first LMD (its LMDs in
// this block are the "head", written here, exactly as the serial emitter writes the remainder of a boundary event, or the
// whole event when the boundary fell between two events), the complete events behind it (enc_lmd_kernel), and the leading
// LMDs of the event that holds the block's last LMD when that event goes on into the next block. One wave per block slot
// does this for every block of a large stream at once; a second kernel checks that no block came out with more than 40 000
// literals (then no literal limit was ever met inside Buffer::push and the result is the serial one) and turns the per-block
// sizes into offsets. Any stream for which that does not hold is left to the serial kernel, untouched.

constexpr uint32_t SEGPAR_MIN_EVENTS = 200000;   // smaller streams: the serial cut is a few dozen microseconds

// LMDs of one event (l literals, match m at distance d) in Buffer::push order: nL x (315, 0, 1), then nM match LMDs of which
// the first carries the remaining literals
struct EvShape {
    uint32_t nL, l_rest, nM, m_last;
};
__device__ __forceinline__ EvShape ev_shape(uint32_t l, uint32_t m) {
    EvShape s;
    s.nL = l ? (l - 1) / MAX_L_VALUE : 0;
    s.l_rest = l - s.nL * MAX_L_VALUE;
    s.nM = m ? (m - 1) / MAX_M_VALUE + 1 : 1;
    s.m_last = m - (s.nM - 1) * MAX_M_VALUE;
    return s;
}

// LMDs [i0, i1) of event ev go to out[0 ..), the D of the first one against prev_d (0 at the start of a block); returns the
// literals and match bytes they carry and the prev_d behind them. All lanes call with uniform arguments.
__device__ void ev_emit(const MatchRec &ev, uint32_t i0, uint32_t i1, uint32_t prev_d, uint2 *out, uint32_t &lit, uint32_t &mat, uint32_t &prev_out) {
    const EvShape s = ev_shape(ev.l, ev.m);
    const int lane = e_lane();
    uint32_t a_l = 0, a_m = 0;
    for (uint32_t i = i0 + (uint32_t)lane; i < i1; i += 64) {
        uint32_t L, M, D;
        if (i < s.nL) { L = MAX_L_VALUE; M = 0; D = 1; }
        else {
            const uint32_t j = i - s.nL;
            L = j == 0 ? s.l_rest : 0u;
            M = j + 1 < s.nM ? MAX_M_VALUE : s.m_last;
            const uint32_t pd = i == i0 ? prev_d : (i - 1 < s.nL ? 1u : ev.d);
            D = pd == ev.d ? 0u : ev.d;
        }
        out[i - i0] = make_uint2(L | (M << 16), D);
        a_l += L; a_m += M;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { a_l += __shfl_xor(a_l, d); a_m += __shfl_xor(a_m, d); }
    lit = a_l; mat = a_m;
    prev_out = i1 > i0 ? (i1 - 1 < s.nL ? 1u : ev.d) : prev_d;
}

// smallest e in [0, E) with PC[e] > x (PC inclusive, non-decreasing; x < PC[E - 1])
__device__ __forceinline__ uint32_t first_above(const uint32_t *PC, uint32_t E, uint32_t x) {
    uint32_t lo = 0, hi = E - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (PC[mid] > x) hi = mid; else lo = mid + 1;
    }
    return lo;
}

__global__ __launch_bounds__(64) void enc_segpar_kernel(const EncStream *__restrict__ streams, const uint32_t *__restrict__ slot_stream,
                                                        uint32_t n_slots, const EncStreamOut *__restrict__ outs,
                                                        const MatchRec *__restrict__ matches, const uint32_t *__restrict__ pc,
                                                        const uint32_t *__restrict__ pl, uint2 *__restrict__ lmds,
                                                        EncBlock *__restrict__ blocks, uint32_t *__restrict__ flags) {
    const uint32_t slot = blockIdx.x;
    if (slot >= n_slots) return;
    const uint32_t si = slot_stream[slot];
    const EncStream &es = streams[si];
    const uint32_t k = slot - es.blk_base;
    const EncStreamOut so = outs[si];
    const uint32_t E = so.n_matches;
    if (so.status || E < SEGPAR_MIN_EVENTS) return;
    const MatchRec *mt = matches + es.match_base;
    const uint32_t *PC = pc + es.match_base, *PL = pl + es.match_base;
    const uint32_t total = PC[E - 1];
    const uint32_t nb = (total + LMDS_PER_BLOCK - 1) / LMDS_PER_BLOCK;
    if (k >= nb) return;
    const uint32_t B0 = k * LMDS_PER_BLOCK, B1 = total - B0 > LMDS_PER_BLOCK ? B0 + LMDS_PER_BLOCK : total;
    const uint32_t n_lmd = B1 - B0;
    uint2 *out = lmds + es.lmd_base + B0;
    // the event of the block's first LMD, and the event of its last
    const uint32_t e_first = first_above(PC, E, B0);
    const uint32_t e_last = first_above(PC, E, B1 - 1);
    const uint32_t before_first = e_first ? PC[e_first - 1] : 0, before_last = e_last ? PC[e_last - 1] : 0, pc_last = PC[e_last];
    uint32_t n_lit = 0, n_match = 0;
    uint32_t head_lmds = 0, head_prev_d = 0, ev_begin = 0;
    if (k > 0) {
        // head: what is left of event e_first (all of it when the block before ended between two events)
        const MatchRec ev = mt[e_first];
        const uint32_t off0 = B0 - before_first, c = PC[e_first] - before_first;
        head_lmds = c - off0 < n_lmd ? c - off0 : n_lmd;
        uint32_t hl, hm;
        ev_emit(ev, off0, off0 + head_lmds, 0u, out, hl, hm, head_prev_d);
        n_lit += hl; n_match += hm;
        ev_begin = e_first + 1;
    }
    // complete events behind the head
    uint32_t ev_end = pc_last == B1 ? e_last + 1 : e_last;
    if (ev_end < ev_begin) ev_end = ev_begin;
    if (ev_end > ev_begin) {
        const MatchRec last = mt[ev_end - 1];
        const uint32_t body_l = PL[ev_end - 1] - (ev_begin ? PL[ev_begin - 1] : 0);
        const uint32_t body_raw = last.lit_pos + last.l + last.m - mt[ev_begin].lit_pos;
        n_lit += body_l; n_match += body_raw - body_l;
    }
    // tail: the leading LMDs of an event that goes on into the next block (unless that event is this block's head)
    if (pc_last > B1 && e_last >= ev_begin) {
        const MatchRec ev = mt[e_last];
        const uint32_t pd = e_last > ev_begin ? mt[e_last - 1].d : head_prev_d;
        uint32_t tl, tm, unused;
        ev_emit(ev, 0u, B1 - before_last, pd, out + (before_last - B0), tl, tm, unused);
        n_lit += tl; n_match += tm;
    }
    if (n_lit > LITERALS_PER_BLOCK) atomicOr(&flags[si], 1u);   // the literal limit closes a block somewhere: not this kernel's case
    if (e_lane() == 0) {
        EncBlock b;
        b.lmd_start = es.lmd_base + B0;
        b.stage_off = stage_need(n_lit, n_lmd);   // (sizes; enc_segfin_kernel turns them into offsets)
        b.src_start = n_lit + n_match;
        b.n_lmd = n_lmd; b.n_lit = n_lit; b.n_match = n_match;
        b.hdr_len = 0; b.lit_len = 0; b.lmd_len = 0;
        // what the next block begins with: the event after e_last when the block's last LMD is e_last's last, else e_last
        // itself behind the bytes of its first B1 - before_last LMDs (315 literals each, then the rest of the literals with
        // the first 2 359 match bytes, then 2 359 match bytes each: fse/buffer.rs:56-97)
        b.cut_ev = NONE; b.cut_skip = 0; b.pad = 0;
        if (pc_last == B1) { if (e_last + 1 < E) b.cut_ev = e_last + 1; }
        else {
            const MatchRec ev = mt[e_last];
            const EvShape sh = ev_shape(ev.l, ev.m);
            const uint32_t c = B1 - before_last;
            b.cut_ev = e_last;
            b.cut_skip = c <= sh.nL ? c * MAX_L_VALUE : ev.l + (c - sh.nL) * MAX_M_VALUE;
        }
        b.ev_begin = ev_begin; b.ev_end = ev_end; b.head_lmds = head_lmds; b.head_prev_d = head_prev_d;
        blocks[slot] = b;
    }
}

// one wave per stream: the blocks' sizes become offsets; flags[si] = 2 tells the serial kernel that the stream is done
__global__ __launch_bounds__(64) void enc_segfin_kernel(const EncStream *__restrict__ streams, uint32_t n_streams,
                                                        const uint32_t *__restrict__ pc, EncBlock *__restrict__ blocks,
                                                        EncStreamOut *__restrict__ outs, uint32_t *__restrict__ flags) {
    const uint32_t si = blockIdx.x;
    if (si >= n_streams) return;
    const EncStream &es = streams[si];
    EncStreamOut so = outs[si];
    const uint32_t E = so.n_matches;
    if (so.status || E < SEGPAR_MIN_EVENTS || flags[si]) return;
    const uint32_t total = (pc + es.match_base)[E - 1];
    const uint32_t nb = (total + LMDS_PER_BLOCK - 1) / LMDS_PER_BLOCK;
    if (nb > es.blk_cap || total > es.lmd_cap) return;   // (the serial kernel reports it)
    EncBlock *bk = blocks + es.blk_base;
    const int lane = e_lane();
    uint64_t stage_used = 0;
    uint32_t raw_pos = es.start ? es.st_raw : 0u;
    for (uint32_t b0 = 0; b0 < nb; b0 += 64) {
        const uint32_t b = b0 + (uint32_t)lane;
        const bool in = b < nb;
        const uint32_t need = in ? (uint32_t)bk[b].stage_off : 0u, raw = in ? bk[b].src_start : 0u;
        uint32_t in_need = need, in_raw = raw;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t un = __shfl_up(in_need, d), ur = __shfl_up(in_raw, d);
            if (lane >= d) { in_need += un; in_raw += ur; }
        }
        if (in) { bk[b].stage_off = es.stage_base + stage_used + (in_need - need); bk[b].src_start = raw_pos + (in_raw - raw); }
        stage_used += e_readlane(in_need, 63);
        raw_pos += e_readlane(in_raw, 63);
    }
    if (stage_used > es.stage_cap) return;   // (offsets are garbage then, and the serial kernel overwrites them)
    if (lane == 0) {
        so.n_blocks = nb; so.status = 0;
        outs[si] = so;
        flags[si] = 2u;
    }
}

// ------------------------------------------------------------------------------------ LMD writing

// One workgroup per block slot: the block's complete events write their LMDs
// (fse/buffer.rs:56-117 without block limits, which enc_segment_kernel resolved).
__global__ __launch_bounds__(256) void enc_lmd_kernel(const EncStream *__restrict__ streams, const uint32_t *__restrict__ slot_stream,
                                                      const EncStreamOut *__restrict__ outs, const EncBlock *__restrict__ blocks,
                                                      const MatchRec *__restrict__ matches, const uint32_t *__restrict__ pc,
                                                      uint2 *__restrict__ lmds) {
    const uint32_t slot = blockIdx.x;
    const uint32_t si = slot_stream[slot];
    const EncStream &es = streams[si];
    const uint32_t bi = slot - es.blk_base;
    const EncStreamOut so = outs[si];
    if (so.status || bi >= so.n_blocks) return;
    const EncBlock b = blocks[slot];
    if (b.ev_end <= b.ev_begin) return;
    const MatchRec *mt = matches + es.match_base;
    const uint32_t *PC = pc + es.match_base;
    uint2 *out = lmds + b.lmd_start + b.head_lmds;
    const uint32_t base_c = b.ev_begin ? PC[b.ev_begin - 1] : 0;
    for (uint32_t j = b.ev_begin + threadIdx.x; j < b.ev_end; j += blockDim.x) {
        const MatchRec m = mt[j];
        uint32_t o = (j ? PC[j - 1] : 0) - base_c;
        uint32_t prev_d = (j == b.ev_begin) ? b.head_prev_d : mt[j - 1].d;
        uint32_t l = m.l, mm = m.m;
        while (l > MAX_L_VALUE) {  // push_l: (315, 0, 1), prev_d = 1
            out[o++] = make_uint2(MAX_L_VALUE, 1);
            l -= MAX_L_VALUE;
            prev_d = 1;
        }
        while (mm > MAX_M_VALUE) {
            out[o++] = make_uint2(l | (MAX_M_VALUE << 16), prev_d == m.d ? 0u : m.d);
            prev_d = m.d;
            mm -= MAX_M_VALUE;
            l = 0;
        }
        out[o] = make_uint2(l | (mm << 16), prev_d == m.d ? 0u : m.d);
    }
}

// ------------------------------------------------------------------------------------ cut of a window
//
// The stream encoder (lzfse_mi_estream_*, stream.hip) feeds a long input in windows. A window is encoded as if it were
// the whole input -- n = the bytes on hand -- and the ring parse makes that sound: a position before t_last = ring_t_last(n)
// is matched by match_long with a head and a tail that depend on the position alone (enc_common.h), and t_last only grows
// with n, so the walk up to the first position >= t_last is the walk of ANY longer input with this prefix. The window is
// cut behind the last bvx2 block that ends between two events, the later of which leaves the walk before t_last and
// leaves it in a state it can be resumed from; blocks [0, n_blocks) are final and the next window starts from that state.
// One lane per stream.
__global__ __launch_bounds__(64) void enc_cut_kernel(const EncStream *__restrict__ streams, uint32_t n_streams,
                                                     const EncStreamOut *__restrict__ outs, const EncBlock *__restrict__ blocks,
                                                     const RangeRec *__restrict__ ranges, const SpecEvent *__restrict__ logs,
                                                     const MatchRec *__restrict__ gaps, const uint4 *__restrict__ gstate,
                                                     EncCut *__restrict__ cuts) {
    const uint32_t si = blockIdx.x * blockDim.x + threadIdx.x;
    if (si >= n_streams) return;
    const EncStream &es = streams[si];
    const EncStreamOut so = outs[si];
    EncCut c;
    c.found = 0; c.n_blocks = 0; c.index = 0; c.lit = 0; c.p_idx = 0; c.p_midx = 0; c.p_len = 0; c.skip = 0; c.out_len = 0;
    if (!so.status && ring_rounds(es.ring, es.n) && so.n_blocks > 1) {
        const uint32_t t_safe = ring_t_last(es.n);
        const EncBlock *bk = blocks + es.blk_base;
        const RangeRec *rg = ranges + es.range_base;
        // the walk's state behind event `ord`: x = index | GS_NO_RESUME, y z w = pending; lit = the literal index there
        auto state_after = [&](uint32_t ord, uint32_t &lit) -> uint4 {
            uint32_t lo = 0, hi = so.n_ranges;   // the range that holds the event (ranges are in order of out_off)
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (rg[mid].out_off <= ord) lo = mid; else hi = mid;
            }
            const RangeRec r = rg[lo];
            const uint64_t at = r.begin + (ord - r.out_off);
            if (r.kind == 0) {
                const SpecEvent ev = logs[at];
                const uint32_t pl = ev_plen(ev);
                lit = ev_lit_after(ev);
                return make_uint4(ev_index_after(ev), pl ? ev_pidx(ev) : 0u, pl ? ev_pmidx(ev) : 0u, pl);
            }
            const MatchRec m = (gaps + es.match_base)[at];
            lit = m.lit_pos + m.l + m.m;
            return gstate[es.match_base + at];
        };
        for (uint32_t b = so.n_blocks - 1; b-- > 0 && !c.found;) {
            const uint32_t e = bk[b].cut_ev, skip = bk[b].cut_skip;
            if (e == NONE || e >= so.n_matches) continue;
            // the event the cut lies in must be one the walk made before t_safe (it is then an event of every longer input) ...
            if (skip) {
                uint32_t unused;
                const uint4 se = state_after(e, unused);
                if ((se.x & ~GS_NO_RESUME) >= t_safe) continue;
            }
            // ... and the walk is taken up in front of it
            uint32_t index, lit, p_idx, p_midx, p_len;
            if (e == 0) {
                index = es.start ? es.st_index : 0u; lit = es.start ? es.st_lit : 0u;
                p_idx = es.start ? es.st_pidx : 0u; p_midx = es.start ? es.st_pmidx : 0u; p_len = es.start ? es.st_plen : 0u;
            } else {
                const uint4 sp = state_after(e - 1, lit);
                if ((sp.x & GS_NO_RESUME) || sp.x >= t_safe) continue;
                index = sp.x; p_idx = sp.y; p_midx = sp.z; p_len = sp.w;
            }
            c.found = 1; c.n_blocks = b + 1;
            c.index = index; c.lit = lit; c.p_idx = p_idx; c.p_midx = p_midx; c.p_len = p_len;
            c.skip = (e == 0 ? es.st_skip : 0u) + skip;
        }
        if (c.found)
            for (uint32_t b = 0; b < c.n_blocks; b++) c.out_len += (uint64_t)bk[b].hdr_len + bk[b].lit_len + bk[b].lmd_len;
    }
    cuts[si] = c;
}

// ------------------------------------------------------------------------------------ launchers

void launch_enc_spec(const uint8_t *src, const EncStream *streams, const uint2 *segs, uint32_t n_segs, uint32_t seg, const uint32_t *prev,
                     const uint32_t *rec, const uint64_t *bitmap, SpecEvent *logs, SpecHeader *hdrs, bool repo, hipStream_t st) {
    if (!n_segs) return;
    const dim3 grid((n_segs + 63) / 64);
    if (repo) {
        if (n_segs > 98304) hipLaunchKernelGGL((enc_spec_kernel<true, true>), grid, dim3(64), 0, st, src, streams, segs, n_segs, seg, prev, rec, bitmap, logs, hdrs);
        else hipLaunchKernelGGL((enc_spec_kernel<false, true>), grid, dim3(64), 0, st, src, streams, segs, n_segs, seg, prev, rec, bitmap, logs, hdrs);
    } else if (n_segs > 98304) hipLaunchKernelGGL((enc_spec_kernel<true, false>), grid, dim3(64), 0, st, src, streams, segs, n_segs, seg, prev, rec, bitmap, logs, hdrs);
    else hipLaunchKernelGGL((enc_spec_kernel<false, false>), grid, dim3(64), 0, st, src, streams, segs, n_segs, seg, prev, rec, bitmap, logs, hdrs);
}
void launch_enc_stitch(const uint8_t *src, const EncStream *streams, uint32_t ns, const uint2 *segs, uint32_t n_segs, uint32_t seg, const uint32_t *prev,
                       const uint32_t *rec, const uint64_t *bitmap, const SpecEvent *logs, const SpecHeader *hdrs, uint4 *sync,
                       RangeRec *ranges, MatchRec *gaps, uint4 *gstate, EncStreamOut *outs, hipStream_t st) {
    hipLaunchKernelGGL(enc_sync_kernel, dim3((n_segs + 63) / 64), dim3(64), 0, st, streams, segs, n_segs, seg, logs, hdrs, sync);
    if (seg_stitch_waves(ns) > 1)
        hipLaunchKernelGGL(enc_stitch_kernel<STITCH_WAVES>, dim3(ns), dim3(64 * STITCH_WAVES), 0, st, src, streams, ns, seg, prev, rec, bitmap, logs,
                           hdrs, sync, ranges, gaps, gstate, outs);
    else
        hipLaunchKernelGGL(enc_stitch_kernel<1>, dim3(ns), dim3(64), 0, st, src, streams, ns, seg, prev, rec, bitmap, logs, hdrs, sync, ranges, gaps,
                           gstate, outs);
}
void launch_enc_cut(const EncStream *streams, uint32_t ns, const EncStreamOut *outs, const EncBlock *blocks, const RangeRec *ranges,
                    const SpecEvent *logs, const MatchRec *gaps, const uint4 *gstate, EncCut *cuts, hipStream_t st) {
    hipLaunchKernelGGL(enc_cut_kernel, dim3((ns + 63) / 64), dim3(64), 0, st, streams, ns, outs, blocks, ranges, logs, gaps, gstate, cuts);
}
void launch_enc_compact(const EncStream *streams, const uint32_t *slot_stream, uint32_t n_slots, uint32_t ns, const EncStreamOut *outs,
                        const RangeRec *ranges, const SpecEvent *logs, const MatchRec *gaps, MatchRec *matches, uint32_t *pc,
                        uint32_t *pl, uint2 *rsum, hipStream_t st) {
    if (!n_slots) return;
    hipLaunchKernelGGL(enc_compact_kernel, dim3(n_slots), dim3(64), 0, st, streams, slot_stream, outs, ranges, logs, gaps, matches, pc, pl,
                       rsum);
    hipLaunchKernelGGL(enc_rscan_kernel, dim3(ns), dim3(RSCAN_THREADS), 0, st, streams, ns, outs, rsum);
    hipLaunchKernelGGL(enc_papply_kernel, dim3(n_slots), dim3(64), 0, st, streams, slot_stream, outs, ranges, rsum, pc, pl);
}
void launch_enc_segment(const EncStream *streams, uint32_t ns, const uint32_t *slot_stream, uint32_t n_slots, bool try_parallel,
                        const MatchRec *matches, const uint32_t *pc, const uint32_t *pl, uint2 *lmds, EncBlock *blocks, EncStreamOut *outs,
                        uint32_t *flags, hipStream_t st) {
    (void)hipMemsetAsync(flags, 0, (size_t)ns * 4, st);
    if (try_parallel && n_slots) {
        hipLaunchKernelGGL(enc_segpar_kernel, dim3(n_slots), dim3(64), 0, st, streams, slot_stream, n_slots, outs, matches, pc, pl, lmds, blocks,
                           flags);
        hipLaunchKernelGGL(enc_segfin_kernel, dim3(ns), dim3(64), 0, st, streams, ns, pc, blocks, outs, flags);
    }
    hipLaunchKernelGGL(enc_segment_kernel, dim3(ns), dim3(SEGM_THREADS), 0, st, streams, ns, matches,
                       pc, pl, lmds, blocks, outs, flags);
}
void launch_enc_lmd(const EncStream *streams, const uint32_t *slot_stream, uint32_t n_slots, const EncStreamOut *outs,
                    const EncBlock *blocks, const MatchRec *matches, const uint32_t *pc, uint2 *lmds, hipStream_t st) {
    if (!n_slots) return;
    hipLaunchKernelGGL(enc_lmd_kernel, dim3(n_slots), dim3(256), 0, st, streams, slot_stream, outs, blocks, matches, pc, lmds);
}

}  // namespace lzmi
