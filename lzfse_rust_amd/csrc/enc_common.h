// Shared definitions of the encode kernels (encode.hip, encode_parse.hip).
#pragma once
#include "internal.h"

namespace lzmi {

constexpr uint32_t TILE_POS = 65472;        // positions per candidate tile (enc_cand_kernel's unit of XCD placement): a multiple of 64 and of 256
// A CHAIN tile (enc_chain_kernel, enc_link_kernel) is 1, 2 or 4 of them: offset + 1 stays below 2^18, and an entry of the
// last-seen table is that | 14 check bits. A longer tile has fewer first occurrences to link across tiles (enc_link: 11 % of
// the Snappy positions at 64 Ki, 3 % at 256 Ki) and fewer summaries and lists to write; a call of few large streams keeps the
// short tiles, which are what it has to fill the chip with (chain_tile_mult).
constexpr uint32_t CH_TILE_MAX_MULT = 4;
// (at least ~1 024 tiles where the call has them: two workgroups of eight waves per CU, two rounds; `count(m)` = chain tiles of the call at m)
template <class F> __host__ inline uint32_t chain_tile_mult(F count, int forced) {
#ifdef LZMI_CH_MULT
    forced = LZMI_CH_MULT;   // (experiment switch, experiments/README.md)
#endif
    if (forced == 1 || forced == 2 || forced == 4) return (uint32_t)forced;
    for (uint32_t m = CH_TILE_MAX_MULT; m > 1; m >>= 1)
        if (count(m) >= 1024) return m;
    return 1;
}
#ifndef LZMI_SEG
#define LZMI_SEG 2048
#endif
constexpr uint32_t SEG = LZMI_SEG;          // positions per speculative-parse segment of a large batch (small ones: 512, seg_for)
#ifndef LZMI_OVER
#define LZMI_OVER 512
#endif
constexpr uint32_t OVER = LZMI_OVER;              // overrun of a segment walker into the next segment
constexpr uint32_t SEG_EV_CAP = (SEG + OVER) / 4 + 4;  // every emit advances literal_index by >= 4
__host__ __device__ __forceinline__ uint32_t seg_ev_cap(uint32_t seg) { return (seg + OVER) / 4 + 4; }   // events a walker of `seg` positions can log
// Segment length of a batch (one value for all its streams). A walker's time goes with its segment, and a small batch is one
// round of resident walkers whose longest walk sets enc_spec's time: quarter segments, four times the walkers (html x 16:
// enc_spec 0.30 -> 0.13 ms, encode 2.55 -> 3.2 GB/s). Not for large streams: the stitcher is one wave per stream and pays per
// boundary (one 64 MiB stream with 1 024-position segments: enc_spec 0.64 -> 0.49 ms but enc_stitch 0.33 -> 0.73; round 4: a lone
// 4 MiB stream -- a small window of the stream encoder -- enc_spec 0.48 -> 0.20, enc_stitch 0.03 -> 0.10 ms, the call 1.18 -> 0.98 ms,
// hence 8 MiB and not 2); not for large batches: they are bound by the traffic of their records and events, to which more
// walkers (each overrunning by OVER) only add (188 MB in 768 streams: no difference; 32 x 4 MiB: encode 27.2 -> 25.6 GB/s).
// Round 4: a call of FEW streams gets the stitcher with several waves per stream (enc_stitch_kernel<STITCH_WAVES>, which takes
// 64 boundaries per wave and step), and then quarter segments pay for large streams too: one 64 MiB stream, enc_spec + enc_stitch
// 0.62 + 0.37 -> see DESIGN.md section 6 (up to 512 MiB of positions: the walkers' logs are 8 instead of 5 bytes per position).
#ifndef LZMI_STITCH_WAVES
#define LZMI_STITCH_WAVES 8
#endif
constexpr int STITCH_WAVES = LZMI_STITCH_WAVES;
#ifndef LZMI_STITCH_FEW
#define LZMI_STITCH_FEW 8
#endif
__host__ __forceinline__ int seg_stitch_waves(uint32_t n_streams) { return n_streams <= LZMI_STITCH_FEW ? STITCH_WAVES : 1; }
__host__ __forceinline__ uint32_t seg_for(uint64_t positions, uint64_t longest, uint32_t n_streams) {
#ifndef LZMI_SEG_SMALL_BATCH
#define LZMI_SEG_SMALL_BATCH 24
#define LZMI_SEG_SMALL_STREAM 8
#endif
#ifndef LZMI_SEG_FEW_MAX
#define LZMI_SEG_FEW_MAX 32
#endif
    if (seg_stitch_waves(n_streams) > 1 && positions <= ((uint64_t)LZMI_SEG_FEW_MAX << 20)) return 512u;
    return (positions <= ((uint64_t)LZMI_SEG_SMALL_BATCH << 20) && longest <= ((uint64_t)LZMI_SEG_SMALL_STREAM << 20)) ? 512u : SEG;
}
constexpr uint32_t NONE = 0xFFFFFFFFu;      // no previous position
constexpr uint32_t NONE_TILE = 0xFFFFFFFEu; // no previous position inside the tile (link pending)
constexpr uint32_t FCAP = 1023;             // cap of the forward length computed per position (>= GOOD_MATCH_LEN); 10 bits of a record
constexpr uint32_t XCAP = 16384;            // cap of the exact per-lane extension in the segment walkers (runs of 2 400 .. 7 000 equal bytes, 64 x 4 MiB:
                                            // with 4 096 every segment gave up and the stitcher walked alone, enc_stitch 8.1 ms; now 0.2; 65 536 costs zeros and the Snappy mix)
constexpr uint32_t BCAP = 13;               // cap of the backward length kept per position (4 bits of a record): what ONE dword-aligned 16-byte load in front of a candidate always holds

// Candidate record of a position (4 bytes): the best match as if the position were visited. distance (18 bits) |
// forward length << 18 (10 bits; FCAP = "at least FCAP": the walkers ask for the exact length) | backward length << 28
// (4 bits; BCAP = "at least BCAP"). 0 = no match. Half the bytes of round 1's {word, length} pair: the segment walkers
// are bound by the rate of L2 misses on this array, and twice as many positions share a sector.
__device__ __forceinline__ uint32_t rec_make(uint32_t dist, uint32_t fwd, uint32_t bwd) { return dist | (fwd << 18) | (bwd << 28); }
__device__ __forceinline__ uint32_t rec_dist(uint32_t r) { return r & 0x3FFFFu; }
__device__ __forceinline__ uint32_t rec_fwd(uint32_t r) { return (r >> 18) & 0x3FFu; }
__device__ __forceinline__ uint32_t rec_bwd(uint32_t r) { return r >> 28; }

__device__ __forceinline__ int e_lane() { return threadIdx.x & 63; }
__device__ __forceinline__ uint32_t e_readlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ uint32_t bucket_of(uint32_t v) { return (v * 0x9E3779B1u) >> (32 - HASH_BITS); }

// Link record of a position (4 bytes): distance to the previous position of its bucket (18 bits: 0 = none within the
// match window, fse/constants.rs:42) | 14 check bits of THAT position's 4 bytes (a second hash: entries of one bucket
// whose bytes differ agree in them with probability 2^-14; the byte compare, which starts at byte 0, settles it). The
// record that names a candidate also says whether it can match, so a chain of 4 candidates costs 3 dependent gathers.
__device__ __forceinline__ uint32_t chk_of(uint32_t v) { return (v * 0x85EBCA6Bu) >> 18; }
__device__ __forceinline__ uint32_t link_make(uint32_t dist, uint32_t chk_prev) { return dist | (chk_prev << 18); }
__device__ __forceinline__ uint32_t link_dist(uint32_t r) { return r & 0x3FFFFu; }
__device__ __forceinline__ uint32_t link_chk(uint32_t r) { return r >> 18; }
// Entry of a chain tile's last-seen table and of its summary: (offset in tile + 1, 18 bits) | check bits << 18; 0 = none
__device__ __forceinline__ uint32_t seen_make(uint32_t off1, uint32_t v) { return off1 | (chk_of(v) << 18); }

// ---- geometry of the reference's ring encoder in flat positions (encode/frontend_ring.rs, encode/constants.rs:23-33) ----
// The ring (512 KiB, filled in 16 KiB blocks) is matched in rounds: round 0 takes positions [0, RING/2 + BLK) with the ring
// holding [0, RING); every later round starts at the position idx the one before stopped at, with
// head = align_down(idx, BLK) - RING/2 (reposition_head :250-254), tail = head + RING, and runs to head + RING/2 + BLK
// = align_down(idx, BLK) + BLK (match_long :359-397) -- as long as the input reaches tail (match_block :216-219). What is
// left is matched by match_short (:401-450) with tail = n. A position that is VISITED at all therefore sees a head and a
// tail that depend on the position and on n only: rounds end at multiples of BLK, and a round that a long match overshot
// only moves the head further up, where it cannot bind (the match ends at literal_idx, and a backward extension never
// passes literal_idx). T_last = first position of the short phase's block.
constexpr uint32_t RING_SIZE = 0x80000u, RING_BLK = 0x4000u, RING_HALF = RING_SIZE / 2;
constexpr uint32_t RING_FIRST_END = RING_HALF + RING_BLK;                    // end of round 0
constexpr uint32_t RING_LONG_MATCH = RING_HALF - RING_BLK - 44u;             // LONG_MATCH_LEN (:110), OVERMATCH_SLACK = 4 + 40 (:21)
// A WINDOW of a longer stream (lzfse_mi_estream_*: the stream encoder feeds the device in windows) carries ring = 3: its
// positions are relative to a base that is a multiple of RING_BLK and lies beyond the first ring, so the two rules about
// the beginning of a stream (round 0 is longer; the ring has not wrapped yet) do not apply, and every other rule is
// invariant under such a shift. n is then the end of the data the window holds.
constexpr uint32_t RING_CONT = 2u;
struct RingGeo {
    uint32_t head, tail;
    bool is_short;     // matched by match_short: forward length limited by the end of the input, not by LONG_MATCH_LEN
    bool wrapped;      // whatever lies one ring before a position past the tail is input (not the zeros of a fresh ring)
};
__host__ __device__ __forceinline__ uint32_t ring_t_last(uint32_t n) { return ((n - RING_HALF) & ~(RING_BLK - 1)) + RING_BLK; }  // n >= RING_SIZE
__host__ __device__ __forceinline__ uint32_t ring_head_at(uint32_t p_aligned) { return p_aligned < RING_HALF ? 0u : p_aligned - RING_HALF; }
__host__ __device__ __forceinline__ RingGeo ring_geo(uint32_t ring, uint32_t n, uint32_t p) {
    RingGeo g;
    const bool cont = (ring & RING_CONT) != 0;
    g.wrapped = cont;
    if (!cont && n < RING_SIZE) { g.head = 0; g.tail = n; g.is_short = true; return g; }
    const uint32_t tl = ring_t_last(n);
    if (p < tl) {
        g.head = (!cont && p < RING_FIRST_END) ? 0u : ring_head_at(p & ~(RING_BLK - 1));
        g.tail = g.head + RING_SIZE; g.is_short = false;
    } else { g.head = ring_head_at(tl); g.tail = n; g.is_short = true; }
    return g;
}
// lowest position a backward extension of a match found at p may reach (find_match :482: match_idx - head); 0 for the slice parse
// (midx: the candidate. rel0: a block of a slice beyond the first -- the reference's self.src begins there; a speculative walker that
// stands in the part of the stream in front of it, where nothing it finds is ever used, must still stay inside the stream)
__host__ __device__ __forceinline__ uint32_t parse_head(uint32_t ring, uint32_t n, uint32_t p, uint32_t rel0 = 0, uint32_t midx = 0xFFFFFFFFu) {
    if (!ring) return rel0 <= midx ? rel0 : 0u;
    if (!(ring & RING_CONT) && (n < RING_SIZE || p < RING_FIRST_END)) return 0u;
    const uint32_t tl = ring_t_last(n);
    return ring_head_at((p < tl ? p : tl) & ~(RING_BLK - 1));
}
// the rounds of match_long exist (frontend_ring.rs:216-219)
__host__ __device__ __forceinline__ bool ring_rounds(uint32_t ring, uint32_t n) { return ring && ((ring & RING_CONT) || n >= RING_SIZE); }
// end of the first round at or after position B0 = a multiple of RING_BLK
__host__ __device__ __forceinline__ uint32_t ring_round_floor(uint32_t ring, uint32_t B) { return (!(ring & RING_CONT) && B < RING_FIRST_END) ? RING_FIRST_END : B; }

struct EncStream {       // one input stream (n > VN_CUTOFF) of the batch
    uint64_t src_off;    // offset of the stream in d_src
    uint64_t pos_base;   // offset of the stream in the per-position arrays (prev, rec)
    uint64_t dst_off, dst_cap;
    uint64_t lmd_base;   // offset into the LMD array
    uint64_t stage_base; // offset into the block staging area
    uint64_t stage_cap;
    uint32_t n;          // stream length
    uint32_t tile_base;  // index of the stream's first tile
    uint32_t blk_base;   // index of the stream's first block slot
    uint32_t blk_cap;    // number of block slots
    uint32_t lmd_cap;
    uint32_t user_index; // index in the caller's arrays
    // speculative parse
    uint32_t seg_base, n_seg;      // segments of this stream in the segment arrays
    uint32_t range_base, range_cap;
    uint64_t match_base;           // offset into the match / gap / prefix-sum arrays
    uint32_t match_cap;
    uint32_t ring;                 // 1: the ring / stream encoder's parse (encode/frontend_ring.rs), 0: the slice parse; 3: a window of it (RING_CONT)
    // a window that continues a stream: the parse starts from this state instead of (0, 0, nothing pending)
    uint32_t start, st_index, st_lit, st_pidx, st_pmidx, st_plen;
    uint32_t st_skip;   // bytes of the window's first event (its literals, then its match) that the window before has emitted already
    // ---- one block of the reference's slice front end beyond the first (frontend_bytes.rs:160-211 match_any, :348-375 reposition): a
    // slice of more than BLOCK_GUIDE + 3 bytes is matched in blocks of BLOCK_GUIDE bytes whose positions up to the block's limit are
    // visited; the device takes a block per call (a "repo window", encode.hip enc_slice_blocks). The stream of such a call begins
    // `rel0` bytes BEFORE the block (the raw bytes of the bvx2 block the front end had not closed yet: its events are carried) ----
    uint32_t stop;      // the walk ends here (the block's limit) instead of at n - 3; 0: n - 3
    uint32_t no_flush;  // != 0: no flush_pending / flush_literals behind the walk (the input goes on: the end state leaves in EncStreamOut)
    uint32_t n_carry;   // events of the block the front end had not closed when the block before ended: the first n_carry gap events of the stream, put there by the host
    uint32_t rel0;      // position of the block's first byte in this stream (self.src of the reference): no backward extension goes below it
    uint32_t st_raw;    // first raw byte of the first bvx2 block this call makes (start != 0)
    uint32_t pad2;
};
__device__ __forceinline__ uint32_t walk_end(const EncStream &es) { return es.stop ? es.stop : es.n - 3; }

// One emitted match of a segment walker + the walker state after it, in 16 bytes (round 4; eight plain words before: the
// events were 2.4 of the 43 bytes of HBM traffic per input byte written by the walkers and read again by the sync search and the
// compaction). A walker starts at its segment's first position S with literal_index = S and visits positions below
// S + seg + OVER < S + 4096, so everything it logs lies within 12 bits of something else in the event:
//   x  e_idx
//   y  e_dist (18) | e_idx - literal_index before (12) | p_len bits 1:0
//   z  pending: p_idx - p_midx (18) | index_after - p_idx (12) | p_len bits 3:2      (0 when nothing is pending)
//   w  e_len (15: <= XCAP + a backward extension inside the walk) | index_after - (e_idx + e_len) (12) | p_len bits 5:4
// (a pending match is shorter than GOOD_MATCH_LEN = 40, match_object.rs:12-33, and after an emit one is pending only when the
// pending one before it was emitted and the incoming match took its place: index_after = p + 1, p_idx = p - backward length)
typedef uint4 SpecEvent;
static_assert(SEG + OVER < 4096 && XCAP + SEG + OVER < 32768, "SpecEvent's fields");
__device__ __forceinline__ SpecEvent ev_pack(uint32_t e_idx, uint32_t e_len, uint32_t e_dist, uint32_t e_lit, uint32_t index_after,
                                             uint32_t p_idx, uint32_t p_midx, uint32_t p_len) {
    SpecEvent e;
    e.x = e_idx;
    e.y = e_dist | ((e_idx - e_lit) << 18) | (p_len << 30);
    e.z = p_len ? ((p_idx - p_midx) | ((index_after - p_idx) << 18) | ((p_len >> 2) << 30)) : 0u;
    e.w = e_len | ((index_after - (e_idx + e_len)) << 15) | ((p_len >> 4) << 27);
    return e;
}
__device__ __forceinline__ uint32_t ev_idx(const SpecEvent &e) { return e.x; }
__device__ __forceinline__ uint32_t ev_len(const SpecEvent &e) { return e.w & 0x7FFFu; }
__device__ __forceinline__ uint32_t ev_dist(const SpecEvent &e) { return e.y & 0x3FFFFu; }
__device__ __forceinline__ uint32_t ev_lit(const SpecEvent &e) { return e.x - ((e.y >> 18) & 0xFFFu); }        // literal_index before the match
__device__ __forceinline__ uint32_t ev_lit_after(const SpecEvent &e) { return e.x + (e.w & 0x7FFFu); }
__device__ __forceinline__ uint32_t ev_index_after(const SpecEvent &e) { return e.x + (e.w & 0x7FFFu) + ((e.w >> 15) & 0xFFFu); }
__device__ __forceinline__ uint32_t ev_plen(const SpecEvent &e) { return (e.y >> 30) | ((e.z >> 30) << 2) | (((e.w >> 27) & 3u) << 4); }
__device__ __forceinline__ uint32_t ev_pidx(const SpecEvent &e) { return ev_index_after(e) - ((e.z >> 18) & 0xFFFu); }
__device__ __forceinline__ uint32_t ev_pmidx(const SpecEvent &e) { return ev_pidx(e) - (e.z & 0x3FFFFu); }
struct SpecHeader {
    uint32_t n_events, status;                     // status 1: aborted on a record needing an exact length
    uint32_t f_index, f_lit, f_pidx, f_pmidx, f_plen, pad;  // final state
};
struct MatchRec {    // ordered match list of a stream: literal run [lit_pos, lit_pos + l), then m bytes at distance d
    uint32_t lit_pos, l, m, d;
};
struct RangeRec {    // a run of events adopted into the match list
    uint64_t begin;  // kind 0: absolute SpecEvent index, kind 1: index into the stream's gap events
    uint32_t count, out_off;
    uint32_t kind, pad;
};

struct EncTile {         // one chain tile; carries what its kernels need of the stream, so that a workgroup starts with ONE descriptor load
    uint32_t stream;
    uint32_t n;          // stream length
    uint32_t start;      // first position of the tile (stream relative)
    uint32_t ring;       // EncStream::ring
    uint64_t src_off;    // offset of the stream in d_src
    uint64_t pos_base;   // offset of the stream in the per-position arrays (prev, rec)
    uint32_t skip_lo, skip_hi;   // positions [skip_lo, skip_hi) are NOT entered into the history (chain tiles only): the reference never pushed
                                 // them -- those between the position a block's last match was found at and the block's limit, when that match
                                 // ran past the limit (frontend_bytes.rs:180,191-193,357)
};

struct EncBlock {        // one bvx2 block (written by the walk kernel)
    uint64_t lmd_start;  // index into the LMD array
    uint64_t stage_off;  // staging offset of this block's bytes
    uint32_t src_start;  // first raw byte of the block (stream relative)
    uint32_t n_lmd, n_lit, n_match;
    // filled by the block kernel
    uint32_t hdr_len, lit_len, lmd_len;
    uint32_t cut_ev;     // the event the NEXT block begins with ...
    // events that lie completely inside the block (enc_segment_kernel -> enc_lmd_kernel)
    uint32_t ev_begin, ev_end, head_lmds, head_prev_d;
    uint32_t cut_skip, pad;   // ... and how many of its bytes (literals first, then match bytes) this and earlier blocks hold already
};

// Where a window of a longer stream may be cut (enc_cut_kernel): behind the last block that ends between two events whose
// parse no later data can change; the state of the parse there.
struct EncCut {
    uint32_t found, n_blocks;            // blocks [0, n_blocks) are final
    uint32_t index, lit, p_idx, p_midx, p_len;   // the walk's state in front of the event the next block begins with
    uint32_t skip;                       // bytes of that event the final blocks hold already
    uint64_t out_len;                    // their bytes
};

struct EncStreamOut {
    uint32_t n_blocks;
    int32_t status;
    uint64_t out_len;
    uint32_t n_matches, n_ranges;
    // the walk's state where it ended (no_flush streams): literal index, pending match, and 1 + the position the last match was
    // found at when that match carried the literal index past the walk's end (0 otherwise)
    uint32_t e_lit, e_pidx, e_pmidx, e_plen, e_cross, e_pad;
    // walk statistics (diagnostics only)
    uint32_t iters, emits, capped, refills;
    uint64_t cycles;
};

// staging layout of one block: [header + weights | literal payload | lmd payload]
__host__ __device__ inline uint32_t stage_lit_off() { return 704; }
__host__ __device__ inline uint32_t stage_lmd_off(uint32_t n_lit) { return 704 + ((((n_lit + 3) / 4 * 4) * 10 + 7) / 8 + 24 + 15) / 16 * 16; }
__host__ __device__ inline uint32_t stage_need(uint32_t n_lit, uint32_t n_lmd) {
    return stage_lmd_off(n_lit) + ((8 + (n_lmd * 54 + 7) / 8 + 24 + 15) / 16 * 16);
}


// encode_match.hip
void launch_enc_chain(const uint8_t *src, const EncTile *tiles, uint32_t n_tiles, uint32_t tile_pos, uint32_t *prev, uint32_t *summary,
                      uint32_t *flist, uint32_t *fcount, uint32_t *redo, bool force_redo, hipStream_t st);
void launch_enc_link(const EncTile *tiles, uint32_t n_tiles, uint32_t tile_pos, uint32_t *prev, const uint32_t *summary,
                     const uint32_t *flist, const uint32_t *fcount, hipStream_t st);
void launch_enc_cand(const uint8_t *src, const EncStream *streams, const EncTile *tiles, uint32_t n_tiles, const uint32_t *prev, uint32_t *rec,
                     uint64_t *bitmap, hipStream_t st);

// encode_parse.hip
void launch_enc_spec(const uint8_t *src, const EncStream *streams, const uint2 *segs, uint32_t n_segs, uint32_t seg, const uint32_t *prev,
                     const uint32_t *rec, const uint64_t *bitmap, SpecEvent *logs, SpecHeader *hdrs, bool repo, hipStream_t st);
void launch_enc_stitch(const uint8_t *src, const EncStream *streams, uint32_t ns, const uint2 *segs, uint32_t n_segs, uint32_t seg, const uint32_t *prev,
                       const uint32_t *rec, const uint64_t *bitmap, const SpecEvent *logs, const SpecHeader *hdrs, uint4 *sync,
                       RangeRec *ranges, MatchRec *gaps, uint4 *gstate /* may be null */, EncStreamOut *outs, hipStream_t st);
void launch_enc_cut(const EncStream *streams, uint32_t ns, const EncStreamOut *outs, const EncBlock *blocks, const RangeRec *ranges,
                    const SpecEvent *logs, const MatchRec *gaps, const uint4 *gstate, EncCut *cuts, hipStream_t st);
void launch_enc_compact(const EncStream *streams, const uint32_t *slot_stream, uint32_t n_slots, uint32_t ns, const EncStreamOut *outs,
                        const RangeRec *ranges, const SpecEvent *logs, const MatchRec *gaps, MatchRec *matches, uint32_t *pc,
                        uint32_t *pl, uint2 *rsum, hipStream_t st);
void launch_enc_segment(const EncStream *streams, uint32_t ns, const uint32_t *slot_stream, uint32_t n_slots, bool try_parallel,
                        const MatchRec *matches, const uint32_t *pc, const uint32_t *pl, uint2 *lmds, EncBlock *blocks, EncStreamOut *outs,
                        uint32_t *flags, hipStream_t st);
void launch_enc_lmd(const EncStream *streams, const uint32_t *slot_stream, uint32_t n_slots, const EncStreamOut *outs,
                    const EncBlock *blocks, const MatchRec *matches, const uint32_t *pc, uint2 *lmds, hipStream_t st);

}  // namespace lzmi
