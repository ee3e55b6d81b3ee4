// Decode side of the MI355X LZFSE codec: hand-written HIP kernels for gfx950 (wave64).
//
//   dec_walk_kernel   header walk per stream            decoder.rs:61-99, probe.rs:11-35
//   dec_scan / rank   the same walk for large streams:  magic scan, candidates sorted, chain from position 0
//   dec_fse_kernel    weights -> tables (LDS) -> FSE     fse_core.rs:49-141, weights.rs:83-105,
//                     decode of literals and LMDs        decoder.rs:244-335, literals.rs:49-91
//   dec_lz_kernel     literal / match copy (LZ77)        lz/writer.rs:97-186, lz/object.rs:27-74
//                     + bvx- and bvxn blocks             raw/block.rs:46-93, vn/vn_core.rs:40-287
//   dec_ck / dec_lzp  the same LZ stage with several workgroups per stream (few streams of some size): tickets of one LMD
//                     group, the part of a tile that needs no earlier output done ahead of the ticket's turn
//   dec_jump_*        the LZ stage of few large streams by pointer jumping over per-byte origins
//
// Parallelism: FSE is serial per bit stream, so the entropy stage runs one workgroup (two
// waves: LMD stream, literal stream) per bvx2 block with its 7 KiB of tables in LDS and
// uses lanes 0..2 / 0..3 of a wave for the interleaved L,M,D / four literal states
// (one LDS table fetch + a DPP prefix sum over bit counts per step). The LZ stage runs one
// workgroup per stream (up to four when the streams are few), stages a tile of output in LDS,
// resolves near matches there and writes the tile back with coalesced 16-byte stores.
#include <algorithm>

#include <type_traits>

#include "internal.h"

namespace lzmi {

// ------------------------------------------------------------------------------------ utils

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// value of `v` in the lane `delta` below (0 for lanes < delta); DPP row_shr within rows of
// 16 lanes is enough for the 3- and 4-lane prefix sums used here.
template <int DELTA>
__device__ __forceinline__ uint32_t dpp_shr(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + DELTA, 0xF, 0xF, true);
}

__device__ __forceinline__ uint32_t read_lane(uint32_t v, int lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}

// wave-level inclusive scan (DPP, common.h)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) { return wave_incl_sum(v); }

// ------------------------------------------------------------------------------------ headers

// ------------------------------------------------------------------------------------ walk

// One thread per stream. EMIT = false: count blocks / LMDs / literals, validate headers, and leave the descriptors
// (with stream-relative LMD / literal bases) in the walk cache. EMIT = true: the serial re-walk for streams with
// more blocks than their share of the cache: writes BlockDesc records at the bases the host assigned from the counts.
// The cached descriptors are placed by dec_emit_kernel, in parallel.
template <bool EMIT>
__global__ void dec_walk_kernel(const uint8_t *__restrict__ src, const StreamIn *__restrict__ streams,
                                uint32_t n_streams, StreamWalk *__restrict__ walk,
                                const StreamPlan *__restrict__ plan, BlockDesc *__restrict__ blocks,
                                const uint32_t *__restrict__ settled) {
    __shared__ BlockDesc stage[EMIT ? 1 : 64][EMIT ? 1 : 8];
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    if (!EMIT && settled && settled[s]) return;   // walked by dec_rank_kernel
    if (EMIT && (plan[s].skip || plan[s].n_blocks <= streams[s].cache_cap)) return;
    const StreamIn in = streams[s];
    const uint8_t *base = src + in.src_off;
    uint64_t n = in.src_len, pos = 0;
    StreamWalk w;
    w.n_lmds = 0; w.n_lits = 0; w.raw_total = 0; w.n_blocks = 0; w.status = 0; w.err_block = 0; w.n_vxn = 0; w.detail = 0; w.pad = 0;
    uint64_t blk_i = 0, lmd_i = 0, lit_i = 0;
    if (EMIT) { blk_i = plan[s].blk_base; lmd_i = plan[s].lmd_base; lit_i = plan[s].lit_base; }
    uint32_t max_blocks = EMIT ? plan[s].n_blocks : 0xFFFFFFFFu;
    for (;;) {
        if (n - pos < 4) { w.status = LZFSE_MI_PAYLOAD_UNDERFLOW; break; }
        uint64_t avail = n - pos;
        // a whole v2 header in one round trip (this loop is a chain of dependent loads, one block after the other)
        const bool wide = avail >= V2_HEADER_SIZE;
        uint64_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;
        if (wide) { q0 = ld_u64(base + pos); q1 = ld_u64(base + pos + 8); q2 = ld_u64(base + pos + 16); q3 = ld_u64(base + pos + 24); }
        uint32_t magic = wide ? (uint32_t)q0 : ld_u32(base + pos);
        if (magic == MAGIC_EOS) {
            if (avail != 4) w.status = LZFSE_MI_PAYLOAD_OVERFLOW;  // decoder.rs:93-95
            break;
        }
        BlockDesc d;
        d.src_pos = in.src_off + pos; d.src_end = in.src_off + n; d.dst_rel = w.raw_total;
        d.lmd_base = lmd_i; d.lit_base = lit_i; d.stream = s;
        d.n_lmd = 0; d.n_lit = 0; d.n_raw = 0; d.payload = 0;
        uint64_t skip = 0;
        int st = 0;
        // A block whose header is sound but whose payload runs past the end of the stream is still emitted (and the walk
        // ends there): the reference takes the payload piece by piece (decoder.rs:102-157), so an error inside the part
        // that is present (weights, literals, an LZVN opcode) comes before the PayloadUnderflow of the missing part.
        bool truncated = false;
        if (magic == MAGIC_VX2 || magic == MAGIC_VX1) {
            FseHeader h;
            bool v1 = magic == MAGIC_VX1;
            if (avail < (v1 ? V1_HEADER_SIZE : V2_HEADER_SIZE)) st = LZFSE_MI_PAYLOAD_UNDERFLOW;
            else st = v1 ? fse_load_v1(base + pos, h) : fse_parse_v2(q0, q1, q2, q3, h);
            if (st == LZFSE_MI_FSE_BAD_LMD_COUNT) w.detail = h.lmd_num;          // FseErrorKind::BadLmdCount(num)
            if (st == LZFSE_MI_FSE_BAD_LITERAL_COUNT) w.detail = h.lit_num;      // FseErrorKind::BadLiteralCount(num)
            if (!st) {
                skip = (uint64_t)h.hdr_size + h.lit_payload + h.lmd_payload;
                truncated = skip > avail;  // decode/take.rs:10-19
                d.kind = v1 ? KIND_VX1 : KIND_VX2;
                d.n_lmd = h.lmd_num; d.n_lit = h.lit_num; d.n_raw = h.n_raw;
            }
        } else if (magic == MAGIC_VXN) {
            if (avail < 12) st = LZFSE_MI_PAYLOAD_UNDERFLOW;
            else {
                d.kind = KIND_VXN;
                w.n_vxn++;
                d.n_raw = ld_u32(base + pos + 4);
                d.payload = ld_u32(base + pos + 8);
                skip = 12ull + d.payload;
                truncated = skip > avail;
            }
        } else if (magic == MAGIC_RAW) {
            if (avail < 8) st = LZFSE_MI_PAYLOAD_UNDERFLOW;
            else {
                d.kind = KIND_RAW;
                d.n_raw = ld_u32(base + pos + 4);
                skip = 8ull + d.n_raw;
                if (skip > avail) st = LZFSE_MI_PAYLOAD_UNDERFLOW;  // raw/block.rs:70-92
            }
        } else {
            st = LZFSE_MI_BAD_BLOCK;
            w.detail = magic;  // Error::BadBlock(magic)
        }
        if (st) { w.status = st; w.err_block = w.n_blocks; break; }
        if (EMIT) {
            if (w.n_blocks >= max_blocks) break;  // cannot happen: same walk as the count pass
            blocks[blk_i + w.n_blocks] = d;
        } else if (w.n_blocks < in.cache_cap) {
            // `blocks` is the walk cache in this pass. The descriptors are staged in LDS, 8 per thread, and written
            // out together: a store per block would make the next header load wait for it (loads and stores
            // complete in order), doubling the time of this serial walk.
            stage[threadIdx.x][w.n_blocks & 7] = d;
            if ((w.n_blocks & 7) == 7)
                for (uint32_t k = 0; k < 8; k++) blocks[in.cache_off + w.n_blocks - 7 + k] = stage[threadIdx.x][k];
        }
        w.n_blocks++;
        w.n_lmds += d.n_lmd; lmd_i += d.n_lmd;
        w.n_lits += d.n_lit; lit_i += d.n_lit;
        // (bvxn: at most 136 output bytes per payload byte -- of what is there, when the block is cut; lzfse_mi_decode_size)
        w.raw_total += d.kind == KIND_VXN ? min((uint64_t)d.n_raw, 136ull * (truncated ? avail : (uint64_t)d.payload)) : (uint64_t)d.n_raw;
        if (truncated) { w.status = LZFSE_MI_PAYLOAD_UNDERFLOW; w.err_block = w.n_blocks; break; }
        pos += skip;
    }
    if (!EMIT) {
        const uint64_t nc = w.n_blocks < in.cache_cap ? w.n_blocks : in.cache_cap;
        for (uint64_t k = nc & ~7ull; k < nc; k++) blocks[in.cache_off + k] = stage[threadIdx.x][k & 7];
        walk[s] = w;
    }
}

// ---- Parallel header walk of LARGE streams. The serial walk is a chain of dependent loads, one round trip per block
// (0.9 us: 0.77 ms for the 860 blocks of a 64 MiB text stream). Instead: every byte position of the stream is tested for
// the bvx2 magic and a header that validates (dec_scan_kernel; a few false candidates inside payloads do no harm), the
// candidates are sorted by position, every candidate finds the candidate that starts where it ends, and ONE thread
// follows that chain from position 0 through LDS (dec_rank_kernel). The result is taken only when the chain runs from
// position 0 over sound bvx2 blocks to the end-of-stream magic in the stream's last 4 bytes, i.e. when the serial walk
// would have found exactly these blocks and no error; anything else (another block kind, a damaged or cut block, more
// candidates or blocks than fit) leaves the stream to the serial walk, whose statuses are the reference's. ----
constexpr uint32_t FW_CAP = 4096;       // candidates per stream
constexpr uint32_t FW_SCAN_BYTES = 4096; // bytes per scan workgroup (256 threads x 16)

__global__ __launch_bounds__(256) void dec_scan_kernel(const uint8_t *__restrict__ src, const StreamIn *__restrict__ streams,
                                                       const uint32_t *__restrict__ elig, uint32_t *__restrict__ count,
                                                       uint2 *__restrict__ cand) {
    const uint32_t e = blockIdx.y, s = elig[e];
    const StreamIn in = streams[s];
    const uint64_t n = in.src_len;
    const uint64_t q0 = (uint64_t)blockIdx.x * FW_SCAN_BYTES + (uint64_t)threadIdx.x * 16;
    if (q0 >= n) return;
    const uint8_t *base = src + in.src_off;
    // 19 bytes cover the 16 magic positions of this thread
    uint64_t lo8, hi8, top4;
    if (q0 + 20 <= n) { lo8 = ld_u64(base + q0); hi8 = ld_u64(base + q0 + 8); top4 = ld_u32(base + q0 + 16); }
    else {
        lo8 = hi8 = top4 = 0;
        for (int k = 0; k < 19; k++) {
            const uint64_t v = q0 + k < n ? base[q0 + k] : 0;
            if (k < 8) lo8 |= v << (8 * k); else if (k < 16) hi8 |= v << (8 * (k - 8)); else top4 |= v << (8 * (k - 16));
        }
    }
    for (int k = 0; k < 16; k++) {
        // bytes k .. k + 3 of the 19
        const uint64_t a = k < 8 ? lo8 : hi8, b2 = k < 8 ? hi8 : top4;
        const int sh = 8 * (k & 7);
        const uint32_t magic = (uint32_t)(sh ? (a >> sh) | (b2 << (64 - sh)) : a);
        if (magic != MAGIC_VX2) continue;
        const uint64_t q = q0 + k;
        if (q + V2_HEADER_SIZE > n) continue;
        FseHeader h;
        if (fse_parse_v2(ld_u64(base + q), ld_u64(base + q + 8), ld_u64(base + q + 16), ld_u64(base + q + 24), h)) continue;
        const uint64_t total = (uint64_t)h.hdr_size + h.lit_payload + h.lmd_payload;
        if (q + total + 4 > n) continue;   // a block of the chain is followed by at least the end-of-stream magic
        const uint32_t slot = atomicAdd(&count[e], 1u);
        if (slot < FW_CAP) cand[(uint64_t)e * FW_CAP + slot] = make_uint2((uint32_t)q, (uint32_t)total);
    }
}

__global__ __launch_bounds__(1024) void dec_rank_kernel(const uint8_t *__restrict__ src, const StreamIn *__restrict__ streams,
                                                        const uint32_t *__restrict__ elig, const uint32_t *__restrict__ count,
                                                        const uint2 *__restrict__ cand, StreamWalk *__restrict__ walk,
                                                        BlockDesc *__restrict__ cache, uint32_t *__restrict__ settled) {
    __shared__ uint64_t key[FW_CAP];      // position << 32 | length, sorted; later: n_lmd << 32 | n_lit of the chain's blocks
    __shared__ uint32_t aux[FW_CAP];      // next candidate of each candidate; later: n_raw of the chain's blocks
    __shared__ uint16_t order[FW_CAP];    // candidates in chain order
    __shared__ uint32_t sh_n;
    __shared__ uint64_t sh_scan[3][16];
    const uint32_t e = blockIdx.x, s = elig[e];
    const int tid = threadIdx.x;
    const StreamIn in = streams[s];
    const uint32_t n = count[e];
    if (n == 0 || n > FW_CAP || in.src_len > 0xFFFFFFF0ull) return;   // settled[s] stays 0: serial walk
    const uint8_t *base = src + in.src_off;
    uint32_t np2 = 64;                    // sort size: the power of two that holds the candidates
    while (np2 < n) np2 <<= 1;
    for (uint32_t i = tid; i < np2; i += 1024) {
        const uint2 c = i < n ? cand[(uint64_t)e * FW_CAP + i] : make_uint2(0xFFFFFFFFu, 0u);
        key[i] = ((uint64_t)c.x << 32) | c.y;
    }
    __syncthreads();
    // bitonic sort (positions are distinct)
    for (uint32_t k = 2; k <= np2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < np2; i += 1024) {
                const uint32_t ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t a = key[i], b = key[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { key[i] = b; key[ixj] = a; }
                }
            }
            __syncthreads();
        }
    // next candidate: the one that starts where this one ends; 0xFFFF: the end-of-stream magic in the last 4 bytes; 0xFFFE: none
    for (uint32_t i = tid; i < n; i += 1024) {
        const uint64_t kv = key[i];
        const uint64_t target = (kv >> 32) + (uint32_t)kv;
        uint32_t nx = 0xFFFEu;
        if (target + 4 == in.src_len) { if (ld_u32(base + target) == MAGIC_EOS) nx = 0xFFFFu; }
        else {
            uint32_t lo = 0, hi = n;   // first candidate with position >= target
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((key[mid] >> 32) < target) lo = mid + 1; else hi = mid; }
            if (lo < n && (key[lo] >> 32) == target) nx = lo;
        }
        aux[i] = nx;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t r = 0, i = 0;
        bool ok = (key[0] >> 32) == 0;
        while (ok) {
            if (r >= n) { ok = false; break; }
            order[r++] = (uint16_t)i;
            const uint32_t nx = aux[i];
            if (nx == 0xFFFFu) break;
            if (nx == 0xFFFEu) { ok = false; break; }
            i = nx;
        }
        sh_n = (ok && r <= in.cache_cap) ? r : 0u;
    }
    __syncthreads();
    const uint32_t nb = sh_n;
    if (nb == 0) return;
    // the chain's blocks: counts from their headers, exclusive prefix sums in block order, descriptors into the walk cache
    uint32_t my_pos[FW_CAP / 1024];
    for (uint32_t q = 0; q < FW_CAP / 1024; q++) {
        const uint32_t r = tid * (FW_CAP / 1024) + q;
        my_pos[q] = r < nb ? (uint32_t)(key[order[r]] >> 32) : 0u;
    }
    __syncthreads();
    uint64_t run_lmd = 0, run_lit = 0, run_raw = 0;
    uint32_t v_lmd[FW_CAP / 1024], v_lit[FW_CAP / 1024], v_raw[FW_CAP / 1024];
    for (uint32_t q = 0; q < FW_CAP / 1024; q++) {
        const uint32_t r = tid * (FW_CAP / 1024) + q;
        v_lmd[q] = 0; v_lit[q] = 0; v_raw[q] = 0;
        if (r < nb) {
            FseHeader h;
            const uint8_t *hp = base + my_pos[q];
            (void)fse_parse_v2(ld_u64(hp), ld_u64(hp + 8), ld_u64(hp + 16), ld_u64(hp + 24), h);   // validated by the scan
            v_lmd[q] = h.lmd_num; v_lit[q] = h.lit_num; v_raw[q] = h.n_raw;
        }
        run_lmd += v_lmd[q]; run_lit += v_lit[q]; run_raw += v_raw[q];
    }
    // block-wide exclusive scan of the per-thread sums (1024 threads: 16 waves)
    uint64_t il = run_lmd, it = run_lit, ir = run_raw;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
        const uint64_t a = __shfl_up(il, dd), b2 = __shfl_up(it, dd), c2 = __shfl_up(ir, dd);
        if (lane >= dd) { il += a; it += b2; ir += c2; }
    }
    if (lane == 63) { sh_scan[0][wave] = il; sh_scan[1][wave] = it; sh_scan[2][wave] = ir; }
    __syncthreads();
    uint64_t ol = 0, ot = 0, orr = 0, tl = 0, tt = 0, tr = 0;
    for (int wv = 0; wv < 16; wv++) {
        const uint64_t a = sh_scan[0][wv], b2 = sh_scan[1][wv], c2 = sh_scan[2][wv];
        if (wv < wave) { ol += a; ot += b2; orr += c2; }
        tl += a; tt += b2; tr += c2;
    }
    uint64_t ex_lmd = ol + il - run_lmd, ex_lit = ot + it - run_lit, ex_raw = orr + ir - run_raw;
    for (uint32_t q = 0; q < FW_CAP / 1024; q++) {
        const uint32_t r = tid * (FW_CAP / 1024) + q;
        if (r < nb) {
            BlockDesc d;
            d.src_pos = in.src_off + my_pos[q]; d.src_end = in.src_off + in.src_len; d.dst_rel = ex_raw;
            d.lmd_base = ex_lmd; d.lit_base = ex_lit; d.stream = s; d.kind = KIND_VX2;
            d.n_lmd = v_lmd[q]; d.n_lit = v_lit[q]; d.n_raw = v_raw[q]; d.payload = 0;
            cache[in.cache_off + r] = d;
        }
        ex_lmd += v_lmd[q]; ex_lit += v_lit[q]; ex_raw += v_raw[q];
    }
    if (tid == 0) {
        StreamWalk w;
        w.n_lmds = tl; w.n_lits = tt; w.raw_total = tr; w.n_blocks = nb; w.status = 0; w.err_block = 0; w.n_vxn = 0; w.detail = 0; w.pad = 0;
        walk[s] = w;
        settled[s] = 1;
    }
}

// one thread per cached descriptor: add the stream's bases and put it at its place
__global__ void dec_emit_kernel(const StreamIn *__restrict__ streams, uint32_t n_streams, const StreamPlan *__restrict__ plan,
                                const BlockDesc *__restrict__ cache, uint64_t cache_total, BlockDesc *__restrict__ blocks) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= cache_total) return;
    uint32_t lo = 0, hi = n_streams - 1;  // last stream with cache_off <= e
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (streams[mid].cache_off <= e) lo = mid; else hi = mid - 1;
    }
    const StreamPlan pl = plan[lo];
    const uint64_t k = e - streams[lo].cache_off;
    if (pl.skip || k >= pl.n_blocks || pl.n_blocks > streams[lo].cache_cap) return;
    BlockDesc d = cache[e];
    d.lmd_base += pl.lmd_base; d.lit_base += pl.lit_base;
    blocks[pl.blk_base + k] = d;
}

// ------------------------------------------------------------------------------------ FSE stage

// Backward bit reader (bits/bit_reader.rs:11-72) for one wave, restated on bit positions: the
// stream is one little-endian integer read from its top; after c bits have been pulled the next
// pull(n) returns bits [rem - n, rem) with rem = 8 len - off - c. The reference's 64-bit accumulator
// and byte-wise flush are an implementation of exactly this, and its final under-run test
// (accum_bits + 8 idx < 64) is rem < 64 (the 8 pad bytes must stay untouched).
//
// All positions are in bits relative to the 4-byte aligned address A <= base. The payload is
// streamed through a linear 128-dword LDS buffer holding dwords [cb, cb + 128) of A-space; when the
// cursor gets near the bottom the lower half moves up and a half prefetched 64 steps earlier drops in.
struct BitWindow {
    int32_t rem;              // bits remaining, A-space (uniform)
    int32_t base_bit;         // 8 * (base - A)
    int32_t cb;               // A-space dword index of buf[0] (uniform, may be negative)
    uint32_t *buf;            // 128 dwords in LDS
    const uint8_t *ga;        // A
    const uint8_t *glo, *ghi; // readable range of the source buffer
    uint32_t pend;            // prefetched dword (cb - 64 + lane)
};

__device__ __forceinline__ uint32_t bw_gload(const BitWindow &w, int32_t dw) {
    const uint8_t *a = w.ga + (int64_t)dw * 4;
    // (the payload is global memory: said explicitly, or the pointer kept in a struct beside an LDS pointer is treated as
    // generic and every refill becomes a flat load, which also waits on the LDS counter)
    typedef const __attribute__((address_space(1))) uint32_t *gptr;
    return (a >= w.glo && a + 4 <= w.ghi) ? *(gptr)(uintptr_t)a : 0u;
}

// the 64 bits below `rem`: bits [rem - 64, rem)
__device__ __forceinline__ uint64_t bw_window(const BitWindow &w) {
    const int32_t lo = w.rem - 64;
    const int32_t slot = (lo >> 5) - w.cb;      // 0 <= slot <= 125 by the rotation invariant
    const uint32_t sh = (uint32_t)lo & 31;
    const uint32_t d0 = w.buf[slot], d1 = w.buf[slot + 1], d2 = w.buf[slot + 2];
    const uint32_t x0 = __builtin_amdgcn_alignbit(d1, d0, sh), x1 = __builtin_amdgcn_alignbit(d2, d1, sh);
    return (uint64_t)x0 | ((uint64_t)x1 << 32);
}

// The window of one decode step. The reference refills its accumulator once per step (bit_reader.rs:39-50) from byte
// index idx = ceil((rem - 64) / 8) of the slice, and a read at a negative index returns ZERO for all 64 bits, not just
// for the bytes in front of the slice (bit_src.rs:35-46): from the moment fewer than 57 bits remain, a (damaged) stream
// decodes as zeros. Valid streams never get there (finalize demands rem >= 64), error codes of damaged ones depend on it.
__device__ __forceinline__ uint64_t bw_step_window(const BitWindow &w) {
    const uint64_t v = bw_window(w);
    return (w.rem - w.base_bit) < 57 ? 0ull : v;
}

// bit_reader.rs:20-30; len counts the 8 pad bytes in front of the payload
__device__ inline int bw_init(BitWindow &w, const uint8_t *base, uint32_t len, uint32_t off, const uint8_t *glo,
                              const uint8_t *ghi, uint32_t *buf) {
    const uintptr_t b = (uintptr_t)base;
    w.ga = (const uint8_t *)(b & ~(uintptr_t)3);
    w.base_bit = (int32_t)(b & 3) * 8;
    w.glo = glo; w.ghi = ghi; w.buf = buf;
    w.rem = w.base_bit + (int32_t)(8 * len) - (int32_t)off;
    const int32_t t0 = (w.rem - 64) >> 5;
    w.cb = (t0 - 32) & ~63;
    const int l = lane_id();
    buf[l] = bw_gload(w, w.cb + l);
    buf[64 + l] = bw_gload(w, w.cb + 64 + l);
    w.pend = bw_gload(w, w.cb - 64 + l);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // the unused top `off` bits of the last byte must be zero (Error::BadBitStream)
    if (off != 0 && (base[len - 1] >> (8 - off)) != 0) return LZFSE_MI_BAD_BIT_STREAM;
    return 0;
}

// consume `total` bits; keeps dwords [t, t + 2] of the next window inside the buffer
__device__ __forceinline__ void bw_advance(BitWindow &w, uint32_t total) {
    w.rem -= (int32_t)total;
    const int32_t t = (w.rem - 64) >> 5;
    if (t - w.cb < 16) {
        const int l = lane_id();
        const uint32_t low = w.buf[l];
        w.buf[64 + l] = low;
        w.buf[l] = w.pend;
        w.cb -= 64;
        w.pend = bw_gload(w, w.cb - 64 + l);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}
// the same in two parts for loops that run their steps with most lanes switched off: bw_consume inside (no lane of the
// wave is needed), bw_refill outside with all 64 lanes, often enough that `steps` steps of at most 64 bits stay inside
__device__ __forceinline__ void bw_consume(BitWindow &w, uint32_t total) { w.rem -= (int32_t)total; }
__device__ __forceinline__ void bw_refill(BitWindow &w, int32_t steps) {
    const int32_t t = (w.rem - 64) >> 5;
    if (t - w.cb < 2 * steps + 2) {
        const int l = lane_id();
        const uint32_t low = w.buf[l];
        w.buf[64 + l] = low;
        w.buf[l] = w.pend;
        w.cb -= 64;
        w.pend = bw_gload(w, w.cb - 64 + l);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// bit_reader.rs:64-71
__device__ __forceinline__ int bw_finalize(const BitWindow &w) {
    return (w.rem - w.base_bit < 64) ? LZFSE_MI_PAYLOAD_UNDERFLOW : 0;
}

constexpr int FSE_THREADS = 128;

// ---- order of the FSE workgroups: a block's time is the length of its longer bit stream (its LMD count, or a quarter of
// its literal count), and a stream ends with a short block. Blocks are handed out longest first (a counting sort into
// 64 classes), so that the last round of workgroups is made of short blocks instead of whatever comes last. ----
__device__ __forceinline__ uint32_t fse_order_key(const BlockDesc &d) {
    if (d.kind != KIND_VX2 && d.kind != KIND_VX1) return 63u;
    const uint32_t work = d.n_lmd > d.n_lit / 4 ? d.n_lmd : d.n_lit / 4;
    const uint32_t c = work / 160u;
    return 63u - (c < 63u ? c : 63u);
}
__global__ void dec_order_count_kernel(const BlockDesc *__restrict__ blocks, uint32_t n_blocks, uint32_t *__restrict__ hist) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n_blocks) atomicAdd(&hist[fse_order_key(blocks[b])], 1u);
}
__global__ __launch_bounds__(64) void dec_order_scan_kernel(uint32_t *__restrict__ hist) {
    const uint32_t v = hist[threadIdx.x];
    hist[threadIdx.x] = wave_incl_scan(v) - v;
}
__global__ void dec_order_place_kernel(const BlockDesc *__restrict__ blocks, uint32_t n_blocks, uint32_t *__restrict__ hist,
                                       uint32_t *__restrict__ order) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n_blocks) order[atomicAdd(&hist[fse_order_key(blocks[b])], 1u)] = b;
}

// the three steps in ONE launch for a call of few blocks (a small call is a chain of launches, each a few microseconds of an
// otherwise idle device: html x 16 decode 0.63 ms, 0.17 of it before the entropy kernel starts)
constexpr uint32_t ORDER_ONE_MAX = 8192;
__global__ __launch_bounds__(256) void dec_order_one_kernel(const BlockDesc *__restrict__ blocks, uint32_t n_blocks, uint32_t *__restrict__ order) {
    __shared__ uint32_t hist[64];
    if (threadIdx.x < 64) hist[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < n_blocks; b += 256) atomicAdd(&hist[fse_order_key(blocks[b])], 1u);
    __syncthreads();
    if (threadIdx.x < 64) { const uint32_t v = hist[threadIdx.x]; hist[threadIdx.x] = wave_incl_scan(v) - v; }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < n_blocks; b += 256) order[atomicAdd(&hist[fse_order_key(blocks[b])], 1u)] = b;
}

// Several small fills in one launch (the same reason): region k is n[k] dwords of value v[k]
struct FillSet {
    uint32_t *p[6];
    uint32_t n[6], v[6];
};
__global__ __launch_bounds__(256) void dec_fill_kernel(const FillSet f) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        if (i < f.n[k]) { f.p[k][i] = f.v[k]; return; }
        i -= f.n[k];
    }
}

// One LDS block, the pool FIRST: the payload rings then sit at LDS addresses 0 and 512, inside the offset fields of
// ds_read2_b32 / ds_read_b32, and a window fetch is one address register + two reads (no base add, no second move).
struct FseLds {
    // The buffers of the table set-up (weight payload, weights, cumulative weights) and those of the two bit streams share
    // one pool: the set-up is over (a workgroup barrier) before the first payload word is staged. 9.3 KB of LDS per block
    // instead of 11.4: the 32 wave slots of a CU, not its LDS, bound the number of resident blocks (16 instead of 14).
    uint32_t pool[168 + N_WEIGHTS];
    uint32_t u_tab[U_STATES];
    uint2 v_tab[L_STATES + M_STATES + D_STATES];
    int status[2];
    uint32_t sums[3];
    // dec_fse_kernel<true>: LMD records stored and visible / the LMD wave is through / records whose origins the literal wave
    // has written / it met a bad D (jump_consume_wave)
    uint32_t jready, jfin, jdone, jbad, jrun_out, jrun_lit;   // (... and the bytes / literals those records cover)
};

// Weights and decode tables of one bvx1 / bvx2 block into `lds` (weights.rs:66-105,189-200, decoder.rs:244-335), by the whole
// workgroup (FSE_THREADS threads, all of them call; workgroup barriers inside). lds.status[0] must be 0 on entry (written
// before a barrier the caller shares). Returns the block's weight status, the same value in every thread.
__device__ int fse_build_tables(FseLds &lds, uint32_t kind, const FseHeader &h, const uint8_t *p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *const pool = lds.pool;
    uint32_t *const u_tab = lds.u_tab;
    uint2 *const v_tab = lds.v_tab;
    int *const sh_status = lds.status;
    uint32_t *const stage = pool;                                               // up to 662 header+weight bytes (v2), dword aligned
    uint16_t *const wts = reinterpret_cast<uint16_t *>(pool + 168);            // N_WEIGHTS
    uint16_t *const cum = reinterpret_cast<uint16_t *>(pool + 168 + N_WEIGHTS / 2);
    // ---- weights (weights.rs:66-105) ----
    if (kind == KIND_VX2) {
        uint32_t nw = h.n_weight;
        for (uint32_t i = tid; i < 168; i += FSE_THREADS) {
            uint32_t v = 0;
            for (int k = 0; k < 4; k++) {
                uint32_t bi = i * 4 + k;
                if (bi < nw) v |= (uint32_t)p[V2_HEADER_SIZE + bi] << (8 * k);
            }
            stage[i] = v;
        }
        __syncthreads();
        if (tid == 0) {
            // weight_encoder.rs:10-20; zero-padded source makes the refill cadence irrelevant:
            // total bits T decides under/overflow exactly as accum_bits/i do in the reference.
            uint64_t accum = 0;
            int32_t bits = 0;
            uint32_t wi = 0, total_bits = 0;
            for (uint32_t k = 0; k < N_WEIGHTS; k++) {
                if (bits <= 32) {
                    uint32_t v = wi < 168 ? stage[wi] : 0u;
                    wi++;
                    accum |= (uint64_t)v << bits;
                    bits += 32;
                }
                uint32_t a = (uint32_t)accum;
                uint32_t cto = (~a) ? __builtin_ctz(~a) : 32u;  // trailing ones
                uint32_t nb, wv;
                if (cto == 0) { nb = 2; wv = (a >> 1) & 1; }
                else if (cto == 1) { nb = 3; wv = 2 + ((a >> 2) & 1); }
                else if (cto == 2) { nb = 5; wv = 4 + ((a >> 3) & 3); }
                else if (cto == 3) { nb = 8; wv = 8 + ((a >> 4) & 0xF); }
                else { nb = 14; wv = 24 + ((a >> 4) & 0x3FF); }
                wts[k] = (uint16_t)wv;
                accum >>= nb;
                bits -= (int32_t)nb;
                total_bits += nb;
            }
            int e = 0;
            if (total_bits > 8 * nw) e = LZFSE_MI_FSE_WEIGHT_PAYLOAD_UNDERFLOW;
            else if (nw > (total_bits + 7) / 8) e = LZFSE_MI_FSE_WEIGHT_PAYLOAD_OVERFLOW;
            sh_status[0] = e;
        }
    } else {
        for (uint32_t i = tid; i < N_WEIGHTS; i += FSE_THREADS) wts[i] = ld_u16(p + V1_HEADER_SIZE + 2 * i);
    }
    __syncthreads();
    // ---- cumulative weights + totals check (weights.rs:189-200) ----
    if (wave == 0) {
        // lane handles 4 U symbols; lanes also cover L (20), M (20), D (64) tables
        uint32_t w0 = wts[104 + 4 * lane], w1 = wts[105 + 4 * lane], w2 = wts[106 + 4 * lane], w3 = wts[107 + 4 * lane];
        uint32_t s4 = w0 + w1 + w2 + w3;
        uint32_t inc = wave_incl_scan(s4);
        uint32_t ex = inc - s4;
        cum[104 + 4 * lane] = (uint16_t)ex;
        cum[105 + 4 * lane] = (uint16_t)(ex + w0);
        cum[106 + 4 * lane] = (uint16_t)(ex + w0 + w1);
        cum[107 + 4 * lane] = (uint16_t)(ex + w0 + w1 + w2);
        uint32_t tot_u = read_lane(inc, 63);
        uint32_t wd = wts[40 + lane];
        uint32_t incd = wave_incl_scan(wd);
        cum[40 + lane] = (uint16_t)(incd - wd);
        uint32_t tot_d = read_lane(incd, 63);
        uint32_t wl = lane < 20 ? wts[lane] : 0u, wm = lane < 20 ? wts[20 + lane] : 0u;
        uint32_t incl = wave_incl_scan(wl), incm = wave_incl_scan(wm);
        if (lane < 20) { cum[lane] = (uint16_t)(incl - wl); cum[20 + lane] = (uint16_t)(incm - wm); }
        uint32_t tot_l = read_lane(incl, 63), tot_m = read_lane(incm, 63);
        if (lane == 0 && sh_status[0] == 0 &&
            (tot_l > L_STATES || tot_m > M_STATES || tot_d > D_STATES || tot_u > U_STATES))
            sh_status[0] = LZFSE_MI_FSE_BAD_WEIGHT_PAYLOAD;
    }
    __syncthreads();
    if (sh_status[0]) return sh_status[0];   // (uniform: read behind the barrier)

    // ---- decode tables (decoder.rs:244-335): per state, binary search the owning symbol ----
    for (uint32_t t = tid; t < U_STATES; t += FSE_THREADS) {
        uint32_t lo = 0, hi = 255;  // last i with cum[i] <= t
        while (lo < hi) {
            uint32_t mid = (lo + hi + 1) >> 1;
            if (cum[104 + mid] <= t) lo = mid; else hi = mid - 1;
        }
        uint32_t w = wts[104 + lo], j = t - cum[104 + lo];
        uint32_t e;
        if (j < w) {
            uint32_t k = __builtin_clz(w) - 21;  // clz(w) - clz(1024)
            uint32_t x = (2048u >> k) - w;
            int32_t delta; uint32_t kk;
            if (j < x) { kk = k; delta = (int32_t)((w + j) << k) - 1024; }
            else { kk = k - 1; delta = (int32_t)((j - x) << (k - 1)); }
            e = (uint32_t)(delta & 0xFFFF) | (lo << 16) | (kk << 24);
        } else {
            e = t;  // latch: k = 0, symbol 0, delta = own index
        }
        u_tab[t] = e;
    }
    for (uint32_t t = tid; t < L_STATES + M_STATES + D_STATES; t += FSE_THREADS) {
        uint32_t which = t < 64 ? 0u : (t < 128 ? 1u : 2u);
        uint32_t tb = which == 0 ? 0u : (which == 1 ? 64u : 128u);
        uint32_t wb = which == 0 ? 0u : (which == 1 ? 20u : 40u);
        uint32_t nsym = which == 2 ? 64u : 20u;
        uint32_t nst = which == 2 ? 256u : 64u;
        uint32_t clzn = which == 2 ? 23u : 25u;
        uint32_t ts = t - tb;
        uint32_t lo = 0, hi = nsym - 1;
        while (lo < hi) {
            uint32_t mid = (lo + hi + 1) >> 1;
            if (cum[wb + mid] <= ts) lo = mid; else hi = mid - 1;
        }
        uint32_t w = wts[wb + lo], j = ts - cum[wb + lo];
        uint2 e;
        if (j < w) {
            uint32_t k = __builtin_clz(w) - clzn;
            uint32_t x = ((2 * nst) >> k) - w;
            int32_t delta; uint32_t kk;
            if (j < x) { kk = k; delta = (int32_t)((w + j) << k) - (int32_t)nst; }
            else { kk = k - 1; delta = (int32_t)((j - x) << (k - 1)); }
            uint32_t vb = which == 0 ? l_extra_bits(lo) : which == 1 ? m_extra_bits(lo) : d_extra_bits(lo);
            uint32_t vv = which == 0 ? l_base_value(lo) : which == 1 ? m_base_value(lo) : d_base_value(lo);
            // k | value bits << 8 | delta << 16 (0 <= delta < states <= 256: (w + j) << k >= states by the choice of k) | -(k + value bits) << 24
            // (a signed byte: the step's prefix sum over L, M, D runs on it and IS the window shift, no negation)
            e.x = kk | (vb << 8) | ((uint32_t)(delta & 0xFF) << 16) | (((0u - (kk + vb)) & 0xFFu) << 24);
            e.y = vv;
        } else {
            e.x = (ts << 16);
            e.y = 0;
        }
        v_tab[t] = e;
    }
    __syncthreads();

    return 0;
}

// LDS table entry formats
//   U: (delta & 0xFFFF) | symbol << 16 | k << 24                      (decoder.rs:222-238; the symbol in byte 2 is what a
//      d16_hi byte store takes without a shift, the delta in the low word what an SDWA add takes)
//   V: .x = k | v_bits << 8 | (delta & 0xFF) << 16 | (-(k + v_bits) & 0xFF) << 24, .y = v_base   (decoder.rs:205-220; delta is below
//      the number of states, <= 256, by the choice of k; the signed byte is what the step's prefix sum over L, M, D runs on: it IS
//      the window shift)
template <int NT>
__device__ __forceinline__ void jump_init_block(const uint32_t b, const BlockDesc &d, const StreamPlan &pl, const BlockResult &br,
                                                const uint8_t *__restrict__ src, const StreamIn *__restrict__ streams,
                                                const LmdRec *__restrict__ lmds, const uint8_t *__restrict__ lits, uint8_t *dst_all,
                                                uint32_t *__restrict__ origin, uint32_t *__restrict__ jerr, uint32_t *sh,
                                                uint32_t lmds_done, uint32_t done_out, uint32_t done_lit, bool pre_bad);
__device__ void jump_consume_wave(volatile uint32_t *jready, volatile uint32_t *jfin, uint32_t *jdone, const LmdRec *bl, const uint8_t *blit,
                                  uint32_t n_raw, uint32_t n_lit, uint32_t o0, uint8_t *dst, uint32_t *org, uint32_t jb);

// JUMP: every stream of the call takes the pointer-jumping LZ path, and a block's workgroup goes on with that path's first step
// for its block (jump_init_block below) as soon as its entropy stage is done -- blocks of other kinds (raw) included
template <bool JUMP>
__global__ __launch_bounds__(FSE_THREADS) void dec_fse_kernel(
    const uint8_t *__restrict__ src, uint64_t src_total, const BlockDesc *__restrict__ blocks,
    uint32_t n_blocks, uint8_t *__restrict__ lit_out, LmdRec *__restrict__ lmd_out,
    BlockResult *__restrict__ results, const uint32_t *__restrict__ order, const StreamIn *__restrict__ streams,
    const StreamPlan *__restrict__ plan, uint8_t *dst_all, uint32_t *__restrict__ origin, uint32_t *__restrict__ jerr) {
    __shared__ __attribute__((aligned(16))) FseLds lds;
    uint32_t *const pool = lds.pool;
    uint32_t *const u_tab = lds.u_tab;
    uint2 *const v_tab = lds.v_tab;
    int *const sh_status = lds.status;
    uint32_t *const sh_sums = lds.sums;
    uint32_t (*const ring)[128] = reinterpret_cast<uint32_t (*)[128]>(pool);   // [2][128] payload windows
    uint32_t *const stg_lmd = pool + 256;          // 64 steps of (L, M, D) values, + one dump slot for the idle lanes (193)
    uint8_t *const stg_lit = reinterpret_cast<uint8_t *>(pool + 256 + 194);    // 64 groups of four literals, + dump slot (260 B)
    static_assert(256 + 194 + 65 <= 168 + N_WEIGHTS, "dec_fse LDS pool");

    if (blockIdx.x >= n_blocks) return;
    const uint32_t b = order[blockIdx.x];   // longest blocks first (dec_order_*)
    const BlockDesc d = blocks[b];
    const bool is_fse = d.kind == KIND_VX2 || d.kind == KIND_VX1;
    if (!is_fse && !JUMP) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    BlockResult br;
    br.status = 0; br.sum_l = 0; br.sum_m = 0; br.ok_until = 0;
    if (is_fse) br = [&]() -> BlockResult {
    const uint8_t *p = src + d.src_pos;
    const uint8_t *glo = src, *ghi = src + ((src_total + 3) & ~3ull);

    FseHeader h;
    int st = d.kind == KIND_VX1 ? fse_load_v1(p, h) : fse_load_v2(p, h);  // validated by the walk
    // the last block of a stream may be cut short (the walk emits it all the same): the reference takes header + weights,
    // literal payload and LMD payload one after the other, each with its own PayloadUnderflow (decoder.rs:102-141)
    const uint64_t avail = d.src_end - d.src_pos;
    if (!st && avail < h.hdr_size) st = LZFSE_MI_PAYLOAD_UNDERFLOW;
    if (st) { BlockResult r; r.status = st; r.sum_l = 0; r.sum_m = 0; r.ok_until = 0; return r; }
    const bool lit_short = avail < (uint64_t)h.hdr_size + h.lit_payload;
    const bool lmd_short = avail < (uint64_t)h.hdr_size + h.lit_payload + h.lmd_payload;
    if (tid < 2) sh_status[tid] = 0;
    if (JUMP && tid == 2) { lds.jready = 0; lds.jfin = 0; lds.jdone = 0; lds.jbad = 0; lds.jrun_out = 0; lds.jrun_lit = 0; }

    // ---- weights and decode tables ----
    {
        const int ts = fse_build_tables(lds, d.kind, h, p);
        if (ts) { BlockResult r; r.status = ts; r.sum_l = 0; r.sum_m = 0; r.ok_until = 0; return r; }
    }

    // ---- the two bit streams ----
    const uint32_t lit_off = h.hdr_size - 8;  // fse_core.rs:30-33,59: 8 bytes lent as reader pad
    const uint32_t lmd_off = h.hdr_size + h.lit_payload;
    if (wave == 1) {
        // literals.rs:49-91: four interleaved states in lanes 0..3, shared cursor
        BitWindow w;
        int e = lit_short ? LZFSE_MI_PAYLOAD_UNDERFLOW : bw_init(w, p + lit_off, h.lit_payload + 8, h.lit_bits, glo, ghi, ring[1]);
        if (lit_short) { w.rem = 0; w.base_bit = 0; w.cb = 0; w.buf = ring[1]; w.ga = glo; w.glo = glo; w.ghi = glo; w.pend = 0; }
        const int q4 = lane & 3;
        uint32_t state = q4 == 0 ? h.lit_state[0] : q4 == 1 ? h.lit_state[1] : q4 == 2 ? h.lit_state[2] : h.lit_state[3];
        uint8_t *out = lit_out + d.lit_base;
        const uint32_t n_groups = e ? 0u : h.lit_num >> 2;
        // The symbols go to LDS, one byte per state lane (the idle lanes write a dump slot, so the store needs no
        // exec masking), and every 64 groups the wave stores 64 dwords coalesced: two instructions per step.
        const uint32_t s_home = lane < 4 ? (uint32_t)lane : 256u, s_inc = lane < 4 ? 4u : 0u;
        uint32_t sidx = s_home;
        // the lookup of step g + 1 needs only the state, so it is issued before the window of step g + 1 is fetched: the
        // two LDS round trips of a step overlap instead of following each other
        uint32_t ent = u_tab[state];
        // The steps run with the lanes of row 0 only (16 at a time between two refill checks, which need the whole wave):
        // every LDS and vector instruction of a step then makes one pass instead of one per 32 lanes, and the LDS pipe is
        // what a full chip of these waves runs out of.
        // One step is about 25 instructions and a wave issues one every four cycles: a full chip of these waves is bound by
        // instruction ISSUE (16 blocks x 2 waves per CU), a lone block by its own instruction count plus one LDS round trip
        // per step -- either way the time is the number of instructions in a step. So: a chunk of SUB steps is unrolled
        // (no loop counter, the staging stores at constant offsets), the cursor is ONE scalar (r2 = rem - 64 - 32 cb: its
        // dword index and its bit offset are what the ring is read with), and the rule for a stream that has run out of
        // bits (bw_step_window) is only evaluated in chunks that can get there (the tail of a sound stream, damaged ones).
        constexpr uint32_t SUB = 16;
        static_assert(64 % SUB == 0 && 2 * SUB + 2 <= 62, "a sub-chunk ends where a 64-step flush does, and its refill margin stays in the lower half of the ring");
        const uint32_t *const ringw = w.buf;
        for (uint32_t g0 = 0; g0 < n_groups; g0 += SUB) {
            bw_refill(w, (int32_t)SUB);
            const uint32_t g1 = n_groups - g0 < SUB ? n_groups : g0 + SUB;
            const int32_t r2_bias = 64 + 32 * w.cb;
            int32_t r2 = w.rem - r2_bias;   // uniform: bits below the window's lowest bit, counted from the ring's first dword
            const int32_t low57 = w.base_bit + 57 - r2_bias;   // bw_step_window's rule (rem - base_bit < 57: the window reads as zero) is r2 < low57
            // (software-pipelined like the LMD loop below: the window's three dwords are asked for by the step before, the next
            // entry's look-up goes first)
            uint32_t d0, d1, d2;
            { const int32_t di = r2 >> 5; d0 = ringw[di]; d1 = ringw[di + 1]; d2 = ringw[di + 2]; }
            auto lit_step = [&](auto safe_tag, uint32_t slot_off) {
                constexpr bool SAFE = decltype(safe_tag)::value;
                const uint32_t k = ent >> 24;
                uint32_t pre = k + dpp_shr<1>(k);
                pre += dpp_shr<2>(pre);  // inclusive prefix over lanes 0..3
                const uint32_t x0 = __builtin_amdgcn_alignbit(d1, d0, (uint32_t)r2), x1 = __builtin_amdgcn_alignbit(d2, d1, (uint32_t)r2);
                uint64_t win = (uint64_t)x0 | ((uint64_t)x1 << 32);
                if (SAFE && r2 < low57) win = 0;
                const uint32_t x = (uint32_t)(win >> ((0u - pre) & 63));   // pre == 0 only with k == 0
                const uint32_t bits = __builtin_amdgcn_ubfe(x, 0u, k);
                state = bits + (ent & 0xFFFFu);   // (0 <= delta, and delta + the k bits read < 1024 by the construction of the table: no mask)
                const uint32_t sym = ent;
                ent = u_tab[state];
                __builtin_amdgcn_sched_barrier(0);
                r2 = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)(r2 - (int32_t)read_lane(pre, 3)));   // (scalar: the address arithmetic below stays off the vector pipe)
                const int32_t di = r2 >> 5;
                d0 = ringw[di]; d1 = ringw[di + 1]; d2 = ringw[di + 2];
                __builtin_amdgcn_sched_barrier(0);
                stg_lit[sidx + slot_off] = (uint8_t)(sym >> 16);
                __builtin_amdgcn_sched_barrier(0);
            };
            if (lane < 4) {
                if (g1 - g0 == SUB && r2 - 64 * (int32_t)SUB >= low57) {
#pragma unroll
                    for (uint32_t j = 0; j < SUB; j++) lit_step(std::false_type{}, 4u * j);
                    sidx += s_inc * SUB;
                } else {
                    for (uint32_t g = g0; g < g1; g++) { lit_step(std::true_type{}, 0u); sidx += s_inc; }
                }
            }
            w.rem = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)r2) + r2_bias;   // (uniform again: the lanes that were off kept the old value)
            if ((g1 & 63) == 0) {
                ((uint32_t *)out)[(g1 - 64) + lane] = ((const uint32_t *)stg_lit)[lane];
                sidx = s_home;
            }
        }
        if (n_groups & 63) {
            if (lane < (int)(n_groups & 63)) ((uint32_t *)out)[(n_groups & ~63u) + lane] = ((const uint32_t *)stg_lit)[lane];
        }
        if (!e) e = bw_finalize(w);
        uint32_t s0 = read_lane(state, 0) | read_lane(state, 1) | read_lane(state, 2) | read_lane(state, 3);
        if (!e && s0 != 0) e = LZFSE_MI_FSE_BAD_LMD_PAYLOAD;  // literals.rs:78-88
        if (lane == 0) sh_status[1] = e;
        if (JUMP) {
            // The literal stream of a block is shorter than its LMD stream more often than not (text: 5 800 against 10 000 steps):
            // this wave spends what is left of the block's time on the origins of the LMD records the other wave has stored so
            // far, and follows it to the end -- the pointer-jumping path's first step then has only the literal BYTES left to do.
            const StreamPlan pl = plan[d.stream];
            // (only behind a literal stream that decoded cleanly: the bytes are copied from it)
            if (!pl.skip && pl.jump && !e)
                jump_consume_wave(&lds.jready, &lds.jfin, &lds.jdone, lmd_out + d.lmd_base, lit_out + d.lit_base, d.n_raw, h.lit_num,
                                  (uint32_t)d.dst_rel, dst_all + streams[d.stream].dst_off, origin + pl.jbase, (uint32_t)pl.jbase);
        }
    } else {
        // fse_core.rs:91-141 (entropy part): L, M, D in lanes 0, 1, 2
        BitWindow w;
        int e = lmd_short ? LZFSE_MI_PAYLOAD_UNDERFLOW : bw_init(w, p + lmd_off, h.lmd_payload, h.lmd_bits, glo, ghi, ring[0]);
        if (lmd_short) { w.rem = 0; w.base_bit = 0; w.cb = 0; w.buf = ring[0]; w.ga = glo; w.glo = glo; w.ghi = glo; w.pend = 0; }
        const int li = lane < 2 ? lane : 2;
        uint32_t state = li == 0 ? h.lmd_state[0] : li == 1 ? h.lmd_state[1] : h.lmd_state[2];
        const uint32_t tbase = li == 0 ? 0u : (li == 1 ? 64u : 128u);
        LmdRec *out = lmd_out + d.lmd_base;
        const uint32_t n = e ? 0u : h.lmd_num;
        // The three values of a step go to LDS (idle lanes write a dump slot: no exec masking); every 64 steps the
        // wave turns them into 64 LMD records: D = 0 takes the previous distance (lmd_type.rs:153-160) by a wave scan,
        // the sums of L and M accumulate per lane, and the records are stored coalesced.
        const uint32_t s_home = lane < 3 ? (uint32_t)lane : 192u, s_inc = lane < 3 ? 3u : 0u;
        uint32_t sidx = s_home;
        uint32_t cum_l = 0, acc_m = 0, carry_d = 0;   // cum_l: literals consumed so far (uniform)
        uint32_t lit_over = 0xFFFFFFFFu;              // first LMD whose literals run past LITERALS_PER_BLOCK (fse_core.rs:119-128)
        auto flush = [&](uint32_t base, uint32_t cnt) {
            const bool have = (uint32_t)lane < cnt;
            const uint32_t vl = have ? stg_lmd[3 * lane] : 0u, vm = have ? stg_lmd[3 * lane + 1] : 0u, vd = have ? stg_lmd[3 * lane + 2] : 0u;
            const uint64_t nz = __ballot(vd != 0);
            const uint64_t upto = nz & (lane == 63 ? ~0ull : ((2ull << lane) - 1));
            const uint32_t from = __shfl(vd, upto ? 63 - __builtin_clzll(upto) : lane);
            const uint32_t dv = upto ? from : carry_d;
            if (JUMP) {
                // (the records of the batches BEFORE this one were stored 64 steps ago: the fence finds them done; the other wave may take them)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) *(volatile uint32_t *)&lds.jready = base;
            }
            if (have) out[base + lane] = make_uint2(vl | (vm << 16), dv);
            const uint32_t il = wave_incl_scan(vl);
            const uint64_t ov = __ballot(have && cum_l + il > LITERALS_PER_BLOCK);
            if (ov && lit_over == 0xFFFFFFFFu) lit_over = base + (uint32_t)__builtin_ctzll(ov);
            cum_l += read_lane(il, 63);
            acc_m += vm;
            if (nz) carry_d = read_lane(vd, 63 - __builtin_clzll(nz));
        };
        uint2 ent = v_tab[tbase + state];   // (looked up one step ahead of its window, see the literal loop)
        constexpr uint32_t SUB = 16;   // (see the literal loop)
        const uint32_t *const ringw = w.buf;
        const uint2 *const vt = v_tab + tbase;
        for (uint32_t i0 = 0; i0 < n; i0 += SUB) {
            bw_refill(w, (int32_t)SUB);
            const uint32_t i1 = n - i0 < SUB ? n : i0 + SUB;
            const int32_t r2_bias = 64 + 32 * w.cb;
            int32_t r2 = w.rem - r2_bias;
            const int32_t low57 = w.base_bit + 57 - r2_bias;
            // The window of a step is put together at the step's beginning from three dwords that the step BEFORE has asked
            // for (d0 .. d2 at cursor r2): a lone wave issues one instruction every ~5 cycles and an LDS read takes ~50, so a
            // step is as long as its instructions plus whatever no other instruction covers. In program order: the prefix of
            // the bit counts, the window, the state -> the NEXT entry's look-up (the chain every step hangs on) -> the cursor
            // and the next window's reads -> the value and its staging store, which nothing waits for.
            uint32_t d0, d1, d2;
            { const int32_t di = r2 >> 5; d0 = ringw[di]; d1 = ringw[di + 1]; d2 = ringw[di + 2]; }
            auto lmd_step = [&](auto safe_tag, uint32_t slot_off) {
                constexpr bool SAFE = decltype(safe_tag)::value;
                // (v_bfe_u32 takes its width from bits 4:0 of the operand: the entry itself serves as k <= 10)
                const uint32_t k = ent.x, vb = __builtin_amdgcn_ubfe(ent.x, 8u, 8u);
                const uint32_t ntot = (uint32_t)((int32_t)ent.x >> 24);   // -(k + vb)
                uint32_t npre = ntot + dpp_shr<1>(ntot);
                npre += dpp_shr<2>(ntot);   // (both moves read `ntot`: no wait between them) minus the bits of lanes 0 .. this one
                const uint32_t x0 = __builtin_amdgcn_alignbit(d1, d0, (uint32_t)r2), x1 = __builtin_amdgcn_alignbit(d2, d1, (uint32_t)r2);
                uint64_t win = (uint64_t)x0 | ((uint64_t)x1 << 32);
                if (SAFE && r2 < low57) win = 0;
                // a lane's field (state bits above value bits) is at most 10 + 15 bits: one 64-bit shift, two bit-field extracts
                const uint32_t x = (uint32_t)(win >> (npre & 63));  // npre == 0 only when k = vb = 0 below
                const uint32_t sb = __builtin_amdgcn_ubfe(x, vb, k);
                // (no mask: delta + the k bits read stay below the number of states by the construction of the table, decoder.rs:244-335)
                state = sb + ((ent.x >> 16) & 0xFFu);
                asm volatile("" : "+v"(state));   // (one SDWA add; left alone the compiler scales both terms by 8 first: one instruction more)
                const uint32_t base = ent.y;
                ent = vt[state];
                __builtin_amdgcn_sched_barrier(0);
                r2 = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)(r2 + (int32_t)read_lane(npre, 2)));
                const int32_t di = r2 >> 5;
                d0 = ringw[di]; d1 = ringw[di + 1]; d2 = ringw[di + 2];
                __builtin_amdgcn_sched_barrier(0);
                stg_lmd[sidx + slot_off] = base + __builtin_amdgcn_ubfe(x, 0u, vb);
                __builtin_amdgcn_sched_barrier(0);
            };
            if (lane < 3) {
                if (i1 - i0 == SUB && r2 - 64 * (int32_t)SUB >= low57) {
#pragma unroll
                    for (uint32_t j = 0; j < SUB; j++) lmd_step(std::false_type{}, 3u * j);
                    sidx += s_inc * SUB;
                } else {
                    for (uint32_t i = i0; i < i1; i++) { lmd_step(std::true_type{}, 0u); sidx += s_inc; }
                }
            }
            w.rem = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)r2) + r2_bias;
            if ((i1 & 63) == 0) { flush(i1 - 64, 64); sidx = s_home; }
        }
        if (n & 63) flush(n & ~63u, n & 63);
        uint32_t sum_m = acc_m;
#pragma unroll
        for (int dd = 32; dd > 0; dd >>= 1) sum_m += __shfl_xor(sum_m, dd);
        const uint32_t sum_l = cum_l;
        // order of the reference (fse_core.rs:91-141): reader init, then per LMD the literal_index test (and the match
        // copy, whose BadDValue the LZ stage raises for the first ok_until LMDs), then reader.finalize, then the totals
        uint32_t ok_until = 0;
        if (!e) {
            if (lit_over != 0xFFFFFFFFu) { e = LZFSE_MI_FSE_BAD_LMD_PAYLOAD; ok_until = lit_over; }
            else {
                ok_until = n;
                e = bw_finalize(w);
                const uint32_t s0 = read_lane(state, 0) | read_lane(state, 1) | read_lane(state, 2);
                if (!e && !(sum_l <= h.lit_num && sum_l + sum_m == h.n_raw && s0 == 0)) e = LZFSE_MI_FSE_BAD_LMD_PAYLOAD;
            }
        }
        if (lane == 0) { sh_status[0] = e; sh_sums[0] = sum_l; sh_sums[1] = sum_m; sh_sums[2] = ok_until; }
        if (JUMP) {
            // (every path of this wave ends here: the other wave waits for jfin)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) { *(volatile uint32_t *)&lds.jready = n; *(volatile uint32_t *)&lds.jfin = 1u; }
        }
    }
    __syncthreads();
    BlockResult r;
    // literals are loaded before the LMD stream is touched (decoder.rs:127-141)
    r.status = sh_status[1] ? sh_status[1] : sh_status[0];
    r.sum_l = sh_sums[0]; r.sum_m = sh_sums[1];
    r.ok_until = sh_status[1] ? 0u : sh_sums[2];
    return r;
    }();
    if (is_fse && tid == 0) results[b] = br;
    if (JUMP) {
        // (the block's literal and LMD arrays were stored by this workgroup's two waves: drained and visible to both)
        __syncthreads();
        const bool took = is_fse && !br.status;
        jump_init_block<FSE_THREADS>(b, d, plan[d.stream], br, src, streams, lmd_out, lit_out, dst_all, origin, jerr, pool,
                                     took ? lds.jdone : 0u, took ? lds.jrun_out : 0u, took ? lds.jrun_lit : 0u, took && lds.jbad != 0);
    }
}

// ------------------------------------------------------------------------------------ LZ stage

// exactly n <= 24 bytes, given as three little-endian words, to byte-aligned LDS in at most five stores of 8 / 8 / 4 / 2 / 1 bytes
// (gfx950 runs LDS in unaligned access mode; a byte loop costs the wave its longest run in iterations)
__device__ __forceinline__ void lds_put24(uint8_t *q, uint32_t n, uint64_t w0, uint64_t w1, uint64_t w2) {
    uint64_t cur = w0;
    if (n >= 8) { __builtin_memcpy(q, &w0, 8); q += 8; cur = w1; }
    if (n >= 16) { __builtin_memcpy(q, &w1, 8); q += 8; cur = w2; }
    if (n == 24) { __builtin_memcpy(q, &w2, 8); return; }
    if (n & 4) { const uint32_t c4 = (uint32_t)cur; __builtin_memcpy(q, &c4, 4); q += 4; cur >>= 32; }
    if (n & 2) { const uint16_t c2 = (uint16_t)cur; __builtin_memcpy(q, &c2, 2); q += 2; cur >>= 16; }
    if (n & 1) *q = (uint8_t)cur;
}

template <int NT>
__device__ __forceinline__ void block_excl_scan2(uint32_t a, uint32_t b, uint32_t &ea, uint32_t &eb,
                                                 uint32_t &ta, uint32_t &tb, uint32_t *sh /* 2 * NT/64 + 2 */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = NT / 64;
    uint32_t ia = wave_incl_scan(a), ib = wave_incl_scan(b);
    if (lane == 63) { sh[wave] = ia; sh[NW + wave] = ib; }
    __syncthreads();
    uint32_t oa = 0, ob = 0, sa = 0, sb = 0;
    for (int w = 0; w < NW; w++) {
        uint32_t va = sh[w], vb = sh[NW + w];
        if (w < wave) { oa += va; ob += vb; }
        sa += va; sb += vb;
    }
    ea = oa + ia - a; eb = ob + ib - b; ta = sa; tb = sb;
    __syncthreads();
}

// What the reference's decode loop (fse_core.rs:103-131) meets first within the first `count` LMDs of a block, in LMD
// order: a bad D (Error::BadDValue: the match is copied inside the loop, so this comes before the block's end-of-stream
// status) or the end of the destination (BUFFER_OVERFLOW: literals first, then D, then the match bytes, lz/writer.rs:
// 144-180). Returns 0 when neither happens. Called by the whole workgroup with uniform arguments; pos0 = stream
// position of the block's first byte, cap = capacity of the stream's destination. sh: 2 * NT / 64 + 4 words.
template <int NT>
__device__ int lmds_first_fault(const LmdRec *bl, uint32_t count, uint64_t pos0, uint64_t cap, uint32_t *sh) {
    constexpr int NW = NT / 64;
    uint32_t *first = sh + 2 * NW + 2;  // [0]: first bad D, [1]: first overflow
    uint64_t run = 0;
    for (uint32_t g0 = 0; g0 < count; g0 += NT) {
        const uint32_t idx = g0 + threadIdx.x;
        const bool valid = idx < count;
        const LmdRec r = valid ? bl[idx] : make_uint2(0, 0);
        const uint32_t l = r.x & 0xFFFF, m = r.x >> 16, dd = r.y;
        uint32_t ex_l, ex_s, tot_l, tot_s;
        if (threadIdx.x < 2) first[threadIdx.x] = 0xFFFFFFFFu;
        block_excl_scan2<NT>(l, l + m, ex_l, ex_s, tot_l, tot_s, sh);
        const uint64_t pm = pos0 + run + ex_s + l;  // position of the match = end of the LMD's literals
        if (valid && m != 0 && (dd == 0 || (uint64_t)dd > pm) && pm <= cap) atomicMin(&first[0], idx);
        if (valid && pm + m > cap) atomicMin(&first[1], idx);
        __syncthreads();
        const uint32_t fb = first[0], fo = first[1];
        __syncthreads();
        if (fb != 0xFFFFFFFFu && fb <= fo) return LZFSE_MI_BAD_D_VALUE;
        if (fo != 0xFFFFFFFFu) return LZFSE_MI_BUFFER_OVERFLOW;
        run += tot_s;
    }
    return 0;
}

// LZVN block decode by one lane (vn/vn_core.rs:134-283); bvxn only occurs in tiny or foreign
// streams so latency, not throughput, is what matters here.
__device__ int vn_decode_serial(const uint8_t *p, uint64_t avail, uint32_t n_raw, uint32_t n_payload,
                                uint8_t *dst, uint64_t &out_pos, uint64_t cap) {
    uint64_t q = 12;
    uint32_t match_distance = 0;
    const uint64_t mark = out_pos;
    if (avail - q < 8) return LZFSE_MI_PAYLOAD_UNDERFLOW;
    for (;;) {
        uint64_t rem = avail - q;
        const uint8_t *s = p + q;
        uint32_t opu = ld_u32(s);
        uint32_t b = opu & 0xFF, hi = b >> 4, lo4 = b & 15, low3 = b & 7;
        uint32_t l = 0, m = 0, op_len = 0;
        int kind;  // 0 literal, 1 match(prev d), 2 lmd, 3 nop, 4 eos, 5 udef
        if (hi == 0xE) { kind = 0; if (lo4 == 0) { l = ((opu >> 8) & 0xFF) + 16; op_len = 2; } else { l = lo4; op_len = 1; } }
        else if (hi == 0xF) { kind = 1; if (lo4 == 0) { m = ((opu >> 8) & 0xFF) + 16; op_len = 2; } else { m = lo4; op_len = 1; } }
        else if (hi == 0x7 || hi == 0xD) kind = 5;
        else if (hi == 0xA || hi == 0xB) {
            kind = 2; op_len = 3;
            m = (((opu & 7) << 2) | ((opu >> 8) & 3)) + 3; l = (opu >> 3) & 3;
            match_distance = (opu >> 10) & 0x3FFF;
        } else if (low3 == 7) {
            kind = 2; op_len = 3; m = ((opu >> 3) & 7) + 3; l = (opu >> 6) & 3;
            match_distance = (opu >> 8) & 0xFFFF;
        } else if (low3 == 6) {
            if (b == 0x06) kind = 4;
            else if (b == 0x0E || b == 0x16) kind = 3;
            else if (b < 0x40) kind = 5;
            else { kind = 2; op_len = 1; m = ((opu >> 3) & 7) + 3; l = (opu >> 6) & 3; }
        } else {
            kind = 2; op_len = 2; m = ((opu >> 3) & 7) + 3; l = (opu >> 6) & 3;
            match_distance = ((opu & 7) << 8) | ((opu >> 8) & 0xFF);
        }
        if (kind == 5) return LZFSE_MI_VN_BAD_OPCODE;
        if (kind == 4) {
            if (ld_u64(s) != 0x06ull) return LZFSE_MI_VN_BAD_PAYLOAD;
            q += 8;
            uint64_t consumed = q - 12, produced = out_pos - mark;
            if (consumed > n_payload) return LZFSE_MI_PAYLOAD_UNDERFLOW;
            if (produced > n_raw) return LZFSE_MI_VN_BAD_PAYLOAD;
            if (consumed != n_payload) return LZFSE_MI_PAYLOAD_OVERFLOW;
            if (produced != n_raw) return LZFSE_MI_VN_BAD_PAYLOAD;
            return 0;
        }
        if (kind == 3) { if (rem - 1 < 8) return LZFSE_MI_PAYLOAD_UNDERFLOW; q += 1; continue; }
        if (kind == 1) { if (rem - op_len < 8) return LZFSE_MI_PAYLOAD_UNDERFLOW; }
        else { if (rem - op_len < (uint64_t)l + 8) return LZFSE_MI_PAYLOAD_UNDERFLOW; }
        if (out_pos + l + m > cap) return LZFSE_MI_BUFFER_OVERFLOW;
        for (uint32_t t = 0; t < l; t++) dst[out_pos + t] = s[op_len + t];
        out_pos += l;
        if (m) {
            if (match_distance == 0 || match_distance > out_pos) return LZFSE_MI_BAD_D_VALUE;
            for (uint32_t t = 0; t < m; t++) {
                dst[out_pos + t] = dst[out_pos + t - match_distance];
            }
            out_pos += m;
        }
        q += op_len + l;
    }
}

constexpr uint32_t SHORT_COPY = 24;
// LMDs per thread of the tile kernels (measured, dec_lz ms on Snappy x 256 / x 64 / 256 x 4 MiB: 1 per thread 2.47 / 0.94 / 5.94,
// 2 per thread 2.17 / 0.81 / 5.07; 3 per thread, or 512 threads with a 16 KiB tile: slower again -- the tile overflows more often)
constexpr int LZ_LPT = 2;
constexpr int JUMP_SWEEPS = 3;   // jumping sweeps over a thread's unresolved bytes per workgroup barrier  // copies up to this many bytes are done by the owning lane

// ---- runs and short periods inside a tile (matches that overlap themselves: d < m) ----
// Byte k >= d of such a match is a copy of byte k mod d of the match itself. Its origin entry says so and carries
// ORG_WRAP: the entry is never swept (it is one hop from a byte that is), other chains pass through it, and the gather takes
// the resolved origin of the byte it names. A tile of zeros is ALL such bytes: swept like the rest they cost ~190
// instructions each (a 32 KiB pass 134 k cycles), now the match's first d bytes carry the chain and the rest ~15.
constexpr uint32_t ORG_WRAP = 0x8000u, ORG_OFF = 0x7FFFu;

// the sweeps of the pointer jumping when the tile holds ORG_WRAP entries (all threads; see the plain loop at the call sites)
template <int NT, int TILE>
__device__ __forceinline__ void tile_jump_wrap(uint16_t *s_org, const uint32_t *s_dm, uint32_t dm, int tid) {
    static_assert(TILE <= 32768, "15-bit tile offsets");
    constexpr uint32_t NTS = 31 - __builtin_clz((unsigned)NT);
    uint32_t um = dm;
    for (;;) {
#pragma unroll 1
        for (int sweep = 0; sweep < JUMP_SWEEPS && um; sweep++)
            for (uint32_t m2 = um; m2; m2 &= m2 - 1) {
                const uint32_t k = (uint32_t)__builtin_ctz(m2), b = tid + (k << NTS);
                const uint32_t o = s_org[b];
                if (o & ORG_WRAP) { um &= ~(1u << k); continue; }   // (static: resolved by the gather)
                if (!((s_dm[o & (NT - 1)] >> (o >> NTS)) & 1u)) { um &= ~(1u << k); continue; }
                const uint32_t o1 = s_org[o] & ORG_OFF;
                if (!((s_dm[o1 & (NT - 1)] >> (o1 >> NTS)) & 1u)) { s_org[b] = (uint16_t)o1; um &= ~(1u << k); continue; }
                s_org[b] = (uint16_t)(s_org[o1] & ORG_OFF);
            }
        if (!__syncthreads_or(um != 0)) break;
    }
}

template <int NT, int TILE>
__device__ __forceinline__ void tile_gather_wrap(uint8_t *t, const uint16_t *s_org, uint32_t dm, int tid) {
    constexpr uint32_t NTS = 31 - __builtin_clz((unsigned)NT);
    for (uint32_t m2 = dm; m2; m2 &= m2 - 1) {
        const uint32_t b2 = tid + ((uint32_t)__builtin_ctz(m2) << NTS);
        uint32_t o = s_org[b2];
        if (o & ORG_WRAP) o = s_org[o & ORG_OFF];   // (the byte of the match's first d that this one repeats: its origin is final and plain)
        t[b2] = t[o];
    }
}

// LPT = LMDs per thread (consecutive slots): what a group costs is mostly its barriers and the latency of its dependent
// loads, and both serve twice the bytes with two LMDs per thread (round 3)
template <int NT, int TILE, int LPT>
__global__ __launch_bounds__(NT) void dec_lz_kernel(
    const uint8_t *__restrict__ src, const StreamIn *__restrict__ streams, const StreamPlan *__restrict__ plan,
    const BlockDesc *__restrict__ blocks, const BlockResult *__restrict__ bres,
    const LmdRec *__restrict__ lmds, const uint8_t *__restrict__ lits, uint8_t *dst_all,
    StreamResult *__restrict__ sres, const OutMirror mirror) {
    constexpr int NW = NT / 64;
    constexpr int NS = NT * LPT;     // LMDs (slots) per group
    __shared__ __attribute__((aligned(16))) uint8_t tile[TILE + 32];
    __shared__ uint32_t s_off[NS];   // tile-relative output offset of the LMD (literals first)
    __shared__ uint32_t s_lm[NS];    // l | m << 16
    __shared__ uint32_t s_d[NS];
    __shared__ uint32_t s_lit[NS];   // literal offset inside the block's literal buffer
    __shared__ uint16_t s_org[TILE];  // per-byte origin inside the tile (pointer jumping)
    __shared__ uint32_t s_dm[NT];     // bit k of s_dm[x]: tile byte x + k * NT is produced by an in-tile match
    __shared__ uint32_t s_long[2 * NS];
    __shared__ uint32_t s_scan[2 * NW + 4];
    __shared__ uint32_t s_cnt[8];   // [4]: the tile holds ORG_WRAP entries
    __shared__ int s_status;

    // one workgroup per stream, and a stream's time goes with its size (14 KB .. 700 KB in one Snappy batch): the
    // workgroups take the streams longest first, so that no long stream starts when the others are nearly done
    const uint32_t s = plan[blockIdx.x].turn;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const StreamPlan pl = plan[s];
    if (pl.skip || pl.jump || pl.pipe) return;
    const StreamIn in = streams[s];
    uint8_t *dst = dst_all + in.dst_off;
    // Round 5, the host-pointer calls: every finished piece of the output is stored TWICE -- into the device buffer, which later
    // matches read, and into the pinned host image of that buffer (same offsets), so that the bytes cross the link while the
    // stage runs instead of in a transfer after it (stores into mapped host memory run at the link's rate from any number of
    // waves: scripts/micro/host_store.hip, profiles/r05_host_store.txt). A word per stream tells the host when its bytes are there.
    uint8_t *mir = (mirror.base && !pl.pad && in.dst_off + in.dst_cap <= mirror.span) ? mirror.base + in.dst_off : nullptr;   // (pl.pad: not this stream, says the host)
    uint64_t out_pos = 0;  // bytes produced so far in this stream
    int status = 0;
    uint64_t cy0 = 0, cy1 = 0, cy2 = 0, cy3 = 0, cy4 = 0;
    uint32_t dg_groups = 0, dg_dep = 0, dg_long = 0;
    const uint64_t t_start = __builtin_amdgcn_s_memtime();
    if (tid == 0) s_status = 0;
    __syncthreads();

    for (uint32_t bi = 0; bi < pl.n_blocks && !status; bi++) {
        const BlockDesc d = blocks[pl.blk_base + bi];
        if (d.kind == KIND_RAW) {
            // raw/block.rs:70-92
            if (out_pos + d.n_raw > in.dst_cap) { status = LZFSE_MI_BUFFER_OVERFLOW; break; }
            const uint8_t *p = src + d.src_pos + 8;
            for (uint32_t i = tid; i < d.n_raw; i += NT) dst[out_pos + i] = p[i];
            if (mir) for (uint32_t i = tid; i < d.n_raw; i += NT) mir[out_pos + i] = p[i];
            out_pos += d.n_raw;
            __syncthreads();
            continue;
        }
        if (d.kind == KIND_VXN) {
            __syncthreads();
            if (tid == 0) {
                uint64_t op = out_pos;
                int e = vn_decode_serial(src + d.src_pos, d.src_end - d.src_pos, d.n_raw, d.payload, dst, op, in.dst_cap);
                s_status = e;
                s_cnt[0] = (uint32_t)(op - out_pos);
            }
            __syncthreads();
            status = s_status;
            if (mir && !status) for (uint32_t i = tid; i < s_cnt[0]; i += NT) mir[out_pos + i] = dst[out_pos + i];   // (what thread 0 has just written)
            out_pos += s_cnt[0];
            __syncthreads();
            continue;
        }
        const BlockResult br = bres[pl.blk_base + bi];
        const LmdRec *bl = lmds + d.lmd_base;
        if (br.status || out_pos + d.n_raw > in.dst_cap) {
            // nothing of this block is written: what would the reference's loop have met first?
            const int e = lmds_first_fault<NT>(bl, br.status ? br.ok_until : d.n_lmd, out_pos, in.dst_cap, s_scan);
            status = e ? e : (br.status ? br.status : LZFSE_MI_BUFFER_OVERFLOW);
            break;
        }
        const uint8_t *blit = lits + d.lit_base;
        uint32_t lit_run = 0;
        for (uint32_t g0 = 0; g0 < d.n_lmd && !status;) {
            const uint64_t ta = __builtin_amdgcn_s_memtime();
            // slots tid * LPT + h, h = 0 .. LPT - 1: consecutive LMDs of one thread
            const uint32_t idx0 = g0 + (uint32_t)tid * LPT;
            bool valid[LPT];
            uint32_t rx[LPT], l[LPT], m[LPT], dd[LPT], span[LPT];
            uint32_t sum_l = 0, sum_s = 0;
#pragma unroll
            for (int h = 0; h < LPT; h++) {
                valid[h] = idx0 + h < d.n_lmd;
                const LmdRec r = valid[h] ? bl[idx0 + h] : make_uint2(0, 0);
                rx[h] = r.x; l[h] = r.x & 0xFFFF; m[h] = r.x >> 16; dd[h] = r.y;
                span[h] = l[h] + m[h];
                sum_l += l[h]; sum_s += span[h];
            }
            uint32_t ex_l0, ex_s0, tot_l, tot_s;
            block_excl_scan2<NT>(sum_l, sum_s, ex_l0, ex_s0, tot_l, tot_s, s_scan);
            uint32_t ex_l[LPT], ex_s[LPT];
#pragma unroll
            for (int h = 0; h < LPT; h++) { ex_l[h] = ex_l0; ex_s[h] = ex_s0; ex_l0 += l[h]; ex_s0 += span[h]; }
            // participants: the longest prefix of LMDs whose output fits the tile
            bool part[LPT];
            uint32_t my_parts = 0;
#pragma unroll
            for (int h = 0; h < LPT; h++) { part[h] = valid[h] && (ex_s[h] + span[h] <= (uint32_t)TILE); my_parts += part[h] ? 1u : 0u; }
            if (tid == 0) { s_cnt[0] = 0; s_cnt[1] = 0; s_cnt[2] = 0; s_cnt[3] = 0; s_cnt[4] = 0; }
            __syncthreads();
            {
                const uint32_t wsum = wave_incl_scan(my_parts);
                if (lane == 63 && wsum) atomicAdd(&s_cnt[0], wsum);
            }
            __syncthreads();
            const uint32_t cnt = s_cnt[0];
            // tile length = end of slot cnt - 1; the literals the participants consume
            if (cnt) {
                const uint32_t last = cnt - 1;
                if ((uint32_t)tid == last / LPT) {
#pragma unroll
                    for (int h = 0; h < LPT; h++)
                        if ((uint32_t)h == last % LPT) { s_cnt[1] = ex_s[h] + span[h]; s_cnt[2] = ex_l[h] + l[h]; }
                }
            }
            __syncthreads();
            const uint32_t tile_len = s_cnt[1], lit_used = s_cnt[2];
            const uint64_t tile_base = out_pos;                       // stream-relative
            const uint32_t pad = (uint32_t)((uintptr_t)(dst + tile_base) & 15);  // LDS/global co-alignment
            uint8_t *t = tile + pad;

            const uint64_t tb = __builtin_amdgcn_s_memtime();
            // ---- classify + short copies by the owning lane ----
            bool dep[LPT], far_long[LPT], lit_long[LPT];
            bool bad_any = false;
            uint32_t n_dep = 0, n_long = 0;
#pragma unroll
            for (int h = 0; h < LPT; h++) {
                const uint32_t slot = (uint32_t)tid * LPT + h;
                const uint64_t p_match = tile_base + ex_s[h] + l[h];  // stream-relative position of the match
                const bool bad_d = part[h] && m[h] != 0 && (dd[h] == 0 || (uint64_t)dd[h] > p_match);  // lz/writer.rs:156-178
                bad_any |= bad_d;
                dep[h] = false; far_long[h] = false; lit_long[h] = false;
                if (part[h] && !bad_d) {
                    s_off[slot] = ex_s[h]; s_lm[slot] = rx[h]; s_d[slot] = dd[h]; s_lit[slot] = lit_run + ex_l[h];
                    if (l[h]) {
                        if (l[h] <= SHORT_COPY) {
                            // three 8-byte loads in flight, then byte stores into the tile (the literal
                            // scratch has 256 bytes of slack behind its last byte)
                            const uint8_t *ls = blit + lit_run + ex_l[h];
                            const uint64_t w0 = ld_u64(ls), w1 = l[h] > 8 ? ld_u64(ls + 8) : 0, w2 = l[h] > 16 ? ld_u64(ls + 16) : 0;
                            lds_put24(t + ex_s[h], l[h], w0, w1, w2);
                        } else lit_long[h] = true;
                    }
                    if (m[h]) {
                        // source [p - d, p - d + min(m, d)) ; far when it ends at or before the tile
                        const uint32_t slen = m[h] < dd[h] ? m[h] : dd[h];
                        if (dd[h] >= m[h] && p_match - dd[h] + slen <= tile_base) {
                            if (m[h] <= SHORT_COPY) {
                                const uint8_t *ms = dst + (p_match - dd[h]);
                                // (24 bytes are read wherever they lie inside the stream's output INCLUDING this tile's own place, which is
                                // allocated and not yet written: only the first m are used. Byte by byte such a copy is a round trip per
                                // byte, and a tile's first matches often start a few bytes before it.)
                                if (p_match - dd[h] + 24 <= tile_base + tile_len) {
                                    const uint64_t w0 = ld_u64(ms), w1 = m[h] > 8 ? ld_u64(ms + 8) : 0, w2 = m[h] > 16 ? ld_u64(ms + 16) : 0;
                                    lds_put24(t + ex_s[h] + l[h], m[h], w0, w1, w2);
                                } else {
                                    for (uint32_t k = 0; k < m[h]; k++) t[ex_s[h] + l[h] + k] = ms[k];
                                }
                            } else far_long[h] = true;
                        } else dep[h] = true;
                    }
                }
                n_dep += dep[h] ? 1u : 0u;
                n_long += (lit_long[h] ? 1u : 0u) + (far_long[h] ? 1u : 0u);
            }
            unsigned long long bb = __ballot(bad_any);
            if (bb && lane == 0) atomicOr((int *)&s_status, LZFSE_MI_BAD_D_VALUE);
            // compaction of the long copies
            uint32_t ex_dep, ex_long, tot_dep, tot_long;
            block_excl_scan2<NT>(n_dep, n_long, ex_dep, ex_long, tot_dep, tot_long, s_scan);
#pragma unroll
            for (int h = 0; h < LPT; h++) {
                const uint32_t slot = (uint32_t)tid * LPT + h;
                if (lit_long[h]) s_long[ex_long++] = slot * 2;
                if (far_long[h]) s_long[ex_long++] = slot * 2 + 1;
            }
            __syncthreads();
            if (s_status) { status = s_status; break; }
            const uint64_t tc = __builtin_amdgcn_s_memtime();
            dg_groups++; dg_dep += tot_dep; dg_long += tot_long;
            // ---- long copies: one wave per segment, 64 bytes per step ----
            for (uint32_t q = wave; q < tot_long; q += NW) {
                uint32_t e = s_long[q], slot = e >> 1;
                uint32_t o = s_off[slot], lm = s_lm[slot];
                uint32_t ll = lm & 0xFFFF, mm = lm >> 16;
                if (e & 1) {
                    const uint8_t *ms = dst + (tile_base + o + ll - s_d[slot]);
                    for (uint32_t k = lane; k < mm; k += 64) t[o + ll + k] = ms[k];
                } else {
                    const uint8_t *ls = blit + s_lit[slot];
                    for (uint32_t k = lane; k < ll; k += 64) t[o + k] = ls[k];
                }
            }
            const uint64_t td = __builtin_amdgcn_s_memtime();
            // ---- matches that read the tile (lz/object.rs:27-74 semantics: out[p + k] = out[p + k - d],
            //      overlap allowed): byte-level pointer jumping inside LDS. Every byte of the tile gets an
            //      origin (itself when it is already final); origin <- origin[origin] until all origins are
            //      final bytes, then one gather. Chains of any length collapse in O(log) rounds.
            if (tot_dep) {
                // s_dm: one bit per tile byte that is produced by an in-tile match (its origin s_org[] is meaningful);
                // thread x owns bytes x, x + NT, ... and finds their bits in s_dm[x], so it needs no sweep over the
                // tile to know which of its bytes still have to jump. Bytes without a bit are final.
                static_assert(TILE / NT <= 32 && (NT & (NT - 1)) == 0, "one mask bit per owned byte");
                constexpr uint32_t NTS = 31 - __builtin_clz((unsigned)NT);
                s_dm[tid] = 0;
                if (tid == 0) s_cnt[3] = 0;
                __syncthreads();
#pragma unroll
                for (int h = 0; h < LPT; h++) {
                    if (!dep[h]) continue;
                    const uint32_t mo = ex_s[h] + l[h];                              // tile offset of the match
                    const int64_t so = (int64_t)mo - (int64_t)dd[h];              // tile offset of its source (may be < 0)
                    if (m[h] <= SHORT_COPY) {
                        // the part of the source that lies before the tile: finished output of earlier tiles (one wide read)
                        uint32_t k0 = 0;
                        if (so < 0) {
                            k0 = (uint32_t)min((int64_t)m[h], -so);
                            const uint8_t *ms = dst + ((int64_t)tile_base + so);
                            if (so + 24 <= (int64_t)tile_len) {
                                const uint64_t w0 = ld_u64(ms), w1 = k0 > 8 ? ld_u64(ms + 8) : 0, w2 = k0 > 16 ? ld_u64(ms + 16) : 0;
                                lds_put24(t + mo, k0, w0, w1, w2);
                            } else {
                                for (uint32_t k = 0; k < k0; k++) t[mo + k] = ms[k];
                            }
                        }
                        for (uint32_t k = k0; k < m[h]; k++) {
                            const uint32_t sp = (uint32_t)(so + k);
                            const uint32_t q = mo + k;
                            s_org[q] = (uint16_t)sp; atomicOr(&s_dm[q & (NT - 1)], 1u << (q >> NTS));
                        }
                    } else {
                        s_long[atomicAdd(&s_cnt[3], 1u)] = (uint32_t)tid * LPT + h;
                    }
                }
                __syncthreads();
                const uint32_t nq = s_cnt[3];
                for (uint32_t q = wave; q < nq; q += NW) {
                    const uint32_t slot = s_long[q];
                    const uint32_t lm = s_lm[slot], ddq = s_d[slot];
                    const uint32_t mq = s_off[slot] + (lm & 0xFFFF), mm = lm >> 16;
                    const int64_t sq = (int64_t)mq - (int64_t)ddq;
                    // A match that overlaps itself (d < m: a run, or a short period) repeats its first d source bytes: byte k
                    // is a copy of byte k mod d of the source. Pointing every origin there instead of d bytes back keeps
                    // the chain one hop long where it would be m / d hops (a run of zeros: 2 359, a dozen jumping rounds).
                    const bool wrap = ddq < mm && sq >= 0;
                    if (wrap && lane == 0) s_cnt[4] = 1u;
                    for (uint32_t k = lane; k < mm; k += 64) {
                        const int64_t sp = sq + k;
                        const uint32_t qq = mq + k;
                        if (wrap && k >= ddq) { s_org[qq] = (uint16_t)((mq + k % ddq) | ORG_WRAP); atomicOr(&s_dm[qq & (NT - 1)], 1u << (qq >> NTS)); }
                        else if (sp >= 0) { s_org[qq] = (uint16_t)sp; atomicOr(&s_dm[qq & (NT - 1)], 1u << (qq >> NTS)); }
                        else t[qq] = dst[(int64_t)tile_base + sp];
                    }
                }
                __syncthreads();
                const uint32_t dm = s_dm[tid];
                if (s_cnt[4]) {   // (uniform) runs / short periods in the tile
                    tile_jump_wrap<NT, TILE>(s_org, s_dm, dm, tid);
                    tile_gather_wrap<NT, TILE>(t, s_org, dm, tid);
                } else {
                uint32_t um = dm;  // owned bytes whose origin is not known to be final yet
                for (;;) {
                    // several sweeps per barrier: a sweep may already see what other threads resolved in this round (every
                    // value ever stored names a byte with the same final content), and the barrier is the expensive part
#pragma unroll 1
                    for (int sweep = 0; sweep < JUMP_SWEEPS && um; sweep++)
                        for (uint32_t m2 = um; m2; m2 &= m2 - 1) {
                            const uint32_t k = (uint32_t)__builtin_ctz(m2), b = tid + (k << NTS);
                            const uint32_t o = s_org[b];
                            if (!((s_dm[o & (NT - 1)] >> (o >> NTS)) & 1u)) { um &= ~(1u << k); continue; }
                            const uint32_t o1 = s_org[o];
                            if (!((s_dm[o1 & (NT - 1)] >> (o1 >> NTS)) & 1u)) { s_org[b] = (uint16_t)o1; um &= ~(1u << k); continue; }
                            s_org[b] = s_org[o1];
                        }
                    if (!__syncthreads_or(um != 0)) break;
                }
                for (uint32_t m2 = dm; m2; m2 &= m2 - 1) {
                    const uint32_t b2 = tid + ((uint32_t)__builtin_ctz(m2) << NTS);
                    t[b2] = t[s_org[b2]];
                }
                }
            }
            __syncthreads();
            const uint64_t te = __builtin_amdgcn_s_memtime();
            // ---- write the tile back: head bytes, 16-byte body, tail bytes ----
            {
                uint8_t *g = dst + tile_base;
                uint32_t head = pad ? (16 - pad) : 0;
                if (head > tile_len) head = tile_len;
                uint32_t body = (tile_len - head) & ~15u;
                if (tid < (int)head) g[tid] = t[tid];
                const uint4 *ts = (const uint4 *)(t + head);
                uint4 *gd = (uint4 *)(g + head);
                for (uint32_t k = tid; k < body / 16; k += NT) gd[k] = ts[k];
                uint32_t tail0 = head + body;
                if (tail0 + tid < tile_len && tid < 16) g[tail0 + tid] = t[tail0 + tid];
                if (mir) {   // (the image lies like the buffer: offsets are multiples of 256 in both)
                    uint8_t *mg = mir + tile_base;
                    if (tid < (int)head) mg[tid] = t[tid];
                    uint4 *md = (uint4 *)(mg + head);
                    for (uint32_t k = tid; k < body / 16; k += NT) md[k] = ts[k];
                    if (tail0 + tid < tile_len && tid < 16) mg[tail0 + tid] = t[tail0 + tid];
                }
            }
            out_pos += tile_len;
            // lit_run advances by the literals of the participants only
            lit_run += lit_used;
            g0 += cnt;
            __syncthreads();
            const uint64_t tf = __builtin_amdgcn_s_memtime();
            cy0 += tb - ta; cy1 += tc - tb; cy2 += td - tc; cy3 += te - td; cy4 += tf - te;
        }
    }
    if (tid == 0) {
        StreamResult r;
        r.out_len = out_pos; r.status = status;
        r.groups = dg_groups; r.n_dep = dg_dep; r.n_long = dg_long;
        r.cyc[0] = cy0; r.cyc[1] = cy1; r.cyc[2] = cy2; r.cyc[3] = cy3; r.cyc[4] = cy4;
        r.cyc[5] = __builtin_amdgcn_s_memtime() - t_start;
        sres[s] = r;
    }
    if (mir && mirror.done) {
        // every thread's stores into the image have left the chip before the word that says so is written (system scope: the reader
        // is a host thread): 1 + the stream's length when it came out whole, all ones otherwise
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(mirror.done + s, status ? ~0ull : out_pos + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}


// ------------------------------------------------------------------------------------ LZ stage, several workgroups per stream
//
// One workgroup per stream copies ~0.6 GB/s: with a few dozen streams of some MiB most of the chip idles while the
// stream sets the time. Most of a tile's work does not need the output of the tiles before it: the scan of the LMD
// lengths, the literal copies, the origins of the matches that read the tile itself and their pointer jumping -- about
// three quarters of the cycles. Only the copies from earlier output, the final gather and the write-back do. So K
// workgroups share a stream: a ticket is one group of NT consecutive LMDs of a block (dec_ck_kernel leaves the literal
// and output offset of every group); a workgroup draws the next ticket, does the independent part, waits until all
// earlier tickets are published (done[s] == ticket), does the dependent part, publishes. Tickets are drawn in order by
// running workgroups, so the holder of the ticket somebody waits for is always running: no assumption on residency.
// Errors are raised by a ticket at its turn only, so the first error in stream order is the one reported, as in the
// one-workgroup kernel (and the reference); everybody else sees the flag and leaves.

// Loads of what ANOTHER workgroup of the same XCD wrote during this launch: past this CU's L1 (which may hold the line as it
// was before), served by the XCD's L2, where the writer's stores are once they are acknowledged. Volatile accesses carry the
// sc0 sc1 bits; scripts/micro/l1_inv.hip shows the stale line without them and that buffer_inv sc0 does not help.
typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
__device__ __forceinline__ uint64_t ld_u64_l2(const uint8_t *p) {
    return *(const volatile __attribute__((address_space(1))) u64_unaligned *)(uintptr_t)p;
}
__device__ __forceinline__ uint32_t ld_u32_l2(const uint32_t *p) { return *(const volatile __attribute__((address_space(1))) uint32_t *)(uintptr_t)p; }
__device__ __forceinline__ uint8_t ld_u8_l2(const uint8_t *p) { return *(const volatile __attribute__((address_space(1))) uint8_t *)(uintptr_t)p; }
__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15u;
}


// Self-test of the hand-over dec_lzp_kernel relies on (run once per context before its first use): two workgroups that
// the round-robin placement puts on ONE XCD pass a 1 KiB tile back and forth exactly as the pipelined kernel does --
// producer: plain payload stores, s_waitcnt 0, workgroup barrier, plain flag store; consumer: poll the flag and read the
// payload with sc0 sc1 loads (ld_u32_l2) after having pulled the OLD lines into its L1 with plain loads. The form is
// measured, not architectural (MI355X_MICROARCH.md's hand-over table asks for sc1 stores or an agent-scope release on
// the producer): a single stale word, a flag that never arrives or two different XCC ids switch the pipelined kernel off
// for the context. out[0] = stale words, out[1] = XCC id of A | XCC id of B << 8 | 0x10000 when a wait timed out.
__global__ __launch_bounds__(256) void dec_lzp_selftest_kernel(uint32_t *buf, uint32_t *out, uint32_t rounds) {
    const bool is_a = blockIdx.x == 0, is_b = blockIdx.x == 8;   // (workgroup b runs on XCC b mod 8)
    if (!is_a && !is_b) return;
    uint32_t *tile = buf, *flag_a = buf + 1024, *flag_b = buf + 1056;   // separate 128-byte lines
    const uint32_t tid = threadIdx.x;
    __shared__ uint32_t sh_ok;
    if (tid == 0) { sh_ok = 1; atomicOr(&out[1], is_a ? xcc_id() : (xcc_id() << 8)); }
    uint32_t bad = 0;
    for (uint32_t r = 1; r <= rounds; r++) {
        if (is_a) {
            // the consumer: old tile into L1 first, then wait for the producer's flag, then read past the L1
            const uint32_t old = tile[tid];
            if (old == 0xFFFFFFFFu) bad += 1u << 16;   // (keeps the load)
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_store(flag_a, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                uint32_t spins = 0;
                while (ld_u32_l2(flag_b) != r && ++spins < (1u << 20)) __builtin_amdgcn_s_sleep(1);
                if (spins >= (1u << 20)) sh_ok = 0;
            }
            __syncthreads();
            if (!sh_ok) break;
            if (ld_u32_l2(tile + tid) != (r << 8 | (tid & 255))) bad++;
        } else {
            if (tid == 0) {
                uint32_t spins = 0;
                while (ld_u32_l2(flag_a) != r && ++spins < (1u << 20)) __builtin_amdgcn_s_sleep(1);
                if (spins >= (1u << 20)) sh_ok = 0;
            }
            __syncthreads();
            if (!sh_ok) break;
            tile[tid] = r << 8 | (tid & 255);
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flag_b, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (bad) atomicAdd(&out[0], bad);
    if (tid == 0 && !sh_ok) atomicOr(&out[1], 0x10000u);
}

void launch_dec_lzp_selftest(uint32_t *buf, uint32_t *out, hipStream_t st) {
    hipLaunchKernelGGL(dec_lzp_selftest_kernel, dim3(9), dim3(256), 0, st, buf, out, 48u);
}

constexpr uint32_t LZP_ERR = 0x80000000u;       // done[s]: the stream has failed, its result is written
constexpr uint32_t LZP_SPIN_MAX = 1u << 22;     // polls of done[s] before a workgroup gives up (never reached: a seized launch must still drain)

// literal / output offsets (block-relative) at the start of every 256 LMDs of a block: ck[(lmd_base >> 8) + block + group]
__global__ __launch_bounds__(256) void dec_ck_kernel(const StreamPlan *__restrict__ plan, const BlockDesc *__restrict__ blocks,
                                                     uint32_t n_blocks, const BlockResult *__restrict__ bres,
                                                     const LmdRec *__restrict__ lmds, uint2 *__restrict__ ck) {
    __shared__ uint32_t sh[2 * 4 + 4];
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) return;
    const BlockDesc d = blocks[b];
    if (d.kind != KIND_VX1 && d.kind != KIND_VX2) return;
    const StreamPlan pl = plan[d.stream];
    if (pl.skip || !pl.pipe || bres[b].status) return;
    const LmdRec *bl = lmds + d.lmd_base;
    uint2 *out = ck + (d.lmd_base >> 8) + b;
    uint32_t run_l = 0, run_s = 0;
    for (uint32_t g0 = 0; g0 < d.n_lmd; g0 += 256) {
        const uint32_t idx = g0 + threadIdx.x;
        const LmdRec r = idx < d.n_lmd ? bl[idx] : make_uint2(0, 0);
        const uint32_t l = r.x & 0xFFFF, m = r.x >> 16;
        uint32_t ex_l, ex_s, tot_l, tot_s;
        block_excl_scan2<256>(l, l + m, ex_l, ex_s, tot_l, tot_s, sh);
        if (threadIdx.x == 0) out[g0 >> 8] = make_uint2(run_l, run_s);
        run_l += tot_l; run_s += tot_s;
    }
}

template <int NT, int TILE, int LPT>
__global__ __launch_bounds__(NT) void dec_lzp_kernel(
    const uint8_t *__restrict__ src, const StreamIn *__restrict__ streams, const StreamPlan *__restrict__ plan,
    const uint32_t *__restrict__ mlist, uint32_t n_multi, uint32_t K,
    const BlockDesc *__restrict__ blocks, const BlockResult *__restrict__ bres,
    const LmdRec *__restrict__ lmds, const uint8_t *__restrict__ lits, const uint2 *__restrict__ ck, uint8_t *dst_all,
    StreamResult *__restrict__ sres, uint32_t *__restrict__ state, uint32_t scatter /* diagnostic: pretend the workgroups of a stream sit on different XCDs */) {
    constexpr int NW = NT / 64;
    constexpr int NS = NT * LPT;     // LMDs (slots) of a ticket: LPT consecutive ones per thread (see dec_lz_kernel)
    __shared__ __attribute__((aligned(16))) uint8_t tile[TILE + 32];
    __shared__ uint32_t s_off[NS];
    __shared__ uint32_t s_lm[NS];
    __shared__ uint32_t s_d[NS];
    __shared__ uint32_t s_lit[NS];
    __shared__ uint16_t s_org[TILE];
    __shared__ uint32_t s_dm[NT];
    __shared__ uint32_t s_long[2 * NS];   // slot * 4 + kind: 0 long literal run, 1 long match from earlier output, 2 long match that reads the tile, 3 = 1 but its source is final already
    __shared__ uint32_t s_turn[NS];       // what has to wait for the turn: slot * 2 + (0: kind 1, 1: the part of a kind 2 match that lies before the tile)
    __shared__ uint32_t s_scan[2 * NW + 4];
    __shared__ uint32_t s_cnt[8];   // [4]: the tile holds ORG_WRAP entries
    __shared__ uint32_t s_tk[2];
    __shared__ int s_status;
#ifdef LZFSE_MI_DIAG
    __shared__ uint32_t s_sum;   // checksum of what this ticket wrote / of what the one before it wrote, as this workgroup reads it
#endif

    // the K workgroups of a stream on one XCD (workgroups go to the XCDs round robin): what one writes the next one
    // finds in the same L2
    const uint32_t xcd = blockIdx.x & 7, wq = blockIdx.x >> 3;
    const uint32_t j = (wq / K) * 8 + xcd;
    if (j >= n_multi) return;
    const uint32_t s = mlist[j];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const StreamPlan pl = plan[s];
    const StreamIn in = streams[s];
    uint8_t *dst = dst_all + in.dst_off;
    uint32_t *sw = state + LZP_STATE_WORDS * (size_t)s;
    uint32_t *next = sw + LZP_NEXT, *done = sw + LZP_DONE, *home = sw + LZP_HOME, *bad = sw + LZP_BAD;
    uint64_t cy_setup = 0, cy_ahead = 0, cy_wait = 0, cy_turn = 0, n_tk = 0, cy_far = 0, cy_gat = 0, cy_wb = 0;   // diagnostics: where a ticket's cycles go
    if (pl.n_blocks == 0) {
        if (wq % K == 0 && tid == 0) { StreamResult r = {}; sres[s] = r; }
        return;
    }
    auto n_tickets = [](const BlockDesc &bd) -> uint32_t {
        if (bd.kind != KIND_VX1 && bd.kind != KIND_VX2) return 1u;
        return bd.n_lmd ? (bd.n_lmd + NS - 1) / NS : 1u;
    };
    // all threads: false when the stream has failed (or the wait gave up) -- leave
    auto wait_turn = [&](uint32_t T) -> bool {
        if (tid == 0) {
            uint32_t v, spins = 0;
            while ((v = ld_u32_l2(done)) != T) {
                if (v & LZP_ERR) break;
                // (the slow word: a workgroup on another XCD cannot reach this L2, and says so through memory)
                if ((++spins & 63u) == 0 && (spins > LZP_SPIN_MAX || __hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    if (spins > LZP_SPIN_MAX) __hip_atomic_store(bad, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v = LZP_ERR;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            s_tk[1] = v;
        }
        __syncthreads();
        const uint32_t v = s_tk[1];
        __syncthreads();
        return v == T;
    };
    // all threads, at the ticket's turn: the stream ends here with `status`
    auto fail = [&](int status, uint64_t out_len) {
        if (tid == 0) {
            StreamResult r = {}; r.out_len = out_len; r.status = status; sres[s] = r;
            __hip_atomic_store(done, LZP_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    // all threads, at the end of the ticket's turn: everything this workgroup stored is at L2 before the flag moves
#ifdef LZFSE_MI_DIAG
    // Diagnostic build: the hand-over is CHECKED. A ticket leaves a checksum of the bytes it wrote (position-weighted, from
    // its LDS tile) beside `done`; the next ticket, at its turn, reads those bytes back the way it reads all earlier output
    // (sc0 sc1 loads through the L2) and compares. A difference means the protocol does not hold on this device: the
    // stream is flagged (LZP_BAD = 3), the host decodes the launch's streams again with the one-workgroup kernel and the
    // context gives the pipelined kernel up (scripts/fuzz_gpu.py pipeck runs whole campaigns this way).
    uint32_t my_sum = 0, my_start = 0, my_len = 0;
    auto sum_of = [](uint32_t pos, uint32_t byte) -> uint32_t { return (byte + 1u) * ((pos & 0xFFFFu) + 1u); };
    auto check_prev = [&](uint32_t T) {
        if (T == 0) return;
        const uint32_t *pc = sw + LZP_SUMS + 4 * ((T - 1) & 1);
        const uint32_t want = ld_u32_l2(pc), start = ld_u32_l2(pc + 1), len = ld_u32_l2(pc + 2);
        if (tid == 0) s_sum = 0;
        __syncthreads();
        uint32_t acc = 0;
        for (uint32_t i = tid; i < len; i += NT) acc += sum_of(start + i, ld_u8_l2(dst + start + i));
        atomicAdd(&s_sum, acc);
        __syncthreads();
        if (tid == 0 && len && s_sum != want) __hip_atomic_store(bad, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    };
#endif
    auto publish = [&](uint32_t T, bool last, uint64_t out_len) {
#ifdef LZFSE_MI_DIAG
        if (tid == 0) s_sum = 0;
        __syncthreads();
        atomicAdd(&s_sum, my_sum);
        __syncthreads();
        if (tid == 0) {
            uint32_t *pc = sw + LZP_SUMS + 4 * (T & 1);
            pc[0] = s_sum + (scatter == 2 ? 1u : 0u);   // (scatter = 2: a deliberately wrong sum, to see the check fire)
            pc[1] = my_start; pc[2] = my_len;
        }
        my_sum = 0; my_len = 0;
#endif
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) {
            if (last) { StreamResult r = {}; r.out_len = out_len; r.status = 0; sres[s] = r; }
            __hip_atomic_store(done, T + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // a plain store: into this XCD's L2
        }
    };

    uint32_t bi = 0, tick_lo = 0;
    bool placed = false;
    // Everything before the end of this workgroup's previous ticket is final (it was published in order): copies from
    // there need not wait for the turn. What is in flight is at most the K - 1 tickets in between.
    uint64_t prev_end = 0;
    BlockDesc d = blocks[pl.blk_base];
    uint32_t nt_b = n_tickets(d);
    if (tid == 0) s_status = 0;
    for (;;) {
        const uint64_t t0 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (tid == 0) s_tk[0] = atomicAdd(next, 1u);
        __syncthreads();
        const uint32_t T = s_tk[0];
        while (T >= tick_lo + nt_b) {
            tick_lo += nt_b;
            if (++bi >= pl.n_blocks) break;
            d = blocks[pl.blk_base + bi];
            nt_b = n_tickets(d);
        }
        if (bi >= pl.n_blocks) break;
        const uint32_t g = T - tick_lo;
        const bool last = bi + 1 == pl.n_blocks && g + 1 == nt_b;
        if (!placed) {
            // the hand-over below relies on one L2: every workgroup that works on the stream must be on the same XCD
            if (tid == 0) {
                const uint32_t mine = xcc_id() + 1 + (scatter == 1 ? (wq % K) * 16 : 0);
                const uint32_t was = atomicCAS(home, 0u, mine);
                s_tk[1] = (was == 0 || was == mine) ? 1u : 0u;
            }
            __syncthreads();
            const bool ok = s_tk[1] != 0;
            __syncthreads();
            if (!ok) {
                if (tid == 0) __hip_atomic_store(bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            placed = true;
        }
        const uint64_t blk_end = d.dst_rel + d.n_raw;

        if (d.kind == KIND_RAW) {
            if (!wait_turn(T)) break;
            if (blk_end > in.dst_cap) { fail(LZFSE_MI_BUFFER_OVERFLOW, d.dst_rel); break; }
            const uint8_t *p = src + d.src_pos + 8;
            for (uint32_t i = tid; i < d.n_raw; i += NT) dst[d.dst_rel + i] = p[i];
            publish(T, last, blk_end);
            prev_end = blk_end;
            continue;
        }
        if (d.kind == KIND_VXN) {   // (the host keeps streams with LZVN blocks on the one-workgroup kernel)
            if (!wait_turn(T)) break;
            fail(LZFSE_MI_IO, d.dst_rel);
            break;
        }
        const BlockResult br = bres[pl.blk_base + bi];
        const LmdRec *bl = lmds + d.lmd_base;
        if (br.status || blk_end > in.dst_cap) {
            // nothing of this block is written; its first ticket says what the reference's loop would have met first,
            // the others (their checkpoints may not exist) only wait for that
            if (!wait_turn(T)) break;
            int e = LZFSE_MI_IO;
            if (g == 0) {
                e = lmds_first_fault<NT>(bl, br.status ? br.ok_until : d.n_lmd, d.dst_rel, in.dst_cap, s_scan);
                if (!e) e = br.status ? br.status : LZFSE_MI_BUFFER_OVERFLOW;
            }
            fail(e, d.dst_rel);
            break;
        }
        const uint8_t *blit = lits + d.lit_base;
        const uint2 c0 = d.n_lmd ? ck[(d.lmd_base >> 8) + (pl.blk_base + bi) + ((g * NS) >> 8)] : make_uint2(0, 0);
        uint32_t lit_run = c0.x;
        uint64_t out_pos = d.dst_rel + c0.y;
        const uint32_t g_end = min(d.n_lmd, (g + 1) * (uint32_t)NS);
        bool have_turn = false, gone = false;
        uint64_t t1 = __builtin_amdgcn_s_memtime(), t2 = t1, t3 = t1, t3b = t1;
        cy_setup += t1 - t0; n_tk++;
        for (uint32_t g0 = g * NS; g0 < g_end;) {
            const uint32_t idx0 = g0 + (uint32_t)tid * LPT;
            bool valid[LPT];
            uint32_t rx[LPT], l[LPT], m[LPT], dd[LPT], span[LPT];
            uint32_t sum_l = 0, sum_s = 0;
#pragma unroll
            for (int h = 0; h < LPT; h++) {
                valid[h] = idx0 + h < g_end;
                const LmdRec r = valid[h] ? bl[idx0 + h] : make_uint2(0, 0);
                rx[h] = r.x; l[h] = r.x & 0xFFFF; m[h] = r.x >> 16; dd[h] = r.y;
                span[h] = l[h] + m[h];
                sum_l += l[h]; sum_s += span[h];
            }
            uint32_t ex_l0, ex_s0, tot_l, tot_s;
            block_excl_scan2<NT>(sum_l, sum_s, ex_l0, ex_s0, tot_l, tot_s, s_scan);
            uint32_t ex_l[LPT], ex_s[LPT];
#pragma unroll
            for (int h = 0; h < LPT; h++) { ex_l[h] = ex_l0; ex_s[h] = ex_s0; ex_l0 += l[h]; ex_s0 += span[h]; }
            bool part[LPT];
            uint32_t my_parts = 0;
#pragma unroll
            for (int h = 0; h < LPT; h++) { part[h] = valid[h] && (ex_s[h] + span[h] <= (uint32_t)TILE); my_parts += part[h] ? 1u : 0u; }
            if (tid == 0) { s_cnt[0] = 0; s_cnt[1] = 0; s_cnt[2] = 0; s_cnt[3] = 0; s_cnt[4] = 0; }
            __syncthreads();
            {
                const uint32_t wsum = wave_incl_scan(my_parts);
                if (lane == 63 && wsum) atomicAdd(&s_cnt[0], wsum);
            }
            __syncthreads();
            const uint32_t cnt = s_cnt[0];
            if (cnt) {
                const uint32_t last = cnt - 1;
                if ((uint32_t)tid == last / LPT) {
#pragma unroll
                    for (int h = 0; h < LPT; h++)
                        if ((uint32_t)h == last % LPT) { s_cnt[1] = ex_s[h] + span[h]; s_cnt[2] = ex_l[h] + l[h]; }
                }
            }
            __syncthreads();
            const uint32_t tile_len = s_cnt[1], lit_used = s_cnt[2];
            const uint64_t tile_base = out_pos;
            const uint32_t pad = (uint32_t)((uintptr_t)(dst + tile_base) & 15);
            uint8_t *t = tile + pad;

            // ---- independent part: classification, literals, origins of the matches that read the tile ----
            bool dep[LPT], far[LPT], early[LPT], m_long[LPT], lit_long[LPT];
            bool bad_any = false;
            uint32_t n_dep = 0, n_long = 0;
#pragma unroll
            for (int h = 0; h < LPT; h++) {
                const uint32_t slot = (uint32_t)tid * LPT + h;
                const uint64_t p_match = tile_base + ex_s[h] + l[h];
                const bool bad_d = part[h] && m[h] != 0 && (dd[h] == 0 || (uint64_t)dd[h] > p_match);
                bad_any |= bad_d;
                dep[h] = false; far[h] = false; lit_long[h] = false; early[h] = false;
                if (part[h] && !bad_d) {
                    s_off[slot] = ex_s[h]; s_lm[slot] = rx[h]; s_d[slot] = dd[h]; s_lit[slot] = lit_run + ex_l[h];
                    if (l[h]) {
                        if (l[h] <= SHORT_COPY) {
                            const uint8_t *ls = blit + lit_run + ex_l[h];
                            const uint64_t w0 = ld_u64(ls), w1 = l[h] > 8 ? ld_u64(ls + 8) : 0, w2 = l[h] > 16 ? ld_u64(ls + 16) : 0;
                            lds_put24(t + ex_s[h], l[h], w0, w1, w2);
                        } else lit_long[h] = true;
                    }
                    if (m[h]) {
                        const uint32_t slen = m[h] < dd[h] ? m[h] : dd[h];
                        if (dd[h] >= m[h] && p_match - dd[h] + slen <= tile_base) {
                            far[h] = true;
                            early[h] = p_match - dd[h] + 24 <= prev_end;    // 24 readable bytes of final output
                            if (early[h] && m[h] <= SHORT_COPY) {
                                const uint8_t *ms = dst + (p_match - dd[h]);
                                const uint64_t w0 = ld_u64_l2(ms), w1 = m[h] > 8 ? ld_u64_l2(ms + 8) : 0, w2 = m[h] > 16 ? ld_u64_l2(ms + 16) : 0;
                                lds_put24(t + ex_s[h] + l[h], m[h], w0, w1, w2);
                            }
                            if (early[h] && m[h] > SHORT_COPY) early[h] = p_match - dd[h] + m[h] <= prev_end;   // the whole source
                        } else dep[h] = true;
                    }
                }
                m_long[h] = (far[h] || dep[h]) && m[h] > SHORT_COPY;
                n_dep += dep[h] ? 1u : 0u;
                n_long += (lit_long[h] ? 1u : 0u) + (m_long[h] ? 1u : 0u);
            }
            unsigned long long bb = __ballot(bad_any);
            if (bb && lane == 0) atomicOr((int *)&s_status, LZFSE_MI_BAD_D_VALUE);
            uint32_t ex_dep, ex_long, tot_dep, tot_long;
            block_excl_scan2<NT>(n_dep, n_long, ex_dep, ex_long, tot_dep, tot_long, s_scan);
#pragma unroll
            for (int h = 0; h < LPT; h++) {
                const uint32_t slot = (uint32_t)tid * LPT + h;
                if (lit_long[h]) s_long[ex_long++] = slot * 4;
                if (m_long[h]) s_long[ex_long++] = slot * 4 + (far[h] ? (early[h] ? 3 : 1) : 2);
                const int64_t so = (int64_t)(ex_s[h] + l[h]) - (int64_t)dd[h];
                if (m_long[h] && ((far[h] && !early[h]) || (dep[h] && so < 0))) s_turn[atomicAdd(&s_cnt[3], 1u)] = slot * 2 + (far[h] ? 0u : 1u);
            }
            static_assert(TILE / NT <= 32 && (NT & (NT - 1)) == 0, "one mask bit per owned byte");
            constexpr uint32_t NTS = 31 - __builtin_clz((unsigned)NT);
            s_dm[tid] = 0;
            __syncthreads();
            const bool tile_bad = s_status != 0;
            if (!tile_bad) {
#pragma unroll
                for (int h = 0; h < LPT; h++)
                    if (dep[h] && m[h] <= SHORT_COPY) {
                        const uint32_t mo = ex_s[h] + l[h];
                        const int64_t so = (int64_t)mo - (int64_t)dd[h];
                        for (uint32_t k = 0; k < m[h]; k++) {
                            const int64_t sp = so + k;
                            const uint32_t q = mo + k;
                            if (sp >= 0) { s_org[q] = (uint16_t)sp; atomicOr(&s_dm[q & (NT - 1)], 1u << (q >> NTS)); }
                        }
                    }
                for (uint32_t q = wave; q < tot_long; q += NW) {
                    const uint32_t e = s_long[q], slot = e >> 2, kind = e & 3;
                    const uint32_t o = s_off[slot], lm = s_lm[slot];
                    const uint32_t ll = lm & 0xFFFF, mm = lm >> 16;
                    if (kind == 0) {
                        const uint8_t *ls = blit + s_lit[slot];
                        for (uint32_t k = lane; k < ll; k += 64) t[o + k] = ls[k];
                    } else if (kind == 3) {
                        const uint32_t mq = o + ll;
                        const uint8_t *ms = dst + ((int64_t)tile_base + (int64_t)mq - (int64_t)s_d[slot]);
                        for (uint32_t k = lane * 8; k < mm; k += 512) {   // (the whole source is final: 8 bytes may be read wherever 8 remain)
                            const uint32_t nv = min(8u, mm - k);
                            if (nv == 8) *(u64_unaligned *)(t + mq + k) = ld_u64_l2(ms + k);
                            else for (uint32_t x = 0; x < nv; x++) t[mq + k + x] = ld_u8_l2(ms + k + x);
                        }
                    } else if (kind == 2) {
                        const uint32_t mq = o + ll;
                        const int64_t sq = (int64_t)mq - (int64_t)s_d[slot];
                        const uint32_t ddq = s_d[slot];
                        const bool wrap = ddq < mm && sq >= 0;   // (a match that overlaps itself: ORG_WRAP entries, see above)
                        if (wrap && lane == 0) s_cnt[4] = 1u;
                        for (uint32_t k = lane; k < mm; k += 64) {
                            const int64_t sp = sq + k;
                            const uint32_t qq = mq + k;
                            if (wrap && k >= ddq) { s_org[qq] = (uint16_t)((mq + k % ddq) | ORG_WRAP); atomicOr(&s_dm[qq & (NT - 1)], 1u << (qq >> NTS)); }
                            else if (sp >= 0) { s_org[qq] = (uint16_t)sp; atomicOr(&s_dm[qq & (NT - 1)], 1u << (qq >> NTS)); }
                        }
                    }
                }
            }
            __syncthreads();
            const uint32_t dm = s_dm[tid];
            const bool any_wrap = s_cnt[4] != 0;   // (uniform; stable until the next tile's reset, which lies behind a barrier)
            if (!tile_bad && tot_dep && any_wrap) tile_jump_wrap<NT, TILE>(s_org, s_dm, dm, tid);
            else if (!tile_bad && tot_dep) {
                uint32_t um = dm;
                for (;;) {
#pragma unroll 1
                    for (int sweep = 0; sweep < JUMP_SWEEPS && um; sweep++)
                        for (uint32_t m2 = um; m2; m2 &= m2 - 1) {
                            const uint32_t k = (uint32_t)__builtin_ctz(m2), b = tid + (k << NTS);
                            const uint32_t o = s_org[b];
                            if (!((s_dm[o & (NT - 1)] >> (o >> NTS)) & 1u)) { um &= ~(1u << k); continue; }
                            const uint32_t o1 = s_org[o];
                            if (!((s_dm[o1 & (NT - 1)] >> (o1 >> NTS)) & 1u)) { s_org[b] = (uint16_t)o1; um &= ~(1u << k); continue; }
                            s_org[b] = s_org[o1];
                        }
                    if (!__syncthreads_or(um != 0)) break;
                }
            }

            // ---- the ticket's turn: everything before this tile is written ----
            if (!have_turn) {
                t2 = __builtin_amdgcn_s_memtime();
                if (!wait_turn(T)) { gone = true; break; }
                have_turn = true;
#ifdef LZFSE_MI_DIAG
                check_prev(T);
                my_start = (uint32_t)tile_base;
#endif
                t3 = __builtin_amdgcn_s_memtime();
            }
            if (tile_bad) { fail(LZFSE_MI_BAD_D_VALUE, tile_base); gone = true; break; }
            // long copies from earlier output: one wave per match, 8 bytes per lane and step where 8 bytes of finished
            // output can be read (single bytes at the very end). The first step of a wave's first match is loaded before
            // the short copies below are, so that both are one round trip to the L2.
            const uint32_t n_turn = s_cnt[3];
            auto turn_copy = [&](uint32_t e, bool have_first, uint64_t first) {
                const uint32_t slot = e >> 1;
                const uint32_t lm = s_lm[slot];
                const uint32_t mq = s_off[slot] + (lm & 0xFFFF), mm = lm >> 16;
                const int64_t sq = (int64_t)mq - (int64_t)s_d[slot];
                const uint8_t *ms = dst + ((int64_t)tile_base + sq);
                if (e & 1) {
                    const uint32_t nb = (uint32_t)min((int64_t)mm, -sq);
                    for (uint32_t k = lane; k < nb; k += 64) t[mq + k] = ld_u8_l2(ms + k);
                    return;
                }
                const uint64_t src_pos = (uint64_t)((int64_t)tile_base + sq);
                for (uint32_t k = lane * 8; k < mm; k += 512) {
                    const uint32_t nv = min(8u, mm - k);
                    if (src_pos + k + 8 <= tile_base + tile_len) {   // (as above: bytes past the source are read, not used)
                        const uint64_t w = (have_first && k < 512) ? first : ld_u64_l2(ms + k);
                        if (nv == 8) *(u64_unaligned *)(t + mq + k) = w;
                        else for (uint32_t x = 0; x < nv; x++) t[mq + k + x] = (uint8_t)(w >> (8 * x));
                    } else {
                        for (uint32_t x = 0; x < nv; x++) t[mq + k + x] = ld_u8_l2(ms + k + x);
                    }
                }
            };
            uint32_t e_first = 0;
            uint64_t w_first = 0;
            bool pre_first = false;
            if ((uint32_t)wave < n_turn) {
                e_first = s_turn[wave];
                if (!(e_first & 1)) {
                    const uint32_t slot = e_first >> 1, lm = s_lm[slot];
                    const uint32_t mq = s_off[slot] + (lm & 0xFFFF), mm = lm >> 16;
                    const uint64_t src_pos = tile_base + mq - s_d[slot];
                    if (lane * 8u < mm && src_pos + lane * 8u + 8 <= tile_base + tile_len) { w_first = ld_u64_l2(dst + src_pos + lane * 8u); pre_first = true; }
                }
            }
#pragma unroll
            for (int h = 0; h < LPT; h++) {
                if (!(part[h] && m[h] && !m_long[h] && !early[h])) continue;
                // (24 bytes are read wherever they lie inside this stream's output so far INCLUDING this tile's own place,
                // which is allocated and not yet written: only the first m of them are used. Byte by byte these copies are one
                // round trip each, and a tile's first matches often start a few bytes before it.)
                const uint32_t mo = ex_s[h] + l[h];
                const int64_t so = (int64_t)mo - (int64_t)dd[h];
                const uint64_t p_match = tile_base + mo;
                if (far[h]) {
                    const uint8_t *ms = dst + (p_match - dd[h]);
                    if (p_match - dd[h] + 24 <= tile_base + tile_len) {
                        const uint64_t w0 = ld_u64_l2(ms), w1 = m[h] > 8 ? ld_u64_l2(ms + 8) : 0, w2 = m[h] > 16 ? ld_u64_l2(ms + 16) : 0;
                        lds_put24(t + mo, m[h], w0, w1, w2);
                    } else {
                        for (uint32_t k = 0; k < m[h]; k++) t[mo + k] = ld_u8_l2(ms + k);
                    }
                } else if (so < 0) {
                    const uint32_t nb = (uint32_t)min((int64_t)m[h], -so);
                    const uint8_t *ms = dst + ((int64_t)tile_base + so);
                    if (so + 24 <= (int64_t)tile_len) {
                        const uint64_t w0 = ld_u64_l2(ms), w1 = nb > 8 ? ld_u64_l2(ms + 8) : 0, w2 = nb > 16 ? ld_u64_l2(ms + 16) : 0;
                        lds_put24(t + mo, nb, w0, w1, w2);
                    } else {
                        for (uint32_t k = 0; k < nb; k++) t[mo + k] = ld_u8_l2(ms + k);
                    }
                }
            }
            if ((uint32_t)wave < n_turn) turn_copy(e_first, pre_first, w_first);
            for (uint32_t q = wave + NW; q < n_turn; q += NW) turn_copy(s_turn[q], false, 0);
            __syncthreads();
            const uint64_t u1 = __builtin_amdgcn_s_memtime();
            if (tot_dep && any_wrap) tile_gather_wrap<NT, TILE>(t, s_org, dm, tid);
            else if (tot_dep)
                for (uint32_t m2 = dm; m2; m2 &= m2 - 1) {
                    const uint32_t b2 = tid + ((uint32_t)__builtin_ctz(m2) << NTS);
                    t[b2] = t[s_org[b2]];
                }
            __syncthreads();
            const uint64_t u2 = __builtin_amdgcn_s_memtime();
            cy_far += u1 - t3; cy_gat += u2 - u1; t3b = u2;
#ifdef LZFSE_MI_DIAG
            for (uint32_t i = tid; i < tile_len; i += NT) my_sum += sum_of((uint32_t)tile_base + i, t[i]);
            my_len += tile_len;
#endif
            {
                uint8_t *gp = dst + tile_base;
                uint32_t head = pad ? (16 - pad) : 0;
                if (head > tile_len) head = tile_len;
                const uint32_t body = (tile_len - head) & ~15u;
                if (tid < (int)head) gp[tid] = t[tid];
                const uint4 *ts = (const uint4 *)(t + head);
                uint4 *gd = (uint4 *)(gp + head);
                for (uint32_t k = tid; k < body / 16; k += NT) gd[k] = ts[k];
                const uint32_t tail0 = head + body;
                if (tail0 + tid < tile_len && tid < 16) gp[tail0 + tid] = t[tail0 + tid];
            }
            out_pos += tile_len;
            lit_run += lit_used;
            g0 += cnt;
            // (a group that did not fit one tile goes on: what was just written must have reached the L2 before the next
            // part reads it past the L1)
            if (g0 < g_end) __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
        }
        if (gone) break;
        if (!have_turn) { if (!wait_turn(T)) break; }   // a block without LMDs
        publish(T, last, blk_end);
        prev_end = d.n_lmd ? out_pos : blk_end;
        const uint64_t t4 = __builtin_amdgcn_s_memtime();
        cy_ahead += t2 - t1; cy_wait += t3 - t2; cy_turn += t4 - t3; cy_wb += t4 - t3b;
    }
    if (tid == 0) {
        unsigned long long *q = (unsigned long long *)(sw + LZP_DIAG);
        atomicAdd(q, cy_setup); atomicAdd(q + 1, cy_ahead); atomicAdd(q + 2, cy_wait); atomicAdd(q + 3, cy_turn); atomicAdd(q + 4, n_tk);
        atomicAdd(q + 5, cy_far); atomicAdd(q + 6, cy_gat); atomicAdd(q + 7, cy_wb);
    }
}

// ------------------------------------------------------------------------------------ LZ stage by pointer jumping
//
// For few, large streams one workgroup per stream is far too serial. Every output byte gets an
// origin: itself for a literal, its position minus D for a match byte (lz/writer.rs:144-180 is
// out[p + k] = out[p + k - D]). origin[q] always names a byte with the same final value, so
// origin[q] <- origin[origin[q]] may be applied in any order, in place, until every origin is a
// literal; chains of length n collapse in O(log n) rounds. The last pass gathers the bytes.

constexpr int JUMP_THREADS = 256;
// top bit of an origin: the index names a final byte (a literal, or the resolved end of a chain); such entries are
// never visited again, so a round costs gathers only for the bytes that are still on a chain
constexpr uint32_t JUMP_FINAL = 0x80000000u;

// The origin entries of one LMD per lane, by a whole wave (`on`: this lane has one whose origins are wanted): o = stream-relative
// position of its first byte, literals first. A literal names itself and is final; a match byte names the byte dd back -- or, for a
// match that overlaps itself (dd < m), byte k mod dd of the dd bytes IN FRONT of the match (lz/object.rs:60-74): every byte of it
// points there directly, so that a run of zeros is one hop per match, not one per byte; the bytes of a match with a bad D name
// nothing. Long runs: the whole wave works on one lane's run at a time.
__device__ __forceinline__ void jump_origins_wave(uint32_t *org, uint32_t jb, bool on, uint32_t o, uint32_t l, uint32_t m, uint32_t dd, bool bad_d) {
    const int lane = threadIdx.x & 63;
    const uint32_t p = o + l;
    const bool l_long = on && l > 32, m_long = on && m > 32 && !bad_d;
    if (on && !l_long) for (uint32_t k = 0; k < l; k++) org[o + k] = (jb + o + k) | JUMP_FINAL;
    if (on && !m_long && !bad_d) {
        uint32_t rr = 0;
        for (uint32_t k = 0; k < m; k++) { org[p + k] = jb + p - dd + rr; if (++rr == dd) rr = 0; }
    }
    if (on && bad_d) for (uint32_t k = 0; k < m; k++) org[p + k] = 0xFFFFFFFFu;
    uint64_t ql = __ballot(l_long);
    while (ql) {
        const int L = __builtin_ctzll(ql); ql &= ql - 1;
        const uint32_t qo = read_lane(o, L), qn = read_lane(l, L);
        for (uint32_t k = lane; k < qn; k += 64) org[qo + k] = (jb + qo + k) | JUMP_FINAL;
    }
    uint64_t qm = __ballot(m_long);
    while (qm) {
        const int L = __builtin_ctzll(qm); qm &= qm - 1;
        const uint32_t qp = read_lane(p, L), qn = read_lane(m, L), qd = read_lane(dd, L);
        if (qd >= qn) for (uint32_t k = lane; k < qn; k += 64) org[qp + k] = jb + qp + k - qd;
        else {   // (overlaps itself: k mod qd, stepped by 64 mod qd per turn)
            const uint32_t step = 64u % qd;
            uint32_t rr = (uint32_t)lane % qd;
            for (uint32_t k = lane; k < qn; k += 64) { org[qp + k] = jb + qp - qd + rr; rr += step; if (rr >= qd) rr -= qd; }
        }
    }
}

// dec_fse_kernel<true>, the literal wave once its own stream is decoded (cleanly): the pointer-jumping path's first step for the
// block's LMD records, 64 at a time, as the LMD wave stores them (jready: records stored and visible; jfin: that wave is through)
// -- literal bytes to their places, origins of everything. jdone[0..4) = records taken, a bad D met, bytes and literals they cover:
// the workgroup's first step goes on from there. It stops at a record that would leave the block's bytes or its literals -- such
// a block fails as a whole and its range is filled by the first step.
__device__ void jump_consume_wave(volatile uint32_t *jready, volatile uint32_t *jfin, uint32_t *jdone, const LmdRec *bl, const uint8_t *blit,
                                  uint32_t n_raw, uint32_t n_lit, uint32_t o0, uint8_t *dst, uint32_t *org, uint32_t jb) {
    const int lane = threadIdx.x & 63;
    uint32_t done = 0, run_out = 0, run_lit = 0;
    bool bad = false;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (this wave's own literal bytes: stored before they are read back)
    for (;;) {
        const uint32_t fin = *jfin;       // (before jready: once it is set, jready is final)
        const uint32_t ready = *jready;
        const uint32_t avail = ready - done;
        if (avail >= 64 || (fin && avail)) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const uint32_t cnt = avail < 64 ? avail : 64u;
            const bool valid = (uint32_t)lane < cnt;
            const LmdRec r = valid ? bl[done + lane] : make_uint2(0, 0);
            const uint32_t l = r.x & 0xFFFF, m = r.x >> 16, dd = r.y;
            const uint32_t incl = wave_incl_scan(l + m), tot = read_lane(incl, 63);
            const uint32_t il = wave_incl_scan(l), tot_l = read_lane(il, 63);
            if (tot > n_raw - run_out || tot_l > n_lit - run_lit) break;
            const uint32_t o = o0 + run_out + incl - (l + m);
            const bool bad_d = valid && m != 0 && (dd == 0 || dd > o + l);  // lz/writer.rs:156-178
            bad |= bad_d;
            // the literal bytes
            const uint8_t *ls = blit + run_lit + il - l;
            const bool l_long = l > 32;
            if (!l_long) for (uint32_t k = 0; k < l; k++) dst[o + k] = ls[k];
            uint64_t ql = __ballot(l_long);
            while (ql) {
                const int L = __builtin_ctzll(ql); ql &= ql - 1;
                const uint32_t qo = read_lane(o, L), qn = read_lane(l, L);
                const uint64_t qs = ((uint64_t)read_lane((uint32_t)((uintptr_t)ls >> 32), L) << 32) | read_lane((uint32_t)(uintptr_t)ls, L);
                const uint8_t *q_ls = (const uint8_t *)(uintptr_t)qs;
                for (uint32_t k = lane; k < qn; k += 64) dst[qo + k] = q_ls[k];
            }
            jump_origins_wave(org, jb, valid, o, l, m, dd, bad_d);
            run_out += tot; run_lit += tot_l; done += cnt;
        } else if (fin) break;
        else __builtin_amdgcn_s_sleep(8);
    }
    const bool any_bad = __any(bad);
    if (lane == 0) { jdone[0] = done; jdone[1] = any_bad ? 1u : 0u; jdone[2] = run_out; jdone[3] = run_lit; }
}

// One block's part of the pointer-jumping path's first step: literals (and raw blocks) are written, origins initialised.
// By a whole workgroup of NT threads (uniform arguments): dec_jump_init_kernel's, or -- when every stream of the call takes
// this path -- the block's own dec_fse workgroup right after its entropy stage (dec_fse_kernel<true>), whose literal wave has
// then done the first lmds_done LMDs already (jump_consume_wave: done_out bytes with done_lit literals; pre_bad: it met a bad D).
// br: the block's entropy result (not read for raw blocks); sh: 2 * NT / 64 + 4 words of LDS.
template <int NT>
__device__ __forceinline__ void jump_init_block(const uint32_t b, const BlockDesc &d, const StreamPlan &pl, const BlockResult &br,
                                                const uint8_t *__restrict__ src, const StreamIn *__restrict__ streams,
                                                const LmdRec *__restrict__ lmds, const uint8_t *__restrict__ lits, uint8_t *dst_all,
                                                uint32_t *__restrict__ origin, uint32_t *__restrict__ jerr, uint32_t *sh,
                                                uint32_t lmds_done, uint32_t done_out, uint32_t done_lit, bool pre_bad) {
    if (pl.skip || !pl.jump) return;
    const StreamIn in = streams[d.stream];
    uint8_t *dst = dst_all + in.dst_off;
    uint32_t *org = origin + pl.jbase;
    const uint32_t jb = (uint32_t)pl.jbase;
    const uint32_t bi = b - (uint32_t)pl.blk_base;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t o0 = (uint32_t)d.dst_rel;
    // Every origin entry is defined before the rounds read it (all bits set = final, names nothing), and by the workgroup of the
    // block it lies in -- round 4; a fill of the whole array before: 0.09 ms for a 64 MiB stream: the up to three padding entries
    // behind a stream, the bytes of a block that failed, the bytes of a match with a bad D.
    if (bi + 1 == pl.n_blocks && tid < 3) {
        const uint32_t e0 = o0 + d.n_raw + (uint32_t)tid;
        if (e0 < ((o0 + d.n_raw + 3u) & ~3u)) org[e0] = 0xFFFFFFFFu;
    }
    if (d.kind == KIND_RAW) {
        const uint8_t *p = src + d.src_pos + 8;
        for (uint32_t i = tid; i < d.n_raw; i += NT) { dst[o0 + i] = p[i]; org[o0 + i] = (jb + o0 + i) | JUMP_FINAL; }
        return;
    }
    const LmdRec *bl = lmds + d.lmd_base;
    if (br.status) {
        for (uint32_t i = tid; i < d.n_raw; i += NT) org[o0 + i] = 0xFFFFFFFFu;
        int e = lmds_first_fault<NT>(bl, br.ok_until, o0, in.dst_cap, sh);  // (a damaged block may overrun it)
        if (!e) e = br.status;
        if (tid == 0) atomicMin(&jerr[d.stream], (bi << 8) | (uint32_t)e);
        return;
    }
    const uint8_t *blit = lits + d.lit_base;
    uint32_t run_lit = done_lit, run_out = done_out;
    bool bad = pre_bad;
    for (uint32_t g0 = lmds_done; g0 < d.n_lmd; g0 += NT) {
        const uint32_t idx = g0 + tid;
        const bool valid = idx < d.n_lmd;
        const LmdRec r = valid ? bl[idx] : make_uint2(0, 0);
        const uint32_t l = r.x & 0xFFFF, m = r.x >> 16, dd = r.y;
        uint32_t ex_l, ex_s, tot_l, tot_s;
        block_excl_scan2<NT>(l, l + m, ex_l, ex_s, tot_l, tot_s, sh);
        const uint32_t o = o0 + run_out + ex_s;       // stream-relative position of the LMD's first byte
        const uint32_t p = o + l;                     // ... of its match
        const bool bad_d = valid && m != 0 && (dd == 0 || dd > p);  // lz/writer.rs:156-178
        bad |= bad_d;
        // the literal bytes
        const uint8_t *ls = blit + run_lit + ex_l;
        const bool l_long = l > 32;
        if (!l_long) for (uint32_t k = 0; k < l; k++) dst[o + k] = ls[k];
        uint64_t ql = __ballot(l_long);
        while (ql) {
            const int L = __builtin_ctzll(ql); ql &= ql - 1;
            const uint32_t qo = read_lane(o, L), qn = read_lane(l, L);
            const uint64_t qs = ((uint64_t)read_lane((uint32_t)((uintptr_t)ls >> 32), L) << 32) | read_lane((uint32_t)(uintptr_t)ls, L);
            const uint8_t *q_ls = (const uint8_t *)(uintptr_t)qs;
            for (uint32_t k = lane; k < qn; k += 64) dst[qo + k] = q_ls[k];
        }
        // the origins
        jump_origins_wave(org, jb, valid, o, l, m, dd, bad_d);
        run_lit += tot_l; run_out += tot_s;
    }
    if (__any(bad) && lane == 0) atomicMin(&jerr[d.stream], (bi << 8) | (uint32_t)LZFSE_MI_BAD_D_VALUE);
}

// one workgroup per block
__global__ __launch_bounds__(JUMP_THREADS) void dec_jump_init_kernel(
    const uint8_t *__restrict__ src, const StreamIn *__restrict__ streams, const StreamPlan *__restrict__ plan,
    const BlockDesc *__restrict__ blocks, uint32_t n_blocks, const BlockResult *__restrict__ bres,
    const LmdRec *__restrict__ lmds, const uint8_t *__restrict__ lits, uint8_t *dst_all, uint32_t *__restrict__ origin,
    uint32_t *__restrict__ jerr) {
    __shared__ uint32_t sh[2 * (JUMP_THREADS / 64) + 4];
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) return;
    const BlockDesc d = blocks[b];
    jump_init_block<JUMP_THREADS>(b, d, plan[d.stream], bres[b], src, streams, lmds, lits, dst_all, origin, jerr, sh, 0u, 0u, 0u, false);
}

// Chains collapsed chunk by chunk in LDS before the global rounds: a workgroup takes 16 Ki consecutive origins (64 KB of
// LDS) and jumps every entry whose origin lies in the same chunk until it is final or points in front of the chunk. An
// origin mostly points a match distance back, so most hops of most chains stay inside a chunk, where a hop is an LDS read
// instead of a gather from a 4-byte-per-output-byte array far larger than any L2. Any intermediate state is valid
// (an origin always names a byte with the same final value), so no ordering between threads is needed.
#ifndef LZMI_JC_N
#define LZMI_JC_N 16384
#endif
constexpr uint32_t JC_N = LZMI_JC_N;
__global__ __launch_bounds__(1024) void dec_jump_collapse_kernel(uint32_t *__restrict__ origin, uint64_t total) {
    __shared__ uint32_t o[JC_N];
    const uint64_t c0 = (uint64_t)blockIdx.x * JC_N;
    if (c0 >= total) return;
    const uint32_t n = total - c0 < JC_N ? (uint32_t)(total - c0) : JC_N;
    const uint32_t base = (uint32_t)c0;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) o[i] = origin[c0 + i];
    __syncthreads();
    for (;;) {
        bool more = false;
        for (uint32_t i = threadIdx.x; i < n; i += 1024) {
            uint32_t cur = o[i];
#pragma unroll
            for (int h = 0; h < 4; h++) {
                if ((cur & JUMP_FINAL) || cur < base || cur - base >= n) break;   // final, or points out of the chunk
                cur = o[cur - base];
            }
            o[i] = cur;
            if (!(cur & JUMP_FINAL) && cur >= base && cur - base < n) more = true;
        }
        if (!__syncthreads_or(more)) break;
    }
    for (uint32_t i = threadIdx.x; i < n; i += 1024) origin[c0 + i] = o[i];
}

// one jumping round over all origins; exits at once when the previous round changed nothing
__global__ __launch_bounds__(256) void dec_jump_round_kernel(uint32_t *__restrict__ origin, uint64_t total, uint32_t *__restrict__ flags,
                                                             uint32_t round) {
    if (round > 0 && flags[round - 1] == 0) return;
    bool changed = false;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    // Workgroups go to the 8 XCDs round-robin. In one pass of the grid every XCD takes ONE contiguous eighth of the
    // entries in flight: an origin mostly points a match distance back (a few KB .. 256 KB), so the hops of an XCD's
    // workgroups land in lines its own L2 already holds, instead of every L2 holding a copy of everything (the grid is a
    // multiple of 8 workgroups).
    const uint64_t per_xcd = stride / 8;
    const uint64_t first = (uint64_t)(blockIdx.x & 7) * per_xcd + (uint64_t)(blockIdx.x >> 3) * blockDim.x + threadIdx.x;
    for (uint64_t q = first; q < total + stride; q += stride) {
        if (q >= total) break;
        uint32_t cur = origin[q];
        if (cur & JUMP_FINAL) continue;
        // up to three hops; an entry carrying JUMP_FINAL holds the final byte of its chain (itself for a literal)
#ifndef LZMI_JR_HOPS
#define LZMI_JR_HOPS 3
#endif
#pragma unroll
        for (int h = 0; h < LZMI_JR_HOPS; h++) {
            if (cur >= total) { cur = (uint32_t)q | JUMP_FINAL; break; }  // never taken: every entry is defined (see launch_dec_jump)
            cur = origin[cur];
            if (cur & JUMP_FINAL) break;
        }
        origin[q] = cur;
        if (!(cur & JUMP_FINAL)) changed = true;
    }
    if (__any(changed) && (threadIdx.x & 63) == 0) flags[round] = 1;
}

// gather: 4 bytes per thread
__global__ __launch_bounds__(JUMP_THREADS) void dec_jump_apply_kernel(const StreamIn *__restrict__ streams, const StreamPlan *__restrict__ plan,
                                                                      const BlockDesc *__restrict__ blocks, uint32_t n_blocks,
                                                                      const BlockResult *__restrict__ bres, uint8_t *dst_all,
                                                                      const uint32_t *__restrict__ origin, uint32_t *__restrict__ jerr) {
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) return;
    const BlockDesc d = blocks[b];
    const StreamPlan pl = plan[d.stream];
    if (pl.skip || !pl.jump || d.kind == KIND_RAW) return;
    if (jerr[d.stream] != 0xFFFFFFFFu) return;
    const StreamIn in = streams[d.stream];
    uint8_t *dst = dst_all + in.dst_off;
    const uint32_t *org = origin + pl.jbase;
    const uint32_t jb = (uint32_t)pl.jbase;
    const uint32_t o0 = (uint32_t)d.dst_rel, n = d.n_raw;
    for (uint32_t i = threadIdx.x * 4; i < n; i += JUMP_THREADS * 4) {
        const uint32_t q = o0 + i;
        uint32_t w = 0;
        const uint32_t cnt = n - i < 4 ? n - i : 4;
        bool loose = false;
        for (uint32_t k = 0; k < cnt; k++) {
            const uint32_t og = org[q + k];
            loose |= !(og & JUMP_FINAL);
            w |= (uint32_t)dst[(og & ~JUMP_FINAL) - jb] << (8 * k);
        }
        // every chain is resolved by now (launch_dec_jump sizes the rounds for that); should one not be, the stream fails
        // instead of carrying a wrong byte
        if (loose) atomicMin(&jerr[d.stream], (0xFFFFFEu << 8) | (uint32_t)LZFSE_MI_IO);
        if (cnt == 4) __builtin_memcpy(dst + q, &w, 4);
        else for (uint32_t k = 0; k < cnt; k++) dst[q + k] = (uint8_t)(w >> (8 * k));
    }
}

__global__ void dec_jump_finish_kernel(const StreamPlan *__restrict__ plan, const StreamWalk *__restrict__ walk, uint32_t n_streams,
                                       const uint32_t *__restrict__ jerr, StreamResult *__restrict__ sres) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    const StreamPlan pl = plan[s];
    if (pl.skip || !pl.jump) return;
    StreamResult r;
    const uint32_t e = jerr[s];
    r.status = e == 0xFFFFFFFFu ? 0 : (int32_t)(e & 0xFF);
    r.out_len = r.status ? 0 : walk[s].raw_total;
    r.groups = 0; r.n_dep = 0; r.n_long = 0;
    for (int k = 0; k < 6; k++) r.cyc[k] = 0;
    sres[s] = r;
}

// ------------------------------------------------------------------------------------ launchers

void launch_dec_walk(bool emit, const uint8_t *src, const StreamIn *streams, uint32_t n_streams,
                     StreamWalk *walk, const StreamPlan *plan, BlockDesc *blocks, const uint32_t *settled, hipStream_t st) {
    dim3 grid((n_streams + 63) / 64), block(64);
    if (emit) hipLaunchKernelGGL(dec_walk_kernel<true>, grid, block, 0, st, src, streams, n_streams, walk, plan, blocks, settled);
    else hipLaunchKernelGGL(dec_walk_kernel<false>, grid, block, 0, st, src, streams, n_streams, walk, plan, blocks, settled);
}

uint32_t fastwalk_cap() { return FW_CAP; }

// parallel walk of the n_elig streams listed in `elig` (max_len: the longest of them); count: n_elig zeroed words;
// cand: n_elig * fastwalk_cap() entries; settled: one zeroed word per stream of the batch
void launch_dec_fastwalk(const uint8_t *src, const StreamIn *streams, const uint32_t *elig, uint32_t n_elig, uint64_t max_len,
                         uint32_t *count, uint2 *cand, StreamWalk *walk, BlockDesc *cache, uint32_t *settled, hipStream_t st) {
    if (!n_elig) return;
    hipLaunchKernelGGL(dec_scan_kernel, dim3((uint32_t)((max_len + FW_SCAN_BYTES - 1) / FW_SCAN_BYTES), n_elig), dim3(256), 0, st, src, streams,
                       elig, count, cand);
    hipLaunchKernelGGL(dec_rank_kernel, dim3(n_elig), dim3(1024), 0, st, src, streams, elig, count, cand, walk, cache, settled);
}

// up to six fills of dwords in one launch (sizes in BYTES, multiples of 4; a null pointer or a size of 0 is skipped)
void launch_dec_fills(void *const *ptrs, const uint64_t *bytes, const uint32_t *values, int count, hipStream_t st) {
    FillSet f = {};
    uint64_t total = 0;
    int k = 0;
    for (int q = 0; q < count; q++) {
        if (!ptrs[q] || !bytes[q]) continue;
        if (bytes[q] / 4 > 0x7FFFFFFFull || k == 6) { (void)hipMemsetD32Async((hipDeviceptr_t)ptrs[q], (int)values[q], bytes[q] / 4, st); continue; }
        f.p[k] = (uint32_t *)ptrs[q]; f.n[k] = (uint32_t)(bytes[q] / 4); f.v[k] = values[q]; total += f.n[k]; k++;
    }
    if (!total) return;
    if (total > 0xFFFFFFFFull - 512) {   // (cannot happen with the sizes of a decode call; plain fills then)
        for (int q = 0; q < k; q++) (void)hipMemsetD32Async((hipDeviceptr_t)f.p[q], (int)f.v[q], f.n[q], st);
        return;
    }
    hipLaunchKernelGGL(dec_fill_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, f);
}

void launch_dec_emit(const StreamIn *streams, uint32_t n_streams, const StreamPlan *plan, const BlockDesc *cache, uint64_t cache_total,
                     BlockDesc *blocks, hipStream_t st) {
    if (!cache_total) return;
    hipLaunchKernelGGL(dec_emit_kernel, dim3((uint32_t)((cache_total + 255) / 256)), dim3(256), 0, st, streams, n_streams, plan, cache,
                       cache_total, blocks);
}

void launch_dec_fse(const uint8_t *src, uint64_t src_total, const BlockDesc *blocks, uint32_t n_blocks,
                    uint8_t *lit_out, LmdRec *lmd_out, BlockResult *results, uint32_t *order_hist, uint32_t *order,
                    const JumpFuse *jf, hipStream_t st) {
    if (!n_blocks) return;
    // order_hist: 64 zeroed words; order: n_blocks words
    if (n_blocks <= ORDER_ONE_MAX) hipLaunchKernelGGL(dec_order_one_kernel, dim3(1), dim3(256), 0, st, blocks, n_blocks, order);
    else {
        hipLaunchKernelGGL(dec_order_count_kernel, dim3((n_blocks + 255) / 256), dim3(256), 0, st, blocks, n_blocks, order_hist);
        hipLaunchKernelGGL(dec_order_scan_kernel, dim3(1), dim3(64), 0, st, order_hist);
        hipLaunchKernelGGL(dec_order_place_kernel, dim3((n_blocks + 255) / 256), dim3(256), 0, st, blocks, n_blocks, order_hist, order);
    }
    if (jf) {
        hipLaunchKernelGGL(dec_fse_kernel<true>, dim3(n_blocks), dim3(FSE_THREADS), 0, st, src, src_total, blocks, n_blocks, lit_out, lmd_out,
                           results, order, jf->streams, jf->plan, jf->dst, jf->origin, jf->jerr);
    } else
        hipLaunchKernelGGL(dec_fse_kernel<false>, dim3(n_blocks), dim3(FSE_THREADS), 0, st, src, src_total, blocks, n_blocks, lit_out, lmd_out,
                           results, order, (const StreamIn *)nullptr, (const StreamPlan *)nullptr, (uint8_t *)nullptr, (uint32_t *)nullptr,
                           (uint32_t *)nullptr);
}

void launch_dec_lz(int variant, const uint8_t *src, const StreamIn *streams, const StreamPlan *plan,
                   uint32_t n_streams, const BlockDesc *blocks, const BlockResult *bres, const LmdRec *lmds,
                   const uint8_t *lits, uint8_t *dst, StreamResult *sres, const OutMirror &mirror, hipStream_t st) {
    if (!n_streams) return;
    if (variant == 0)
        hipLaunchKernelGGL((dec_lz_kernel<256, 8192, LZ_LPT>), dim3(n_streams), dim3(256), 0, st, src, streams, plan,
                           blocks, bres, lmds, lits, dst, sres, mirror);
    else
        hipLaunchKernelGGL((dec_lz_kernel<1024, 32768, LZ_LPT>), dim3(n_streams), dim3(1024), 0, st, src, streams, plan,
                           blocks, bres, lmds, lits, dst, sres, mirror);
}

void launch_dec_lzp(int variant, uint32_t K, int lpt, const uint8_t *src, const StreamIn *streams, const StreamPlan *plan,
                    const uint32_t *mlist, uint32_t n_multi, const BlockDesc *blocks, uint32_t n_blocks, const BlockResult *bres,
                    const LmdRec *lmds, const uint8_t *lits, uint2 *ck, uint8_t *dst, StreamResult *sres, uint32_t *state,
                    uint32_t scatter, hipStream_t st) {
    if (!n_multi || !K) return;
    hipLaunchKernelGGL(dec_ck_kernel, dim3(n_blocks), dim3(256), 0, st, plan, blocks, n_blocks, bres, lmds, ck);
    const uint32_t grid = ((n_multi + 7) / 8) * 8 * K;
    // lpt: LMDs per thread = tickets of NT or 2 NT LMDs. Two halve a large stream's hand-overs; a stream of a few thousand
    // LMDs would be left with fewer tickets than workgroups (html, 3 691 LMDs: 4 tickets of 1 024 keep K = 4 busy, 2 of 2 048 do not)
    if (variant == 0) {
        if (lpt == 2)
            hipLaunchKernelGGL((dec_lzp_kernel<256, 8192, 2>), dim3(grid), dim3(256), 0, st, src, streams, plan, mlist, n_multi, K,
                               blocks, bres, lmds, lits, ck, dst, sres, state, scatter);
        else
            hipLaunchKernelGGL((dec_lzp_kernel<256, 8192, 1>), dim3(grid), dim3(256), 0, st, src, streams, plan, mlist, n_multi, K,
                               blocks, bres, lmds, lits, ck, dst, sres, state, scatter);
    } else {
        if (lpt == 2)
            hipLaunchKernelGGL((dec_lzp_kernel<1024, 32768, 2>), dim3(grid), dim3(1024), 0, st, src, streams, plan, mlist, n_multi, K,
                               blocks, bres, lmds, lits, ck, dst, sres, state, scatter);
        else
            hipLaunchKernelGGL((dec_lzp_kernel<1024, 32768, 1>), dim3(grid), dim3(1024), 0, st, src, streams, plan, mlist, n_multi, K,
                               blocks, bres, lmds, lits, ck, dst, sres, state, scatter);
    }
}

void launch_dec_jump(const uint8_t *src, const StreamIn *streams, const StreamPlan *plan, const StreamWalk *walk, uint32_t n_streams,
                     const BlockDesc *blocks, uint32_t n_blocks, const BlockResult *bres, const LmdRec *lmds, const uint8_t *lits,
                     uint8_t *dst, uint32_t *origin, uint64_t total, uint32_t *jerr, uint32_t *flags, StreamResult *sres,
                     lzfse_mi_ctx *c, bool init_done, hipStream_t st) {
    if (!n_blocks || !total) return;
    if (!init_done) {
        StageTimer t(c, "dec_jump_init");
        // (every origin entry is defined before the rounds read it: jump_init_block)
        hipLaunchKernelGGL(dec_jump_init_kernel, dim3(n_blocks), dim3(JUMP_THREADS), 0, st, src, streams, plan, blocks, n_blocks, bres, lmds,
                           lits, dst, origin, jerr);
    }
    {
        StageTimer t(c, "dec_jump_rounds");
        hipLaunchKernelGGL(dec_jump_collapse_kernel, dim3((uint32_t)((total + JC_N - 1) / JC_N)), dim3(1024), 0, st, origin, total);
        uint32_t grid = (uint32_t)std::min<uint64_t>((total + 255) / 256, 256ull * 16);
        grid = (grid + 7) & ~7u;   // (dec_jump_round_kernel: eight equal parts)
        // every round collapses chains by 4x (three dependent hops): 4^16 covers any stream below 4 GiB. After the
        // chunk-wise collapse a hop leaves its chunk, so a chain has at most as many hops as there are chunks.
        uint32_t n_rounds = 2;
        for (uint64_t reach = 1; reach < (total + JC_N - 1) / JC_N && n_rounds < 17; reach *= 4) n_rounds++;
        for (uint32_t r = 0; r < n_rounds; r++)
            hipLaunchKernelGGL(dec_jump_round_kernel, dim3(grid), dim3(256), 0, st, origin, total, flags, r);
    }
    {
        StageTimer t(c, "dec_jump_apply");
        hipLaunchKernelGGL(dec_jump_apply_kernel, dim3(n_blocks), dim3(JUMP_THREADS), 0, st, streams, plan, blocks, n_blocks, bres, dst,
                           origin, jerr);
        hipLaunchKernelGGL(dec_jump_finish_kernel, dim3((n_streams + 63) / 64), dim3(64), 0, st, plan, walk, n_streams, jerr, sres);
    }
}

}  // namespace lzmi
