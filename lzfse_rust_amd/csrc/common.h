// Shared host/device definitions for the MI355X LZFSE codec (gfx950 only).
// Wire-format constants are behaviour-defining and equal the reference's:
//   src/fse/constants.rs:22-69, src/encode/constants.rs:3-10, src/encode/history.rs:10-13,
//   src/base/magic_bytes.rs:3-7.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lzfse_mi.h"

namespace lzmi {

constexpr uint32_t LMDS_PER_BLOCK = 10000;
constexpr uint32_t LITERALS_PER_BLOCK = 40000;
constexpr uint32_t L_SYMBOLS = 20, M_SYMBOLS = 20, D_SYMBOLS = 64, U_SYMBOLS = 256;
constexpr uint32_t L_STATES = 64, M_STATES = 64, D_STATES = 256, U_STATES = 1024;
constexpr uint32_t MAX_L_VALUE = 315, MAX_M_VALUE = 2359, MAX_D_VALUE = 262139;
constexpr uint32_t N_WEIGHTS = 360;
constexpr uint32_t V1_HEADER_SIZE = 50, V2_HEADER_SIZE = 32;
constexpr uint32_t V1_WEIGHT_PAYLOAD_BYTES = 722, V2_WEIGHT_PAYLOAD_BYTES_MAX = 630;
constexpr uint32_t GOOD_MATCH_LEN = 40, RAW_CUTOFF = 20, RAW_LIMIT = 0x4000, VN_CUTOFF = 0x1000;
constexpr uint32_t HASH_BITS = 14;
constexpr uint32_t MAGIC_EOS = 0x24787662u, MAGIC_RAW = 0x2D787662u, MAGIC_VX1 = 0x31787662u,
                   MAGIC_VX2 = 0x32787662u, MAGIC_VXN = 0x6E787662u;

enum BlockKind : uint32_t { KIND_VX2 = 0, KIND_VX1 = 1, KIND_RAW = 2, KIND_VXN = 3 };

// One independent LZFSE stream of a batch (offsets into the batch's device buffers).
struct StreamIn {
    uint64_t src_off, src_len, dst_off, dst_cap;
    uint64_t cache_off;   // decode: first entry of the stream in the header-walk cache (BlockDesc with stream-relative bases)
    uint64_t cache_cap;   // ... and its capacity in blocks; longer streams are re-walked when the descriptors are emitted
};

// Result of the header walk for one stream (decode).
struct StreamWalk {
    uint64_t n_lmds;     // sum of lmd.num over FSE blocks
    uint64_t n_lits;     // sum of (padded) literal.num over FSE blocks
    uint64_t raw_total;  // sum of n_raw_bytes over all blocks
    uint32_t n_blocks;
    int32_t status;      // first header-level error (0 = walk reached bvx$ cleanly)
    uint32_t err_block;  // block index at which `status` was raised
    uint32_t n_vxn;      // number of bvxn blocks (decoded serially by the tile LZ kernel)
    uint32_t detail;     // payload of the error kind: BadBlock(magic), BadLmdCount(num), BadLiteralCount(num)
    uint32_t pad;
};

// Per-stream bases assigned by the host after the counting walk.
struct StreamPlan {
    uint64_t blk_base, lmd_base, lit_base;
    uint32_t n_blocks;
    int32_t skip;  // != 0: stream not decoded (status already final)
    uint64_t jbase;  // pointer-jumping LZ path: first entry of the stream in the origin array
    int32_t jump;    // != 0: LZ stage by pointer jumping (large streams), else by the tile kernel
    uint32_t turn;   // plan[b].turn = the stream workgroup b of the tile kernel decodes: longest streams first
    uint32_t pipe;   // != 0: LZ stage by the pipelined tile kernel (several workgroups per stream), else one workgroup
    uint32_t pad;    // != 0: no second copy of this stream's output in the host image (OutMirror, internal.h)
};

struct BlockDesc {
    uint64_t src_pos;   // absolute offset of the block magic in d_src
    uint64_t src_end;   // absolute end of the owning stream in d_src
    uint64_t dst_rel;   // output offset relative to the stream's dst_off (from header sums)
    uint64_t lmd_base;  // index into the LMD scratch
    uint64_t lit_base;  // byte offset into the literal scratch
    uint32_t stream, kind;
    uint32_t n_lmd, n_lit, n_raw, payload;  // payload: bvxn n_payload_bytes
};

struct BlockResult {
    int32_t status;
    uint32_t sum_l, sum_m;
    // status != 0 only: number of LMDs the reference's decode loop (fse_core.rs:103-131) gets through before this
    // status is raised (0 for everything raised before the loop). A bad D among them is raised first (BadDValue).
    uint32_t ok_until;
};

struct StreamResult {
    uint64_t out_len;
    int32_t status;
    uint32_t groups;       // diagnostics: LZ groups processed, dependent matches, long copies
    uint32_t n_dep, n_long;
    uint64_t cyc[6];       // diagnostics: cycles per LZ phase (scan, short copies, long copies, dependent, write-back, total)
};

// Decoded LMD record handed from the entropy stage to the LZ stage: l | m << 16, d
// (d already substituted: lmd_type.rs:153-160).
typedef uint2 LmdRec;

#define LZMI_HD __host__ __device__
// ---- unaligned little-endian loads (global memory tolerates any alignment on gfx950) ----
LZMI_HD __forceinline__ uint32_t ld_u32(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
LZMI_HD __forceinline__ uint64_t ld_u64(const uint8_t *p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
LZMI_HD __forceinline__ uint16_t ld_u16(const uint8_t *p) {
    uint16_t v;
    __builtin_memcpy(&v, p, 2);
    return v;
}
LZMI_HD __forceinline__ uint32_t mask32(uint32_t n) { return n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u); }

// L/M/D symbol tables (fse/constants.rs:127-134,159-166,305-321)
LZMI_HD __forceinline__ uint32_t l_extra_bits(uint32_t s) { return s < 16 ? 0u : (s == 16 ? 2u : s == 17 ? 3u : s == 18 ? 5u : 8u); }
LZMI_HD __forceinline__ uint32_t l_base_value(uint32_t s) { return s < 16 ? s : (s == 16 ? 16u : s == 17 ? 20u : s == 18 ? 28u : 60u); }
LZMI_HD __forceinline__ uint32_t m_extra_bits(uint32_t s) { return s < 16 ? 0u : (s == 16 ? 3u : s == 17 ? 5u : s == 18 ? 8u : 11u); }
LZMI_HD __forceinline__ uint32_t m_base_value(uint32_t s) { return s < 16 ? s : (s == 16 ? 16u : s == 17 ? 24u : s == 18 ? 56u : 312u); }
LZMI_HD __forceinline__ uint32_t d_extra_bits(uint32_t s) { return s >> 2; }
// base(s) = sum_{i<s} 2^(i/4) = 4*(2^q - 1) + r*2^q with q = s/4, r = s%4
LZMI_HD __forceinline__ uint32_t d_base_value(uint32_t s) {
    uint32_t q = s >> 2, r = s & 3;
    return 4u * ((1u << q) - 1u) + r * (1u << q);
}

// ---- FSE block headers ----
struct FseHeader {
    uint32_t n_raw, lit_num, lit_payload, lit_bits, lmd_num, lmd_payload, lmd_bits;
    uint32_t lit_state[4];
    uint32_t lmd_state[3];
    uint32_t hdr_size;  // header + weights (bytes before the literal payload)
    uint32_t n_weight;  // weight payload bytes
};

// fse/block.rs:218-226,267-283,324-341 (order: lmd, literal, raw byte count)
LZMI_HD inline int fse_validate(const FseHeader &h) {
    uint32_t lmd_limit = 1024u + 8u + (h.lmd_num * 54u + 7u) / 8u;
    if (h.lmd_num > LMDS_PER_BLOCK || h.lmd_payload < 8u || h.lmd_payload > lmd_limit)
        return LZFSE_MI_FSE_BAD_LMD_COUNT;
    if (h.lmd_bits > 7u) return LZFSE_MI_FSE_BAD_LMD_BITS;
    if (h.lmd_state[0] >= L_STATES || h.lmd_state[1] >= M_STATES || h.lmd_state[2] >= D_STATES)
        return LZFSE_MI_FSE_BAD_LMD_STATE;
    if ((h.lit_num & 3u) != 0u || h.lit_num > LITERALS_PER_BLOCK ||
        h.lit_payload > 1024u + (h.lit_num * 10u + 7u) / 8u)
        return LZFSE_MI_FSE_BAD_LITERAL_COUNT;
    if (h.lit_bits > 7u) return LZFSE_MI_FSE_BAD_LITERAL_BITS;
    if (h.lit_state[0] >= U_STATES || h.lit_state[1] >= U_STATES || h.lit_state[2] >= U_STATES ||
        h.lit_state[3] >= U_STATES)
        return LZFSE_MI_FSE_BAD_LMD_PAYLOAD;
    if (h.n_raw > h.lit_num + h.lmd_num * MAX_M_VALUE) return LZFSE_MI_FSE_BAD_RAW_BYTE_COUNT;
    return 0;
}

// fse/block.rs:108-136
// the same from the header's four 8-byte words (the walk fetches them with one round trip)
LZMI_HD inline int fse_parse_v2(uint64_t q0, uint64_t q1, uint64_t q2, uint64_t q3, FseHeader &h) {
    h.n_raw = (uint32_t)(q0 >> 32);
    uint64_t q = q1;
    h.lit_num = (uint32_t)(q & 0xFFFFF);
    h.lit_payload = (uint32_t)((q >> 20) & 0xFFFFF);
    h.lmd_num = (uint32_t)((q >> 40) & 0xFFFFF);
    h.lit_bits = 7u - (uint32_t)((q >> 60) & 7);
    q = q2;
    h.lit_state[0] = (uint32_t)(q & 0x3FF);
    h.lit_state[1] = (uint32_t)((q >> 10) & 0x3FF);
    h.lit_state[2] = (uint32_t)((q >> 20) & 0x3FF);
    h.lit_state[3] = (uint32_t)((q >> 30) & 0x3FF);
    h.lmd_payload = (uint32_t)((q >> 40) & 0xFFFFF);
    h.lmd_bits = 7u - (uint32_t)((q >> 60) & 7);
    q = q3;
    uint32_t header_size = (uint32_t)q;
    h.lmd_state[0] = (uint32_t)((q >> 32) & 0x3FF);
    h.lmd_state[1] = (uint32_t)((q >> 42) & 0x3FF);
    h.lmd_state[2] = (uint32_t)((q >> 52) & 0x3FF);
    h.n_weight = header_size - V2_HEADER_SIZE;  // wrapping_sub
    if (h.n_weight > V2_WEIGHT_PAYLOAD_BYTES_MAX) return LZFSE_MI_FSE_BAD_WEIGHT_PAYLOAD;
    h.hdr_size = V2_HEADER_SIZE + h.n_weight;
    return fse_validate(h);
}

LZMI_HD inline int fse_load_v2(const uint8_t *p, FseHeader &h) {
    h.n_raw = ld_u32(p + 4);
    uint64_t q = ld_u64(p + 8);
    h.lit_num = (uint32_t)(q & 0xFFFFF);
    h.lit_payload = (uint32_t)((q >> 20) & 0xFFFFF);
    h.lmd_num = (uint32_t)((q >> 40) & 0xFFFFF);
    h.lit_bits = 7u - (uint32_t)((q >> 60) & 7);
    q = ld_u64(p + 16);
    h.lit_state[0] = (uint32_t)(q & 0x3FF);
    h.lit_state[1] = (uint32_t)((q >> 10) & 0x3FF);
    h.lit_state[2] = (uint32_t)((q >> 20) & 0x3FF);
    h.lit_state[3] = (uint32_t)((q >> 30) & 0x3FF);
    h.lmd_payload = (uint32_t)((q >> 40) & 0xFFFFF);
    h.lmd_bits = 7u - (uint32_t)((q >> 60) & 7);
    q = ld_u64(p + 24);
    uint32_t header_size = (uint32_t)q;
    h.lmd_state[0] = (uint32_t)((q >> 32) & 0x3FF);
    h.lmd_state[1] = (uint32_t)((q >> 42) & 0x3FF);
    h.lmd_state[2] = (uint32_t)((q >> 52) & 0x3FF);
    h.n_weight = header_size - V2_HEADER_SIZE;  // wrapping_sub
    if (h.n_weight > V2_WEIGHT_PAYLOAD_BYTES_MAX) return LZFSE_MI_FSE_BAD_WEIGHT_PAYLOAD;
    h.hdr_size = V2_HEADER_SIZE + h.n_weight;
    return fse_validate(h);
}

// fse/block.rs:80-104
LZMI_HD inline int fse_load_v1(const uint8_t *p, FseHeader &h) {
    h.n_raw = ld_u32(p + 4);
    uint32_t n_payload = ld_u32(p + 8);
    h.lit_num = ld_u32(p + 12);
    h.lmd_num = ld_u32(p + 16);
    h.lit_payload = ld_u32(p + 20);
    h.lmd_payload = ld_u32(p + 24);
    h.lit_bits = 0u - ld_u32(p + 28);
    for (int i = 0; i < 4; i++) h.lit_state[i] = ld_u16(p + 32 + 2 * i);
    h.lmd_bits = 0u - ld_u32(p + 40);
    for (int i = 0; i < 3; i++) h.lmd_state[i] = ld_u16(p + 44 + 2 * i);
    h.n_weight = V1_WEIGHT_PAYLOAD_BYTES;
    h.hdr_size = V1_HEADER_SIZE + V1_WEIGHT_PAYLOAD_BYTES;
    if (n_payload < h.lit_payload + h.lmd_payload) return LZFSE_MI_FSE_BAD_PAYLOAD_COUNT;
    return fse_validate(h);
}



#if defined(__HIPCC__)
// Wave-wide (64 lanes) inclusive prefix sum by DPP moves: row_shr 1, 2, 4, 8 inside the rows of 16 lanes, then the last lane
// of a row broadcast into the next row (row_bcast:15 for rows 1 and 3, row_bcast:31 for rows 2 and 3). Six dependent vector
// instructions; the same scan by __shfl_up is six dependent trips through the LDS crossbar (ds_bpermute), ten times the
// latency -- and a lone wave or a barrier-bound workgroup pays latency. All 64 lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_take(uint32_t ident, uint32_t v) {   // v of the source lane, `ident` where there is none
    return (uint32_t)__builtin_amdgcn_update_dpp((int)ident, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v) {
    v += dpp_take<0x111, 0xF>(0u, v);
    v += dpp_take<0x112, 0xF>(0u, v);
    v += dpp_take<0x114, 0xF>(0u, v);
    v += dpp_take<0x118, 0xF>(0u, v);
    v += dpp_take<0x142, 0xA>(0u, v);
    v += dpp_take<0x143, 0xC>(0u, v);
    return v;
}
#endif

}  // namespace lzmi
