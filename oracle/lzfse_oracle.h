/*
 * lzfse_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the slice codec of shampoofactory/lzfse_rust v0.2.0:
 * LzfseEncoder::encode_bytes (src/encode/encoder.rs:49) and
 * LzfseDecoder::decode_bytes (src/decode/decoder.rs:61).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library. The product (lzfse_rust_amd/) never links or calls it.
 *
 * Parity pin: the reference cannot be built here (Rust toolchain absent). The oracle is
 * pinned by the reference's own fixtures and known-answer tests:
 *   decode : 12 data/snappy/ *.lzfse + .hash, data/special/compound, data/mutate/ *,
 *            data/synth/ *, and the 12 data/snappy/lmdy_output/ *.lmd LMD streams
 *   encode : the 7 byte-exact vectors of src/encode/frontend_bytes.rs:455-531 and the
 *            doc-test vector of src/encode/encoder.rs:35-47; SURVEY.md App. B.3 hashes
 *            (an independent restatement) as a cross-check.
 * Real-data encoder bytes are NOT pinned by the reference's own tests.
 */
#ifndef LZFSE_ORACLE_H
#define LZFSE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes. Numeric values equal include/lzfse_mi.h (checked by tests/test_abi.py).
 * They map 1:1 on crate::Error (src/error/mod.rs:40-61), FseErrorKind
 * (src/fse/error_kind.rs:9-40) and VnErrorKind (src/vn/error_kind.rs:9-16). */
enum {
    LZO_OK = 0,
    LZO_IO = 1,
    LZO_BAD_BLOCK = 2,
    LZO_BAD_BIT_STREAM = 3,
    LZO_BAD_D_VALUE = 4,
    LZO_BAD_READER_STATE = 5,
    LZO_BUFFER_OVERFLOW = 6, /* dst capacity too small (the Vec would have grown) */
    LZO_PAYLOAD_OVERFLOW = 7,
    LZO_PAYLOAD_UNDERFLOW = 8,
    LZO_UNSUPPORTED = 9, /* inputs > 2^31+2 bytes: reposition path not restated */
    LZO_FSE_BAD_LITERAL_BITS = 16,
    LZO_FSE_BAD_LITERAL_COUNT = 17,
    LZO_FSE_BAD_LITERAL_PAYLOAD = 18,
    LZO_FSE_BAD_LITERAL_STATE = 19,
    LZO_FSE_BAD_LMD_BITS = 20,
    LZO_FSE_BAD_LMD_COUNT = 21,
    LZO_FSE_BAD_LMD_PAYLOAD = 22,
    LZO_FSE_BAD_LMD_STATE = 23,
    LZO_FSE_BAD_PAYLOAD_COUNT = 24,
    LZO_FSE_BAD_RAW_BYTE_COUNT = 25,
    LZO_FSE_BAD_READER_STATE = 26,
    LZO_FSE_BAD_WEIGHT_PAYLOAD = 27,
    LZO_FSE_BAD_WEIGHT_PAYLOAD_COUNT = 28,
    LZO_FSE_WEIGHT_PAYLOAD_OVERFLOW = 29,
    LZO_FSE_WEIGHT_PAYLOAD_UNDERFLOW = 30,
    LZO_VN_BAD_PAYLOAD_COUNT = 48,
    LZO_VN_BAD_PAYLOAD = 49,
    LZO_VN_BAD_OPCODE = 50
};

/* Trace hooks (all optional). */
typedef struct lzo_trace {
    void *ctx;
    /* decode: one call per decoded LMD, D after substitution (lmd_type.rs:153-160). */
    void (*lmd)(void *ctx, uint32_t l, uint32_t m, uint32_t d);
    /* encode: one call per match handed to the backend (frontend_bytes.rs:287-302):
     * literal_index before the push, match idx, match len, distance. Final literals
     * are reported with len = 0, dist = 1. */
    void (*match)(void *ctx, uint32_t literal_index, uint32_t idx, uint32_t len, uint32_t dist);
    /* encode: one call per bvx2 block emitted: n_lmds, n_literals (unpadded), n_raw. */
    void (*block)(void *ctx, uint32_t n_lmds, uint32_t n_literals, uint32_t n_raw);
    /* encode: every LmdPack pushed into the fse Buffer (D zeroed as stored). */
    void (*pack)(void *ctx, uint32_t l, uint32_t m, uint32_t d_zeroed);
} lzo_trace;

size_t lzo_encode_bound(size_t n);

/* encode_bytes: appends the stream for src[0..n) at dst, *out_len = bytes written. */
int lzo_encode(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_len,
               const lzo_trace *trace);
/* the same with another BLOCK_GUIDE / SLACK (frontend_bytes.rs:19-23): lets the tests see the front end reposition
 * (:348-375) on inputs of a few MiB. lzo_encode == lzo_encode_guide(0x7FFF_FFFF, 0x1000_0000), inputs of any length. */
int lzo_encode_guide(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_len, uint32_t guide, uint32_t slack);

/* decode_bytes: decodes the whole stream src[0..n) (must end exactly at bvx$ + 4). */
int lzo_decode(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_len,
               const lzo_trace *trace);

/* Sum of n_raw_bytes over block headers (decode/probe.rs:11-35; bvxn uses the payload
 * length for skipping). */
int lzo_decode_size(const uint8_t *src, size_t n, uint64_t *raw_len);

/* Per-position candidate dump for the first stage of the encoder: for every position
 * i in [0, n-4], the result of find_match BEFORE backward extension
 * (frontend_bytes.rs:214-231) as if position i were visited: match_idx[i] (0xFFFFFFFF
 * if none) and fwd_len[i]. Valid because every position is inserted exactly once in
 * order (frontend_bytes.rs:187,336-344). n must be > 4096 (Fse backend type). */
int lzo_candidates(const uint8_t *src, size_t n, uint32_t *match_idx, uint32_t *fwd_len);
/* candidate queue (4 positions, newest first, 0xFFFFFFFF = empty) of every position 0 .. n-4 */
int lzo_table_rows(const uint8_t *src, size_t n, uint32_t *rows);

/* ---- ring / stream encoder: LzfseRingEncoder::encode, LzfseWriter, LzfseWriterBytes (encode/ring_encoder.rs:55-97,
 * encode/writer.rs:39-75, encode/frontend_ring.rs). Its bytes differ from lzo_encode's (another parse). A handle is a
 * fresh encoder; ring_size = 0 takes the reference's Input ring (512 KiB, 16 KiB blocks, encode/constants.rs:23-33). */
typedef struct lzo_ring lzo_ring;
lzo_ring *lzo_ring_new(uint32_t ring_size, uint32_t ring_blk, uint32_t ring_limit, const lzo_trace *trace);
int lzo_ring_write(lzo_ring *r, const uint8_t *src, size_t len);                  /* Write::write */
int lzo_ring_finish(lzo_ring *r, const uint8_t **out, size_t *out_len);           /* finalize(); *out is owned by r */
void lzo_ring_free(lzo_ring *r);
int lzo_ring_encode(const uint8_t *src, size_t n, size_t piece, uint8_t *dst, size_t cap, size_t *out_len,
                    const lzo_trace *trace);
/* the hand-made states of the reference's in-file KATs (frontend_ring.rs:861-992), Dummy backend */
int lzo_ring_kat(int mode, uint32_t ring_size, uint32_t ring_blk, uint32_t ring_limit, const uint8_t *ring_data,
                 size_t n_data, uint32_t idx0, uint32_t n, const lzo_trace *trace);

/* Low-level restatements exported for known-answer tests. */
void lzo_normalize_m1(uint16_t *weights, uint32_t n_weights, uint32_t in_total, uint32_t out_total);
uint32_t lzo_weights_store_v2(const uint16_t *weights360, uint8_t *dst630);
int lzo_weights_load_v2(const uint8_t *src, uint32_t n, uint16_t *weights360);

#ifdef __cplusplus
}
#endif
#endif
