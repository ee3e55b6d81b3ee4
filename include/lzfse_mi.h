/*
 * lzfse_mi.h -- C ABI of the MI355X-native LZFSE block codec.
 *
 * Drop-in boundary for lzfse_rust's slice path (citations relative to the reference tree):
 *   LzfseEncoder::encode_bytes(&mut self, src:&[u8], dst:&mut Vec<u8>) -> io::Result<u64>
 *       src/encode/encoder.rs:49-53  -> lzfse_mi_encode
 *   LzfseDecoder::decode_bytes(&mut self, src:&[u8], dst:&mut Vec<u8>) -> crate::Result<u64>
 *       src/decode/decoder.rs:61-69  -> lzfse_mi_decode
 *   decode::probe (sum of n_raw_bytes) src/decode/probe.rs:11-35 -> lzfse_mi_decode_size
 * Shape follows the in-tree FFI precedent lzfse_sys (lzfse_sys/src/lib.rs:29-56): plain
 * pointers and sizes, caller-owned buffers, no exceptions, an int status per call.
 * The Rust-side binding a maintainer would add is shown in INTEGRATION.md.
 *
 * All compute runs in hand-written HIP kernels for gfx950. There is NO CPU fallback:
 * lzfse_mi_create fails with LZFSE_MI_NO_DEVICE when no HIP device is usable.
 */
#ifndef LZFSE_MI_H
#define LZFSE_MI_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__) || defined(__clang__)
#define LZFSE_MI_API __attribute__((visibility("default")))   /* the library is built with -fvisibility=hidden */
#else
#define LZFSE_MI_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes: 1:1 with crate::Error (src/error/mod.rs:40-61), FseErrorKind
 * (src/fse/error_kind.rs:9-40) and VnErrorKind (src/vn/error_kind.rs:9-16). */
enum {
    LZFSE_MI_OK = 0,
    LZFSE_MI_IO = 1,                 /* Error::Io: HIP runtime / allocation failure */
    LZFSE_MI_BAD_BLOCK = 2,          /* Error::BadBlock(magic) */
    LZFSE_MI_BAD_BIT_STREAM = 3,     /* Error::BadBitStream */
    LZFSE_MI_BAD_D_VALUE = 4,        /* Error::BadDValue */
    LZFSE_MI_BAD_READER_STATE = 5,   /* Error::BadReaderState (unused on the slice path) */
    LZFSE_MI_BUFFER_OVERFLOW = 6,    /* Error::BufferOverflow: dst capacity too small */
    LZFSE_MI_PAYLOAD_OVERFLOW = 7,   /* Error::PayloadOverflow */
    LZFSE_MI_PAYLOAD_UNDERFLOW = 8,  /* Error::PayloadUnderflow */
    LZFSE_MI_UNSUPPORTED = 9,        /* inputs the device path does not take */
    LZFSE_MI_NO_DEVICE = 10,         /* no usable HIP device: the product never falls back */
    LZFSE_MI_BAD_ARGUMENT = 11,
    LZFSE_MI_FSE_BAD_LITERAL_BITS = 16,
    LZFSE_MI_FSE_BAD_LITERAL_COUNT = 17,
    LZFSE_MI_FSE_BAD_LITERAL_PAYLOAD = 18,
    LZFSE_MI_FSE_BAD_LITERAL_STATE = 19,
    LZFSE_MI_FSE_BAD_LMD_BITS = 20,
    LZFSE_MI_FSE_BAD_LMD_COUNT = 21,
    LZFSE_MI_FSE_BAD_LMD_PAYLOAD = 22,
    LZFSE_MI_FSE_BAD_LMD_STATE = 23,
    LZFSE_MI_FSE_BAD_PAYLOAD_COUNT = 24,
    LZFSE_MI_FSE_BAD_RAW_BYTE_COUNT = 25,
    LZFSE_MI_FSE_BAD_READER_STATE = 26,
    LZFSE_MI_FSE_BAD_WEIGHT_PAYLOAD = 27,
    LZFSE_MI_FSE_BAD_WEIGHT_PAYLOAD_COUNT = 28,
    LZFSE_MI_FSE_WEIGHT_PAYLOAD_OVERFLOW = 29,
    LZFSE_MI_FSE_WEIGHT_PAYLOAD_UNDERFLOW = 30,
    LZFSE_MI_VN_BAD_PAYLOAD_COUNT = 48,
    LZFSE_MI_VN_BAD_PAYLOAD = 49,
    LZFSE_MI_VN_BAD_OPCODE = 50
};

/* One context = one HIP device + one HIP stream + reusable device scratch. Like
 * LzfseEncoder/LzfseDecoder (`&mut self`, encoder.rs:14-18, decoder.rs:17-21) a context is
 * not thread-safe; distinct contexts are independent. Results never depend on prior calls. */
typedef struct lzfse_mi_ctx lzfse_mi_ctx;

/* HIP devices visible to the process (0: none, and lzfse_mi_create fails with LZFSE_MI_NO_DEVICE): what a caller that wants one
 * context per device (lzfse_mi_encode_chunked / _decode_chunked) iterates over. Initialises nothing on any device. */
LZFSE_MI_API int lzfse_mi_device_count(void);
LZFSE_MI_API int lzfse_mi_create(int device, lzfse_mi_ctx **out);
LZFSE_MI_API void lzfse_mi_destroy(lzfse_mi_ctx *ctx);
LZFSE_MI_API const char *lzfse_mi_status_string(int status);
LZFSE_MI_API const char *lzfse_mi_version(void);

/* Use an existing HIP stream (hipStream_t passed as void*, e.g. torch's current stream).
 * NULL restores the context's own stream. */
LZFSE_MI_API int lzfse_mi_set_stream(lzfse_mi_ctx *ctx, void *hip_stream);

/* Tuning knobs. A large batch call is cut into sub-batches ("lanes") that run side by side on their own HIP streams
 * (several stages are latency-bound); results never depend on these. The LZFSE_MI_OPT_DIAG_* options exist in the
 * diagnostic build of the library only (liblzfse_mi_diag.so, used by the test-suite to force code paths); the product
 * library answers LZFSE_MI_UNSUPPORTED and reads no environment variable at all. */
enum {
    LZFSE_MI_OPT_ENCODE_LANES = 1,  /* 0: chosen by batch size (default), 1: one pass ("exclusive" kernel timing), 2..4 */
    LZFSE_MI_OPT_DECODE_LANES = 2,
    LZFSE_MI_OPT_STAGGER = 3,       /* 1: encode lanes start one after the other (default 0: together) */
    LZFSE_MI_OPT_DECODE_PIPE = 4,   /* several workgroups per stream in the LZ stage of decode: 0 by the batch's shape (default),
                                       1 never, else K | variant << 8: K workgroups (2..64) for every stream of the tile
                                       kernel, variant 0 = 256 threads / 8 KiB tiles, 1 = 1024 threads / 32 KiB tiles */
    LZFSE_MI_OPT_STREAM_SPARE = 5,  /* 1 (default): the window buffers of a destroyed stream object (lzfse_mi_dstream / _estream)
                                       stay with the context for the next one -- pinned host memory, about 4.5 windows for a decoder
                                       (its input, the copy of the window in flight, two windows of output and one for a window
                                       decoded in the call) and 4.75 for an encoder (three of input, 1.75 of output): ~300 MiB each
                                       at the default window, until lzfse_mi_destroy; 0: free what is held now and keep nothing
                                       from now on */
    LZFSE_MI_OPT_DIAG_LZ_PATH = 100, /* -1: by cost, 0: tile kernel only, 1: pointer jumping for every stream */
    LZFSE_MI_OPT_DIAG_LZ_TILE = 101, /* -1: by stream count, 0: 256-thread / 8 KiB tile, 1: 1024-thread / 32 KiB tile */
    LZFSE_MI_OPT_DIAG_STATS = 102,   /* bit mask: per-stage statistics on stderr */
    LZFSE_MI_OPT_DIAG_CHAIN = 103,   /* bit 0: every chain tile through the ballot kernel (the fallback of the LDS-exchange one); bits 4-6: chain tiles of 1, 2 or 4 x 65 472 positions (0x10, 0x20, 0x40) instead of the call's own choice */
    LZFSE_MI_OPT_DIAG_WALK = 104,    /* decode header walk: 0 by size, 1: every stream tries the parallel walk first, 2: serial only */
    LZFSE_MI_OPT_DIAG_PIPE_SCATTER = 105, /* 1: the pipelined LZ kernel is told that the workgroups of a stream sit on different XCDs
                                             (it must refuse, and the streams are decoded again by the one-workgroup kernel) */
    LZFSE_MI_OPT_DIAG_GUIDE = 106    /* guide | slack << 32: BLOCK_GUIDE and SLACK of the slice front end (frontend_bytes.rs:19-23) for
                                        lzfse_mi_encode, so that tests see it reposition (:348-375) on inputs of a few MiB; 0: the
                                        reference's 0x7FFF_FFFF and 0x1000_0000 */
};
LZFSE_MI_API int lzfse_mi_set_option(lzfse_mi_ctx *ctx, int option, int64_t value);

/* Read-only facts about a context. LZFSE_MI_INFO_PIPE_REFUSALS: how often the pipelined LZ stage of decode (several
 * workgroups per stream handing output over through one XCD's L2) was given up on this context and its helpers -- its
 * start-up self-test failed, or a launch found a stream's workgroups on different XCDs -- after which the context decodes
 * such batches with one workgroup per stream (correct, slower). 0 on the hardware this was written on. */
enum { LZFSE_MI_INFO_PIPE_REFUSALS = 1 };
LZFSE_MI_API int lzfse_mi_get_info(lzfse_mi_ctx *ctx, int what, int64_t *value);

/* Upper bound of the encoded size of an n-byte input (fse/constants.rs:54-69: every full
 * bvx2 block carries >= 39 996 raw bytes and costs <= 54 bits per LMD + 10 bits per literal). */
LZFSE_MI_API size_t lzfse_mi_encode_bound(size_t n);

/* ---- host-pointer entry points: exactly what the Rust shim binds ---------------------- */

/* encode_bytes: writes the complete stream (blocks + bvx$) for src[0..n) at dst, never more
 * than cap bytes; *out_len = bytes written (the u64 the Rust method returns). Any n: a slice of more than
 * 0x8000_0002 bytes is matched in several blocks, as the reference's front end does it (frontend_bytes.rs:160-211,
 * 348-375: match_any / reposition; its test/src/big_mem.rs encodes 0x8000_0003 .. 0x2_0000_0000 bytes), one device
 * call per block; size cap with lzfse_mi_encode_bound(n). (The batch entry points take streams of up to 0x8000_0002
 * bytes -- one block of the front end -- and answer LZFSE_MI_UNSUPPORTED for a longer one.) */
LZFSE_MI_API int lzfse_mi_encode(lzfse_mi_ctx *ctx, const uint8_t *src, size_t n, uint8_t *dst, size_t cap,
                    size_t *out_len);

/* decode_bytes: src[0..n) must be one complete stream ending exactly at bvx$ + 4 bytes
 * (decoder.rs:90-96). *out_len = raw bytes written. LZFSE_MI_BUFFER_OVERFLOW if cap is short;
 * size dst with lzfse_mi_decode_size. */
LZFSE_MI_API int lzfse_mi_decode(lzfse_mi_ctx *ctx, const uint8_t *src, size_t n, uint8_t *dst, size_t cap,
                    size_t *out_len);

/* The size classes the reference keeps on the host CPU (encode/frontend_bytes.rs:63-111): n <= 20 -> one bvx-
 * block, 21..=4096 -> one bvxn block (or bvx- when not smaller), then bvx$. Pure host code, no context needed;
 * lzfse_mi_encode / _batch / _batch_device route inputs of this size class here themselves. n > 4096 is
 * LZFSE_MI_BAD_ARGUMENT (those inputs are bvx2 and belong to the device path). */
LZFSE_MI_API int lzfse_mi_encode_small(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_len);

/* Header walk on the host: sum of n_raw_bytes of all blocks. */
LZFSE_MI_API int lzfse_mi_decode_size(const uint8_t *src, size_t n, uint64_t *raw_len);

/* Many independent streams per call (the unit of GPU parallelism). The
 * return value reports call-level failures only; statuses[i] is per stream.
 * dsts[i] may be memory that has never been touched (a Vec<u8> just made): only dsts[i][0 .. out_lens[i]) is written, but
 * pages of dsts[i][0 .. caps[i]) may be mapped by the call (madvise MADV_POPULATE_WRITE while the kernels run). */
LZFSE_MI_API int lzfse_mi_encode_batch(lzfse_mi_ctx *ctx, size_t count, const uint8_t *const *srcs,
                          const size_t *lens, uint8_t *const *dsts, const size_t *caps,
                          size_t *out_lens, int *statuses);
LZFSE_MI_API int lzfse_mi_decode_batch(lzfse_mi_ctx *ctx, size_t count, const uint8_t *const *srcs,
                          const size_t *lens, uint8_t *const *dsts, const size_t *caps,
                          size_t *out_lens, int *statuses);

/* The u32 some error kinds carry (src/error/mod.rs:47 Error::BadBlock(magic); src/fse/error_kind.rs:12-21
 * FseErrorKind::BadLmdCount(n_lmds), BadLiteralCount(n_literals)): value for stream `stream_index` of the LAST batch
 * call on this context (index 0 for the single-stream calls); 0 when the stream's status carries none.
 * VnErrorKind::BadPayloadCount(u32) (src/vn/error_kind.rs:11) is raised by the reference's encoder-side constructor
 * only (vn/block.rs:16-22) and cannot occur on this path. */
LZFSE_MI_API int lzfse_mi_last_error_detail(lzfse_mi_ctx *ctx, size_t stream_index, uint32_t *detail);

/* ---- device-resident entry points (inputs and outputs already in HBM) ----------------- */
/* Stream i reads d_src[src_off[i] .. src_off[i] + src_len[i]) and writes at
 * d_dst[dst_off[i] ..], at most dst_cap[i] bytes. Offset/length arrays are HOST arrays.
 * out_lens / statuses are HOST arrays filled when the call returns (the call synchronises
 * the context's stream once at its end). */
LZFSE_MI_API int lzfse_mi_decode_batch_device(lzfse_mi_ctx *ctx, size_t count, const void *d_src,
                                 const uint64_t *src_off, const uint64_t *src_len, void *d_dst,
                                 const uint64_t *dst_off, const uint64_t *dst_cap,
                                 uint64_t *out_lens, int *statuses);
LZFSE_MI_API int lzfse_mi_encode_batch_device(lzfse_mi_ctx *ctx, size_t count, const void *d_src,
                                 const uint64_t *src_off, const uint64_t *src_len, void *d_dst,
                                 const uint64_t *dst_off, const uint64_t *dst_cap,
                                 uint64_t *out_lens, int *statuses);

/* ---- chunked container, one process driving several devices ------------- */
/* A large input is cut into `chunk`-byte pieces (0 = LZFSE_MI_CHUNK_DEFAULT), each encoded as its own complete LZFSE
 * stream; chunk c is handled by ctxs[c mod n_ctx] (one context per device: the chunks are independent, nothing is
 * exchanged between devices). The frame is this build's own (magic "LZMC", chunk table, streams back to back): a plain
 * LzfseDecoder decodes each chunk stream, not the frame. What the lzfoo-like CLI (python -m lzfse_rust_amd.cli) writes. */
#define LZFSE_MI_CHUNK_DEFAULT ((size_t)4 << 20)
LZFSE_MI_API size_t lzfse_mi_chunked_bound(size_t n, size_t chunk);
LZFSE_MI_API int lzfse_mi_encode_chunked(lzfse_mi_ctx *const *ctxs, int n_ctx, const uint8_t *src, size_t n, size_t chunk,
                                         uint8_t *dst, size_t cap, size_t *out_len);
LZFSE_MI_API int lzfse_mi_decode_chunked_size(const uint8_t *src, size_t n, uint64_t *raw_len);
LZFSE_MI_API int lzfse_mi_decode_chunked(lzfse_mi_ctx *const *ctxs, int n_ctx, const uint8_t *src, size_t n, uint8_t *dst,
                                         size_t cap, size_t *out_len);

/* ---- measurement ---------------------------------------------------------------------- */
/* Per-kernel device time of the LAST batch call on this context, measured with HIP events
 * recorded on the stream the kernels were launched on. names[i] points at static strings. */
#define LZFSE_MI_MAX_STAGES 24
typedef struct lzfse_mi_timings {
    int n_stages;
    const char *names[LZFSE_MI_MAX_STAGES];
    float ms[LZFSE_MI_MAX_STAGES];
    uint64_t launches[LZFSE_MI_MAX_STAGES];
} lzfse_mi_timings;
LZFSE_MI_API int lzfse_mi_enable_timing(lzfse_mi_ctx *ctx, int enable);
LZFSE_MI_API int lzfse_mi_get_timings(lzfse_mi_ctx *ctx, lzfse_mi_timings *out);

/* The reference decodes into a Vec (decode/decoder.rs:52-57), so a damaged block that produces more than its header's
 * n_raw_bytes runs to its last LMD and fails there (BadLmdPayload, fse/fse_core.rs:132-140) where a destination of exactly
 * lzfse_mi_decode_size bytes reports LZFSE_MI_BUFFER_OVERFLOW first. A caller that wants the reference's status for such a
 * stream decodes again with this many more bytes of capacity (an upper bound of what the stream's blocks can over-produce). */
LZFSE_MI_API size_t lzfse_mi_decode_headroom(const uint8_t *src, size_t n);

/* ---- Streaming decode (decode half): LzfseRingDecoder::decode(reader, writer),
 * decode/ring_decoder.rs:58-68. Input arrives in pieces (lzfse_mi_dstream_feed), output leaves through `write` in
 * pieces; the result is the slice path's: the same bytes, and for a damaged stream the same status at the feed call
 * that completes the evidence (sticky afterwards). finish != 0 marks the end of the input: a stream that does not end
 * with bvx$ in its last 4 bytes is an error (decode/decoder.rs:93-95). `window` (0 = LZFSE_MI_STREAM_WINDOW) is the
 * number of raw bytes decoded per device call: a window's blocks are decoded when they are all there (or the input
 * ends), so output follows input by up to a window. A full window with more input to come is decoded in the BACKGROUND (a
 * helper thread of the object, a context of the library's own beside `ctx`): its bytes reach `write` -- on the caller's
 * thread, in stream order -- in the feed call that sends the NEXT window off, or in the one that ends the input, so output
 * follows input by up to two windows, and an error a window met is returned by that call (the windows before it have been
 * written by then). Memory (pinned host memory): the input not yet decoded, a copy of the window in flight, and two windows
 * of output. `write` returns 0 to go on. */
/* (a window is ONE stream on the device, and one stream costs its latency floors -- a block's entropy chain, the header
 * walk -- whatever its size: through the Python mirror 64 MiB windows decode 256 MiB of text at 13.5 GB/s, 16 MiB windows at 8.5,
 * 4 MiB windows at 2.9; the slice call into a reused buffer reaches 15: profiles/r04_stream_bench.txt) */
#define LZFSE_MI_STREAM_WINDOW ((size_t)64 << 20)
typedef struct lzfse_mi_dstream lzfse_mi_dstream;
typedef int (*lzfse_mi_write_fn)(void *user, const uint8_t *bytes, size_t n);
LZFSE_MI_API int lzfse_mi_dstream_create(lzfse_mi_ctx *ctx, size_t window, lzfse_mi_dstream **out);
LZFSE_MI_API int lzfse_mi_dstream_feed(lzfse_mi_dstream *s, const uint8_t *src, size_t n, int finish, lzfse_mi_write_fn write,
                                       void *user);
/* feed without the copy, for LzfseRingDecoder::decode(reader, writer) (decode/ring_decoder.rs:57-67 reads straight into its
 * ring): reserve says where the next `want` input bytes go, the caller reads into *ptr and commits what came (n <= want),
 * which then does what feed does behind its copy. feed(src, n, ..) is reserve + memcpy + commit. */
LZFSE_MI_API int lzfse_mi_dstream_reserve(lzfse_mi_dstream *s, size_t want, uint8_t **ptr);
LZFSE_MI_API int lzfse_mi_dstream_commit(lzfse_mi_dstream *s, size_t n, int finish, lzfse_mi_write_fn write, void *user);
/* bytes of input consumed / of output written so far: the (u, v) LzfseRingDecoder::decode returns */
LZFSE_MI_API int lzfse_mi_dstream_totals(const lzfse_mi_dstream *s, uint64_t *bytes_in, uint64_t *bytes_out);
/* A stream object (decoder or encoder) and its context may be destroyed in either order: lzfse_mi_destroy(ctx) detaches the
 * stream objects still alive, whose feed / finish then return LZFSE_MI_BAD_ARGUMENT and whose destroy frees only their own. */
LZFSE_MI_API void lzfse_mi_dstream_destroy(lzfse_mi_dstream *s);

/* ---- Ring / stream encode (encode half): LzfseRingEncoder::encode(reader, writer)
 * (encode/ring_encoder.rs:55-67), LzfseRingEncoder::writer / writer_bytes -> LzfseWriter / LzfseWriterBytes
 * (ring_encoder.rs:79-97, encode/writer.rs:39-75, encode/writer_bytes.rs:44-78). The reference's ring front end
 * (encode/frontend_ring.rs) is another parse than the slice encoder's -- a 512 KiB ring matched in rounds, forward
 * lengths capped and measured coarsely, literals that pass the ring's head pushed as they are -- so its streams are
 * other BYTES than lzfse_mi_encode's for the same input (both decode to it). These entry points produce the ring
 * encoder's bytes (those of a fresh LzfseRingEncoder: where the reference's compares run past the end of the input a
 * reused encoder would see its previous stream's bytes in the ring), whatever the sizes of the pieces fed. */
LZFSE_MI_API int lzfse_mi_encode_ring(lzfse_mi_ctx *ctx, const uint8_t *src, size_t n, uint8_t *dst, size_t cap,
                                      size_t *out_len);
LZFSE_MI_API int lzfse_mi_encode_ring_batch(lzfse_mi_ctx *ctx, size_t count, const uint8_t *const *srcs,
                                            const size_t *lens, uint8_t *const *dsts, const size_t *caps,
                                            size_t *out_lens, int *statuses);
LZFSE_MI_API int lzfse_mi_encode_ring_batch_device(lzfse_mi_ctx *ctx, size_t count, const void *d_src,
                                                   const uint64_t *src_off, const uint64_t *src_len, void *d_dst,
                                                   const uint64_t *dst_off, const uint64_t *dst_cap,
                                                   uint64_t *out_lens, int *statuses);
/* LzfseWriter: feed = Write::write (any piece sizes), finish = finalize(); the stream leaves through `write` (which
 * returns 0 to go on; anything else ends the call with LZFSE_MI_IO), *bytes_in / *bytes_out are the (u64, u64)
 * LzfseRingEncoder::encode returns. The input is encoded a WINDOW at a time (`window` new bytes per device call, 0 =
 * LZFSE_MI_STREAM_WINDOW, at least 1 MiB): the reference's ring encoder decides everything about a position from the 256 KiB
 * behind and ahead of it, so the blocks of a window that end more than that before its last byte are final -- they leave
 * through `write` during feed, and the object keeps only what the parse may still reach back to (328 KiB) plus the input it
 * has not finished with. A window that is not the stream's last is encoded in the BACKGROUND (a helper thread of the object,
 * a context of the library's own beside `ctx`), while feed returns and takes the next window's input: its blocks leave
 * through `write` in a later feed, or in finish -- always on the caller's thread, in stream order -- and an error it met is
 * returned by that call. Memory (pinned host memory): up to three windows of input and one window's output, whatever the
 * length of the stream (inputs that are a few matches of many MiB each can take longer to yield a final block; the window
 * grows then). The bytes are those of lzfse_mi_encode_ring on the whole input, whatever the window and the pieces. Used once. */
typedef struct lzfse_mi_estream lzfse_mi_estream;
LZFSE_MI_API int lzfse_mi_estream_create(lzfse_mi_ctx *ctx, size_t window, lzfse_mi_estream **out);
LZFSE_MI_API int lzfse_mi_estream_feed(lzfse_mi_estream *s, const uint8_t *src, size_t n, lzfse_mi_write_fn write, void *user);
/* feed without the copy, for LzfseRingEncoder::encode(reader, writer) (encode/ring_encoder.rs:55-67: its copy(reader) reads
 * straight into the ring): reserve says where the next input bytes go and how many fit (1 <= *room <= want; making room may
 * hand a finished window's blocks to `write` and send the next window off), the caller reads into *ptr and commits what came
 * (n <= *room; 0 is allowed). feed(src, n) is reserve + memcpy + commit until n bytes are in. */
LZFSE_MI_API int lzfse_mi_estream_reserve(lzfse_mi_estream *s, size_t want, uint8_t **ptr, size_t *room, lzfse_mi_write_fn write,
                                          void *user);
LZFSE_MI_API int lzfse_mi_estream_commit(lzfse_mi_estream *s, size_t n);
LZFSE_MI_API int lzfse_mi_estream_finish(lzfse_mi_estream *s, lzfse_mi_write_fn write, void *user, uint64_t *bytes_in,
                                         uint64_t *bytes_out);
LZFSE_MI_API void lzfse_mi_estream_destroy(lzfse_mi_estream *s);

#ifdef __cplusplus
}
#endif
#endif
