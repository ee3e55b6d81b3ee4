// C ABI of the MI355X LZFSE codec (include/lzfse_mi.h): context, device scratch, batching.
// Host side of LzfseEncoder::encode_bytes (encode/encoder.rs:49-53) and
// LzfseDecoder::decode_bytes (decode/decoder.rs:61-99). No CPU codec lives here: every
// byte of bvx2 entropy coding, match finding and LZ copy is produced by the HIP kernels
// in decode.hip / encode.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <sys/mman.h>
#include <thread>

#include "common.h"
#include "internal.h"
#include <atomic>

using namespace lzmi;

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    // grow-only device buffer; contents are NOT preserved
    bool ensure(size_t n) {
        if (n <= cap) return true;
        if (p) (void)hipFree(p);
        p = nullptr;
        size_t nc = n + n / 4 + 4096;
        if (hipMalloc(&p, nc) != hipSuccess) { cap = 0; p = nullptr; return false; }
        cap = nc;
        return true;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct HostBuf {  // pinned staging
    void *p = nullptr;
    size_t cap = 0;
    bool ensure(size_t n) {
        if (n <= cap) return true;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        size_t nc = n + n / 4 + 4096;
        if (hipHostMalloc(&p, nc, hipHostMallocDefault) != hipSuccess) { cap = 0; p = nullptr; return false; }
        cap = nc;
        return true;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

}  // namespace

#define LZFSE_MI_MAX_LANES 4


struct lzfse_mi_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // batch descriptors / results
    DevBuf d_streams, d_walk, d_plan, d_blocks, d_bres, d_sres;
    // decode scratch
    DevBuf d_lmds, d_lits, d_origin, d_jerr, d_wcache, d_fwalk, d_ck, d_lzp;
    // encode scratch (encode.hip)
    EncScratch enc;
    // host-pointer API staging
    DevBuf d_in, d_out, d_small;
    HostBuf h_in, h_out, h_small;
    std::vector<hipEvent_t> host_ev;   // host_batch: one per output group in flight
    std::vector<LaneWorker *> copy_workers;   // host_batch: helper threads of the staging copies
    lzfse_mi_ctx *host_peer = nullptr;        // host_batch: the context of the second half of a large call ...
    LaneWorker *host_worker = nullptr;        // ... and the thread that drives it
    bool is_peer = false;
    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    struct Span { const char *name; hipEvent_t a, b; };
    std::vector<Span> spans;
    lzfse_mi_timings last{};
    // second lane of a split batch call: own stream, scratch and timers (created on first use)
    lzfse_mi_ctx *shadow[LZFSE_MI_MAX_LANES - 1] = {};
    LaneWorker *worker[LZFSE_MI_MAX_LANES - 1] = {};         // helper thread of shadow[k] (created with it)
    lzmi::LaneGate gates[LZFSE_MI_MAX_LANES - 1];            // owned by the main context
    lzmi::LaneGate *gate_in = nullptr, *gate_out = nullptr;  // set per lane for the duration of a split encode
    hipEvent_t split_ev = nullptr;
    int last_split = 0;  // helper lanes used by the last batch call
    // u32 payload of the error kinds that carry one (Error::BadBlock(magic), FseErrorKind::BadLmdCount(n) /
    // BadLiteralCount(n)): per stream of the last sub-batch run on this context / of the last API call
    std::vector<uint32_t> detail, detail_out;
    // lzfse_mi_set_option
    int opt_lanes_enc = 0, opt_lanes_dec = 0;  // sub-batches run side by side (0: chosen by size, 1: one)
    int opt_stagger = 0;
    int opt_pipe = 0;      // LZFSE_MI_OPT_DECODE_PIPE
    bool pipe_broken = false;   // a launch found the workgroups of one stream on different XCDs: never again on this context
    bool pipe_tested = false;   // the hand-over self-test (dec_lzp_selftest_kernel) has run on this context
    uint32_t pipe_refusals = 0; // times the pipelined LZ kernel was given up (self-test failed / a launch refused): lzfse_mi_get_info
    int lane_share = 1;    // sub-batches running side by side with this one (split_batch)
    lzmi::StreamSpare spare;             // buffers a finished stream object leaves for the next one (stream.hip)
    std::vector<lzfse_mi_ctx **> stream_refs;   // the `ctx` members of the stream objects alive on this context (ctx_attach)
    std::mutex stream_refs_m;
    lzmi::EncWindow *window = nullptr;   // set for the duration of one window of a stream encode (stream.hip)
    bool pinned_out = false;             // ... of one window of either stream object: the destination is pinned memory
    // ... of the stream decoder, in the background: the window's output is left to travel on a stream of its own while the call
    // returns (all but its last 262 139 bytes, which the next window needs at once): two device output buffers in turn, an event each
    bool defer_out = false;
    DevBuf d_out2;
    int out_flip = 0;
    hipStream_t xfer_stream = nullptr;
    hipEvent_t defer_ev[2] = {nullptr, nullptr};
    bool defer_pending[2] = {false, false};
    hipEvent_t defer_last = nullptr;     // the event of the last call's deferred transfer (null: there was none)
    std::shared_ptr<lzmi::StreamBox> stream_box;   // where stream objects run windows in the background (internal.h)
    bool parse_ring = false;   // set for the duration of a ring / stream encode call (lzfse_mi_encode_ring*, lzfse_mi_estream_*)
    int diag_lz_jump = -1, diag_lz_variant = -1, diag_stats = 0, diag_chain = 0, diag_walk = 0, diag_pipe_scatter = 0;  // diagnostic build only
    uint64_t diag_last_lmds = 0;   // LMD records the entropy stage of the last decode pass on this context left in d_lmds (stage hook)

    lzmi::PinVec h_ctl[8];   // control arrays of a decode call (DH_*): pinned, so that their transfers are asynchronous and cheap
    lzmi::RepoWindow *repo = nullptr;   // set by encode_slice_blocks for the device call over one block of a long slice
    uint64_t diag_guide = 0;            // diagnostic build: guide | slack << 32 of the slice front end (LZFSE_MI_OPT_DIAG_GUIDE)
    lzmi::OutMirror mirror;  // set by the host-pointer decode call for the device call it makes: the LZ stage also writes the pinned image
    lzmi::PinVec h_done;     // ... and its word per stream

    hipEvent_t get_event() {
        if (ev_used == ev_pool.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            ev_pool.push_back(e);
        }
        return ev_pool[ev_used++];
    }
};

namespace lzmi {
// RAII helper: brackets a kernel (or group) with events when timing is enabled
StageTimer::StageTimer(lzfse_mi_ctx *c, const char *name) : ctx(c), stage(name) {
    if (!ctx->timing) return;
    hipEvent_t a = ctx->get_event(), b = ctx->get_event();
    if (!a || !b) return;
    (void)hipEventRecord(a, ctx->stream);
    ctx->spans.push_back({name, a, b});
    idx = (int)ctx->spans.size() - 1;
}
StageTimer::~StageTimer() {
    if (idx >= 0) (void)hipEventRecord(ctx->spans[idx].b, ctx->stream);
#ifdef LZFSE_MI_DIAG
    if (ctx->diag_stats & 8) {   // (LZFSE_MI_OPT_DIAG_STATS & 8: every stage waited for and named -- which one faults)
        const hipError_t e = hipStreamSynchronize(ctx->stream);
        fprintf(stderr, "stage %s: %s\n", stage, e == hipSuccess ? "ok" : hipGetErrorString(e));
    }
#endif
}
hipStream_t ctx_stream(lzfse_mi_ctx *c) { return c->stream; }
EncScratch &ctx_enc(lzfse_mi_ctx *c) { return c->enc; }
LaneGate *ctx_gate_in(lzfse_mi_ctx *c) { return c->gate_in; }
LaneGate *ctx_gate_out(lzfse_mi_ctx *c) { return c->gate_out; }
int ctx_diag_stats(lzfse_mi_ctx *c) { return c->diag_stats; }
bool ctx_parse_ring(lzfse_mi_ctx *c) { return c->parse_ring; }
EncWindow *ctx_window(lzfse_mi_ctx *c) { return c->window; }
RepoWindow *ctx_repo(lzfse_mi_ctx *c) { return c->repo; }
void ctx_diag_guide(lzfse_mi_ctx *c, uint32_t &guide, uint32_t &slack) {
    guide = 0x7FFFFFFFu; slack = 0x10000000u;   // BLOCK_GUIDE, SLACK (frontend_bytes.rs:19-23)
    if (c->diag_guide) { guide = (uint32_t)c->diag_guide; slack = (uint32_t)(c->diag_guide >> 32); }
}
StreamSpare &ctx_spare(lzfse_mi_ctx *c) { return c->spare; }
void ctx_attach(lzfse_mi_ctx *c, lzfse_mi_ctx **ref) {
    std::lock_guard<std::mutex> g(c->stream_refs_m);
    c->stream_refs.push_back(ref);
}
void ctx_detach(lzfse_mi_ctx *c, lzfse_mi_ctx **ref) {
    std::lock_guard<std::mutex> g(c->stream_refs_m);
    for (size_t k = 0; k < c->stream_refs.size(); k++)
        if (c->stream_refs[k] == ref) { c->stream_refs[k] = c->stream_refs.back(); c->stream_refs.pop_back(); break; }
}
void ctx_set_window(lzfse_mi_ctx *c, EncWindow *w) { c->window = w; }
void ctx_set_pinned_out(lzfse_mi_ctx *c, bool on) { c->pinned_out = on; }
void ctx_set_defer_out(lzfse_mi_ctx *c, bool on) { c->defer_out = on; c->defer_last = nullptr; }
void *ctx_deferred_event(lzfse_mi_ctx *c) { return (void *)c->defer_last; }
std::shared_ptr<StreamBox> ctx_stream_box(lzfse_mi_ctx *c) {
    std::lock_guard<std::mutex> g(c->stream_refs_m);
    if (!c->stream_box) {
        try { c->stream_box = std::make_shared<StreamBox>(); } catch (...) { return nullptr; }
        c->stream_box->device = c->device;
    }
    return c->stream_box;
}
lzfse_mi_ctx *box_peer(StreamBox &b) {
    if (b.dead) return nullptr;
    if (!b.peer) {
        if (lzfse_mi_create(b.device, &b.peer) != LZFSE_MI_OK) { b.peer = nullptr; return nullptr; }
        b.peer->is_peer = true;
    }
    return b.peer;
}
int ctx_diag_chain(lzfse_mi_ctx *c) { return c->diag_chain; }
}  // namespace lzmi

static void timing_begin(lzfse_mi_ctx *c) {
    c->ev_used = 0;
    c->spans.clear();
}

static void timing_end(lzfse_mi_ctx *c) {
    lzfse_mi_timings &t = c->last;
    memset(&t, 0, sizeof t);
    if (!c->timing) return;
    for (auto &sp : c->spans) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) != hipSuccess) continue;
        int k = -1;
        for (int i = 0; i < t.n_stages; i++)
            if (t.names[i] == sp.name || !strcmp(t.names[i], sp.name)) k = i;
        if (k < 0) {
            if (t.n_stages == LZFSE_MI_MAX_STAGES) continue;
            k = t.n_stages++;
            t.names[k] = sp.name;
        }
        t.ms[k] += ms;
        t.launches[k] += 1;
    }
}

#define HIP_TRY(x) do { if ((x) != hipSuccess) { return LZFSE_MI_IO; } } while (0)

// b[b_off ..] = a[a_off ..] for len bytes, one workgroup per descriptor: gathers the small inputs of a batch into one
// staging buffer and scatters their encoded streams back
struct SmallDesc {
    uint64_t a_off, b_off;
    uint32_t len, pad;
};
__global__ void small_copy_kernel(const uint8_t *__restrict__ a, uint8_t *__restrict__ b, const SmallDesc *__restrict__ desc, uint32_t n) {
    if (blockIdx.x >= n) return;
    const SmallDesc d = desc[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < d.len; i += blockDim.x) b[d.b_off + i] = a[d.a_off + i];
}

extern "C" {

#ifdef LZFSE_MI_DIAG
const char *lzfse_mi_version(void) { return "lzfse-mi355x 0.2.0 (gfx950, diagnostic build)"; }
#else
const char *lzfse_mi_version(void) { return "lzfse-mi355x 0.2.0 (gfx950)"; }
#endif

const char *lzfse_mi_status_string(int s) {
    switch (s) {
    case LZFSE_MI_OK: return "ok";
    case LZFSE_MI_IO: return "io / HIP runtime error";
    case LZFSE_MI_BAD_BLOCK: return "bad block";
    case LZFSE_MI_BAD_BIT_STREAM: return "bad bitstream";
    case LZFSE_MI_BAD_D_VALUE: return "bad D value";
    case LZFSE_MI_BAD_READER_STATE: return "bad reader state";
    case LZFSE_MI_BUFFER_OVERFLOW: return "buffer overflow";
    case LZFSE_MI_PAYLOAD_OVERFLOW: return "bad payload overflow";
    case LZFSE_MI_PAYLOAD_UNDERFLOW: return "bad payload underflow";
    case LZFSE_MI_UNSUPPORTED: return "unsupported input";
    case LZFSE_MI_NO_DEVICE: return "no usable HIP device";
    case LZFSE_MI_BAD_ARGUMENT: return "bad argument";
    case LZFSE_MI_FSE_BAD_LITERAL_BITS: return "FSE: bad literal bits";
    case LZFSE_MI_FSE_BAD_LITERAL_COUNT: return "FSE: bad literal count";
    case LZFSE_MI_FSE_BAD_LITERAL_PAYLOAD: return "FSE: bad literal payload";
    case LZFSE_MI_FSE_BAD_LITERAL_STATE: return "FSE: bad literal state";
    case LZFSE_MI_FSE_BAD_LMD_BITS: return "FSE: bad LMD bits";
    case LZFSE_MI_FSE_BAD_LMD_COUNT: return "FSE: bad LMD count";
    case LZFSE_MI_FSE_BAD_LMD_PAYLOAD: return "FSE: bad LMD payload";
    case LZFSE_MI_FSE_BAD_LMD_STATE: return "FSE: bad LMD state";
    case LZFSE_MI_FSE_BAD_PAYLOAD_COUNT: return "FSE: bad payload count";
    case LZFSE_MI_FSE_BAD_RAW_BYTE_COUNT: return "FSE: bad raw byte count";
    case LZFSE_MI_FSE_BAD_READER_STATE: return "FSE: bad reader state";
    case LZFSE_MI_FSE_BAD_WEIGHT_PAYLOAD: return "FSE: bad weight payload";
    case LZFSE_MI_FSE_BAD_WEIGHT_PAYLOAD_COUNT: return "FSE: bad weight payload count";
    case LZFSE_MI_FSE_WEIGHT_PAYLOAD_OVERFLOW: return "FSE: weight payload overflow";
    case LZFSE_MI_FSE_WEIGHT_PAYLOAD_UNDERFLOW: return "FSE: weight payload underflow";
    case LZFSE_MI_VN_BAD_PAYLOAD_COUNT: return "VN: bad payload count";
    case LZFSE_MI_VN_BAD_PAYLOAD: return "VN: bad payload";
    case LZFSE_MI_VN_BAD_OPCODE: return "VN: bad opcode";
    default: return "unknown status";
    }
}

int lzfse_mi_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0 ? n : 0;
}

int lzfse_mi_create(int device, lzfse_mi_ctx **out) {
    if (!out) return LZFSE_MI_BAD_ARGUMENT;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return LZFSE_MI_NO_DEVICE;
    if (device < 0 || device >= n) return LZFSE_MI_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return LZFSE_MI_NO_DEVICE;
    lzfse_mi_ctx *c = new (std::nothrow) lzfse_mi_ctx();
    if (!c) return LZFSE_MI_IO;
    c->device = device;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return LZFSE_MI_NO_DEVICE;
    }
    c->stream = c->own_stream;
    *out = c;
    return LZFSE_MI_OK;
}

void lzfse_mi_destroy(lzfse_mi_ctx *c) {
    if (!c) return;
    {
        // stream objects that outlive the context: from now on they have none (their calls fail, their destructors free their own buffers)
        std::lock_guard<std::mutex> g(c->stream_refs_m);
        for (lzfse_mi_ctx **r : c->stream_refs) *r = nullptr;
        c->stream_refs.clear();
    }
    if (c->stream_box) {
        // (a window in flight on the stream objects' context finishes first: its job holds the mutex)
        std::lock_guard<std::mutex> g(c->stream_box->m);
        c->stream_box->dead = true;
        if (c->stream_box->peer) lzfse_mi_destroy(c->stream_box->peer);
        c->stream_box->peer = nullptr;
    }
    if (c->xfer_stream) {
        (void)hipStreamSynchronize(c->xfer_stream);
        (void)hipStreamDestroy(c->xfer_stream);
        for (auto &e : c->defer_ev) if (e) (void)hipEventDestroy(e);
    }
    c->d_out2.release();
    for (auto &w : c->worker) { delete w; w = nullptr; }
    for (auto &s : c->shadow) { if (s) lzfse_mi_destroy(s); s = nullptr; }
    (void)hipSetDevice(c->device);
    if (c->split_ev) (void)hipEventDestroy(c->split_ev);
    for (auto &g : c->gates) if (g.ev) (void)hipEventDestroy(g.ev);
    (void)hipStreamSynchronize(c->stream);
    for (DevBuf *b : {&c->d_streams, &c->d_walk, &c->d_plan, &c->d_blocks, &c->d_bres, &c->d_sres,
                      &c->d_lmds, &c->d_lits, &c->d_origin, &c->d_jerr, &c->d_wcache, &c->d_fwalk, &c->d_ck, &c->d_lzp, &c->d_in, &c->d_out, &c->d_small})
        b->release();
    enc_scratch_release(c->enc);
    for (lzmi::PinVec &h : c->h_ctl) h.release();
    c->h_done.release();
    delete c->host_worker;
    c->host_worker = nullptr;
    if (c->host_peer) lzfse_mi_destroy(c->host_peer);
    c->host_peer = nullptr;
    for (LaneWorker *w : c->copy_workers) delete w;
    c->copy_workers.clear();
    for (hipEvent_t e : c->host_ev) (void)hipEventDestroy(e);
    c->host_ev.clear();
    c->h_in.release();
    c->h_out.release();
    c->h_small.release();
    for (auto &b : c->spare.b) b.release();
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int lzfse_mi_set_stream(lzfse_mi_ctx *c, void *hip_stream) {
    if (!c) return LZFSE_MI_BAD_ARGUMENT;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return LZFSE_MI_OK;
}

size_t lzfse_mi_encode_bound(size_t n) { return n + n / 2 + n / 4 + 4096; }

int lzfse_mi_enable_timing(lzfse_mi_ctx *c, int enable) {
    if (!c) return LZFSE_MI_BAD_ARGUMENT;
    c->timing = enable != 0;
    return LZFSE_MI_OK;
}

int lzfse_mi_get_timings(lzfse_mi_ctx *c, lzfse_mi_timings *out) {
    if (!c || !out) return LZFSE_MI_BAD_ARGUMENT;
    *out = c->last;
    // a split call ran sub-batches side by side: stages are reported with the launches and the device time of all
    // lanes (the spans of the lanes overlap on the device)
    for (int lane = 0; lane < c->last_split; lane++) {
        if (!c->shadow[lane]) continue;
        const lzfse_mi_timings &s = c->shadow[lane]->last;
        for (int j = 0; j < s.n_stages; j++) {
            int k = -1;
            for (int i = 0; i < out->n_stages; i++)
                if (out->names[i] == s.names[j] || !strcmp(out->names[i], s.names[j])) k = i;
            if (k < 0) {
                if (out->n_stages == LZFSE_MI_MAX_STAGES) continue;
                k = out->n_stages++;
                out->names[k] = s.names[j];
                out->ms[k] = 0; out->launches[k] = 0;
            }
            out->ms[k] += s.ms[j];
            out->launches[k] += s.launches[j];
        }
    }
    return LZFSE_MI_OK;
}

// decode/probe.rs:11-35 on the host (pure header arithmetic, no payload is touched). On a header-level error *raw_len
// still receives the raw size of the blocks before it: enough capacity for lzfse_mi_decode to reach the error the
// reference would report (an error inside an earlier block comes first, decoder.rs:76-99).
int lzfse_mi_decode_size(const uint8_t *src, size_t n, uint64_t *raw_len) {
    if ((!src && n) || !raw_len) return LZFSE_MI_BAD_ARGUMENT;
    size_t pos = 0;
    uint64_t total = 0;
    int rc = LZFSE_MI_OK;
    for (;;) {
        if (n - pos < 4) { rc = LZFSE_MI_PAYLOAD_UNDERFLOW; break; }
        uint32_t magic = ld_u32(src + pos);
        size_t avail = n - pos;
        uint64_t skip;
        uint32_t n_raw;
        if (magic == MAGIC_EOS) {
            if (avail != 4) rc = LZFSE_MI_PAYLOAD_OVERFLOW;
            break;
        }
        if (magic == MAGIC_VX2 || magic == MAGIC_VX1) {
            bool v1 = magic == MAGIC_VX1;
            if (avail < (v1 ? V1_HEADER_SIZE : V2_HEADER_SIZE)) { rc = LZFSE_MI_PAYLOAD_UNDERFLOW; break; }
            FseHeader h;
            int e = v1 ? fse_load_v1(src + pos, h) : fse_load_v2(src + pos, h);
            if (e) { rc = e; break; }
            skip = (uint64_t)h.hdr_size + h.lit_payload + h.lmd_payload;
            n_raw = h.n_raw;
        } else if (magic == MAGIC_VXN) {
            if (avail < 12) { rc = LZFSE_MI_PAYLOAD_UNDERFLOW; break; }
            n_raw = ld_u32(src + pos + 4);
            skip = 12ull + ld_u32(src + pos + 8);
        } else if (magic == MAGIC_RAW) {
            if (avail < 8) { rc = LZFSE_MI_PAYLOAD_UNDERFLOW; break; }
            n_raw = ld_u32(src + pos + 4);
            skip = 8ull + n_raw;
        } else {
            rc = LZFSE_MI_BAD_BLOCK;
            break;
        }
        if (skip >= avail) {
            // cut short: a bvx1 / bvx2 / bvxn block is still decoded as far as it goes (its own errors come first)
            // (an LZVN op byte yields at most 271 / 2 output bytes, so a cut block cannot produce more than 136 x avail)
            if (magic == MAGIC_VXN) total += std::min<uint64_t>(n_raw, 136ull * avail);
            else if (magic != MAGIC_RAW) total += n_raw;
            rc = LZFSE_MI_PAYLOAD_UNDERFLOW;
            break;
        }
        pos += (size_t)skip;
        // (a sound bvxn block never yields more than 136 bytes per payload byte: a header that promises more -- 4 GiB over a
        // few bytes -- fails in the LZVN decoder with the reference's code, and must not size anybody's buffers meanwhile)
        total += magic == MAGIC_VXN ? std::min<uint64_t>(n_raw, 136ull * (skip - 12)) : n_raw;
    }
    *raw_len = total;
    return rc;
}

// ---------------------------------------------------------------------------- decode (device)

// streams from this size on are worth several workgroups in the LZ stage (measured on the Snappy files: never slower from 100 KB on)
static constexpr uint64_t PIPE_MIN_RAW = 96ull << 10;

static int decode_batch_device_one(lzfse_mi_ctx *c, size_t count, const void *d_src, const uint64_t *src_off,
                                   const uint64_t *src_len, void *d_dst, const uint64_t *dst_off,
                                   const uint64_t *dst_cap, uint64_t *out_lens, int *statuses) {
    if (!c || (count && (!src_off || !src_len || !dst_off || !dst_cap || !out_lens || !statuses)))
        return LZFSE_MI_BAD_ARGUMENT;
    if (count == 0) return LZFSE_MI_OK;
    if (count > 0x7FFFFFFFu) return LZFSE_MI_BAD_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    timing_begin(c);
    hipStream_t st = c->stream;
    const uint32_t ns = (uint32_t)count;
#ifdef LZFSE_MI_DIAG
    // (LZFSE_MI_OPT_DIAG_STATS & 4: where the host's time goes in this call -- scripts/stall_probe.py)
    const bool trace_host = (c->diag_stats & 4) != 0;
    std::chrono::steady_clock::time_point tp[6];
    auto mark = [&](int k) { if (trace_host) tp[k] = std::chrono::steady_clock::now(); };
#else
    auto mark = [](int) {};
#endif
    mark(0);
    enum { DH_STREAMS, DH_WALK, DH_PLAN, DH_SRES, DH_STATE, DH_MLIST };
    lzmi::CtlArray<StreamIn> h_streams(c->h_ctl[DH_STREAMS], ns);
    uint64_t src_total = 0;
    // walk cache: the count pass leaves its descriptors there (one per ~2 KiB of input plus a few per stream; a real
    // bvx2 block is far larger), so that placing them afterwards is a parallel copy instead of a second serial walk
    uint64_t cache_total = 0;
    for (uint32_t i = 0; i < ns; i++) {
        const uint64_t cap = src_len[i] / 2048 + 4;
        h_streams[i] = {src_off[i], src_len[i], dst_off[i], dst_cap[i], cache_total, cap};
        cache_total += cap;
        src_total = std::max<uint64_t>(src_total, src_off[i] + src_len[i]);
    }
    if (!c->d_wcache.ensure(cache_total * sizeof(BlockDesc)) ||
        !c->d_streams.ensure(ns * sizeof(StreamIn)) || !c->d_walk.ensure(ns * sizeof(StreamWalk)) ||
        !c->d_plan.ensure(ns * sizeof(StreamPlan)) || !c->d_sres.ensure(ns * sizeof(StreamResult)))
        return LZFSE_MI_IO;
    HIP_TRY(hipMemcpyAsync(c->d_streams.p, h_streams.data(), ns * sizeof(StreamIn), hipMemcpyHostToDevice, st));
    // pass 1: count. Streams of 8 MiB and more (hundreds of blocks: the serial walk pays a round trip per block) first go
    // through the parallel walk; what it cannot settle (and every smaller stream) is walked serially, one thread per stream.
    {
        StageTimer t(c, "dec_walk");
        std::vector<uint32_t> elig;
        uint64_t max_len = 0;
        const uint64_t fw_min = c->diag_walk == 1 ? 0ull : (8ull << 20);
        if (c->diag_walk != 2)
            for (uint32_t i = 0; i < ns && elig.size() < 4096; i++)
                if (src_len[i] >= fw_min && src_len[i] >= 36 && src_len[i] < 0xFFFFFFF0ull) { elig.push_back(i); max_len = std::max<uint64_t>(max_len, src_len[i]); }
        const uint32_t *d_settled = nullptr;
        if (!elig.empty()) {
            const size_t ne = elig.size();
            // layout: settled[ns] | count[ne] | elig[ne] | cand[ne * cap] (8-byte aligned)
            const size_t head = ((size_t)ns + 2 * ne + 1) & ~(size_t)1;
            if (!c->d_fwalk.ensure(head * 4 + ne * lzmi::fastwalk_cap() * sizeof(uint2))) return LZFSE_MI_IO;
            uint32_t *fw = (uint32_t *)c->d_fwalk.p;
            HIP_TRY(hipMemsetAsync(fw, 0, ((size_t)ns + ne) * 4, st));
            HIP_TRY(hipMemcpyAsync(fw + ns + ne, elig.data(), ne * 4, hipMemcpyHostToDevice, st));
            launch_dec_fastwalk((const uint8_t *)d_src, (const StreamIn *)c->d_streams.p, fw + ns + ne, (uint32_t)ne, max_len, fw + ns,
                                (uint2 *)(fw + head), (StreamWalk *)c->d_walk.p, (BlockDesc *)c->d_wcache.p, fw, st);
            d_settled = fw;
        }
        launch_dec_walk(false, (const uint8_t *)d_src, (const StreamIn *)c->d_streams.p, ns,
                        (StreamWalk *)c->d_walk.p, nullptr, (BlockDesc *)c->d_wcache.p, d_settled, st);
    }
    mark(1);
    lzmi::CtlArray<StreamWalk> h_walk(c->h_ctl[DH_WALK], ns);
    HIP_TRY(hipMemcpyAsync(h_walk.data(), c->d_walk.p, ns * sizeof(StreamWalk), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    mark(2);
    lzmi::CtlArray<StreamPlan> h_plan(c->h_ctl[DH_PLAN], ns);
    uint64_t nb = 0, nl = 0, nu = 0, nj = 0;
    int jump_mode = c->diag_lz_jump;  // -1: by cost; the diagnostic build can force 0 (never) or 1 (always)
    // Streams >= 2 MiB may take the pointer-jumping LZ path. One workgroup per stream copies ~0.63 GB/s whatever else
    // runs, the jumping passes move ~40 GB/s over all eligible bytes of ALL sub-batches together (this is one of up to
    // three running side by side, hence 42 and not 64): jumping pays when the largest stream, not the batch, sets the
    // time (one 64 MiB stream: yes; 128 streams of 4 MiB: no).
    if (jump_mode < 0) {
        uint64_t el_bytes = 0, el_max = 0;
        for (uint32_t i = 0; i < ns; i++) {
            if (h_walk[i].raw_total > dst_cap[i] || h_walk[i].n_vxn != 0) continue;
            if (h_walk[i].raw_total >= (2ull << 20)) { el_bytes += h_walk[i].raw_total; el_max = std::max<uint64_t>(el_max, h_walk[i].raw_total); }
        }
        // (with at most 64 streams the tile path gives each of them four workgroups and about 1.7 GB/s: dec_lzp_kernel)
        const bool piped = c->opt_pipe != 1 && !c->pipe_broken && (size_t)ns * (size_t)std::max(1, c->lane_share) <= 64;
        if (el_max * (piped ? 24 : 42) <= el_bytes) jump_mode = 0;
    }
    for (uint32_t i = 0; i < ns; i++) {
        StreamPlan &p = h_plan[i];
        p.blk_base = nb; p.lmd_base = nl; p.lit_base = nu; p.n_blocks = h_walk[i].n_blocks; p.skip = 0;
        p.jbase = 0; p.jump = 0; p.turn = i; p.pipe = 0;
        p.pad = h_walk[i].status != 0;   // (a stream that fails in the end leaves nothing in the caller's buffer: no host image of its first blocks)
        statuses[i] = LZFSE_MI_OK;
        out_lens[i] = 0;
        // A stream whose walk failed at block k still has its first k blocks decoded: the reference decodes in order, so
        // an error inside one of them is the one it reports (decoder.rs:76-99); otherwise the walk's status stands.
        if (h_walk[i].status && h_walk[i].n_blocks == 0) { statuses[i] = h_walk[i].status; p.skip = 1; p.n_blocks = 0; continue; }
        // (a stream whose headers promise more than dst_cap is decoded block by block up to the one that does not fit:
        // the header counts of a damaged stream are not to be trusted, and the first error in stream order wins)
        nb += h_walk[i].n_blocks; nl += h_walk[i].n_lmds; nu += h_walk[i].n_lits;
        // large streams: LZ stage by pointer jumping (origin indices are 31-bit over the whole batch, the top bit marks final bytes)
        const bool big = jump_mode < 0 ? h_walk[i].raw_total >= (2ull << 20) : jump_mode > 0;
        // (an origin is 31 bits + the "final" bit, and all ones is the filler of undefined entries: positions up to 0x7FFFFFFE, so
        // that a stream of 0x7FFF_FFFF bytes -- the largest the slice calls take -- still goes this way when it is alone)
        if (big && h_walk[i].n_vxn == 0 && nj + h_walk[i].raw_total <= 0x7FFFFFFFull && h_walk[i].raw_total > 0 &&
            h_walk[i].raw_total <= dst_cap[i]) {
            p.jump = 1; p.jbase = nj;
            nj += (h_walk[i].raw_total + 3) & ~3ull;
        }
    }
    if (nb > 0x7FFFFFFFull) return LZFSE_MI_UNSUPPORTED;
    c->diag_last_lmds = nl;
    // Streams of the tile kernel that are large and few share several workgroups each (dec_lzp_kernel): with hundreds of
    // streams, or small ones, one workgroup per stream already fills the chip.
    std::vector<uint32_t> mlist;
    int pipe_variant = 1;
    uint32_t pipe_k = 0;
    if (c->opt_pipe != 1 && !c->pipe_broken && !c->pipe_tested) {
        // before the pipelined LZ kernel is used for the first time on this context: its hand-over, on this device
        c->pipe_tested = true;
        uint32_t res[2] = {1, 0};   // (a self-test that could not run -- no memory, a failed copy -- is a refusal like one that failed)
        if (c->d_lzp.ensure((1088 + 2) * 4)) {
            uint32_t *tb = (uint32_t *)c->d_lzp.p;
            if (hipMemsetAsync(tb, 0, (1088 + 2) * 4, c->stream) == hipSuccess) {
                launch_dec_lzp_selftest(tb, tb + 1088, c->stream);
                if (hipMemcpyAsync(res, tb + 1088, 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) res[0] = 1;
            }
        }
        const bool same_xcd = (res[1] & 0xFF) == ((res[1] >> 8) & 0xFF);
        if (res[0] != 0 || (res[1] & 0x10000u) || !same_xcd) { c->pipe_broken = true; c->pipe_refusals++; }
    }
    if (c->opt_pipe != 1 && !c->pipe_broken) {
        const bool forced = c->opt_pipe > 1;
        size_t n_tile = 0;   // streams of the tile kernel: one workgroup each fills the chip when they are many
        for (uint32_t i = 0; i < ns; i++) {
            if (h_plan[i].skip || h_plan[i].jump || !h_plan[i].n_blocks) continue;
            n_tile++;
            if (h_walk[i].n_vxn == 0 && (forced || h_walk[i].raw_total >= PIPE_MIN_RAW)) mlist.push_back(i);
        }
        const size_t share = (size_t)std::max(1, c->lane_share);
        if (mlist.empty()) pipe_k = 0;
        else if (forced) { pipe_k = (uint32_t)(c->opt_pipe & 0xFF); pipe_variant = (c->opt_pipe >> 8) & 1; }
        else if (n_tile * share <= 128) {
            // 1024 threads and 124 KB of LDS: one workgroup per CU, and the hand-over chain of a stream (about 5 us per
            // ticket) is what bounds it from K = 4 on
            pipe_variant = 1;
            pipe_k = (uint32_t)std::min<size_t>(4, 256 / (mlist.size() * share));
            if (pipe_k < 2) pipe_k = 0;
        }
        if (pipe_k < 1) mlist.clear();
        for (uint32_t i : mlist) h_plan[i].pipe = pipe_k;
    }
    {
        // order of the tile kernel's workgroups: streams by decreasing size (plan[b].turn)
        std::vector<uint32_t> turn(ns);
        for (uint32_t i = 0; i < ns; i++) turn[i] = i;
        std::stable_sort(turn.begin(), turn.end(), [&](uint32_t a, uint32_t b) { return h_walk[a].raw_total > h_walk[b].raw_total; });
        for (uint32_t b = 0; b < ns; b++) h_plan[b].turn = turn[b];
        std::stable_sort(mlist.begin(), mlist.end(), [&](uint32_t a, uint32_t b) { return h_walk[a].raw_total > h_walk[b].raw_total; });
    }
    uint32_t *d_lzp_state = nullptr, *d_mlist = nullptr;
    if (!mlist.empty()) {
        if (!c->d_ck.ensure(((nl >> 8) + nb + 2) * sizeof(uint2)) || !c->d_lzp.ensure(((size_t)LZP_STATE_WORDS * ns + mlist.size()) * 4)) return LZFSE_MI_IO;
        d_lzp_state = (uint32_t *)c->d_lzp.p;
        d_mlist = d_lzp_state + LZP_STATE_WORDS * (size_t)ns;
        HIP_TRY(hipMemcpyAsync(d_mlist, mlist.data(), mlist.size() * 4, hipMemcpyHostToDevice, st));
    }
    if (!c->d_blocks.ensure((nb + 1) * sizeof(BlockDesc)) || !c->d_bres.ensure((nb + 1) * sizeof(BlockResult)) ||
        !c->d_lmds.ensure((nl + 64) * sizeof(LmdRec)) || !c->d_lits.ensure(nu + 256) || !c->d_origin.ensure((nj + 16) * 4) ||
        !c->d_jerr.ensure((size_t)ns * 4 + 128 * 4 + (nb + 1) * 4))
        return LZFSE_MI_IO;
    uint32_t *d_jerr = (uint32_t *)c->d_jerr.p, *d_jflags = d_jerr + ns;
    uint32_t *d_ohist = d_jflags + 64, *d_order = d_ohist + 64;   // FSE workgroup order (launch_dec_fse)
    {
        // (one launch for the call's small fills: every launch of a small call is microseconds of an idle device)
        static_assert(sizeof(BlockResult) % 4 == 0 && sizeof(StreamResult) % 4 == 0, "dword fills");
        void *const fp[5] = {d_jerr, d_jflags, c->d_bres.p, c->d_sres.p, d_lzp_state};
        const uint64_t fb[5] = {(uint64_t)ns * 4, 128 * 4, (nb + 1) * sizeof(BlockResult), ns * sizeof(StreamResult),
                                d_lzp_state ? (uint64_t)LZP_STATE_WORDS * ns * 4 : 0};
        const uint32_t fv[5] = {0xFFFFFFFFu, 0u, 0u, 0u, 0u};
        launch_dec_fills(fp, fb, fv, 5, st);
    }
    HIP_TRY(hipMemcpyAsync(c->d_plan.p, h_plan.data(), ns * sizeof(StreamPlan), hipMemcpyHostToDevice, st));
    {
        StageTimer t(c, "dec_walk");
        launch_dec_emit((const StreamIn *)c->d_streams.p, ns, (const StreamPlan *)c->d_plan.p, (const BlockDesc *)c->d_wcache.p,
                        cache_total, (BlockDesc *)c->d_blocks.p, st);
        bool rewalk = false;  // streams whose blocks did not fit their share of the cache
        for (uint32_t i = 0; i < ns; i++) rewalk |= !h_plan[i].skip && h_plan[i].n_blocks > h_streams[i].cache_cap;
        if (rewalk)
            launch_dec_walk(true, (const uint8_t *)d_src, (const StreamIn *)c->d_streams.p, ns, nullptr,
                            (const StreamPlan *)c->d_plan.p, (BlockDesc *)c->d_blocks.p, nullptr, st);
    }
    // all streams on the pointer-jumping path (a lone large stream, a window of a stream): its first step rides on the entropy kernel
    bool jump_fused = nj != 0;
    for (uint32_t i = 0; i < ns; i++) jump_fused &= h_plan[i].skip || h_plan[i].jump;
    const JumpFuse jf = {(const StreamIn *)c->d_streams.p, (const StreamPlan *)c->d_plan.p, (uint8_t *)d_dst, (uint32_t *)c->d_origin.p, d_jerr, nj};
    {
        StageTimer t(c, "dec_fse");
        launch_dec_fse((const uint8_t *)d_src, src_total, (const BlockDesc *)c->d_blocks.p, (uint32_t)nb,
                       (uint8_t *)c->d_lits.p, (LmdRec *)c->d_lmds.p, (BlockResult *)c->d_bres.p, d_ohist, d_order,
                       jump_fused ? &jf : nullptr, st);
    }
    {
        StageTimer t(c, "dec_lz");
        // one workgroup per stream: the 1024-thread / 32 KiB-tile kernel (one workgroup per CU) finishes a stream 2.6 times
        // sooner than the 256-thread / 8 KiB-tile kernel (five per CU), which moves more bytes once there are about seven
        // streams per CU in flight (Snappy files x R, dec_lz ms: 1 080 streams 1.45 / 2.20, 1 536: 1.94 / 2.23, 2 040: 2.62 / 2.26)
        int variant = (uint64_t)ns * (uint64_t)std::max(1, c->lane_share) >= 1792 ? 0 : 1;
        if (c->diag_lz_variant >= 0) variant = c->diag_lz_variant;
        if (!jump_fused)   // (all on the pointer-jumping path: nothing for the tile kernel)
        launch_dec_lz(variant, (const uint8_t *)d_src, (const StreamIn *)c->d_streams.p,
                      (const StreamPlan *)c->d_plan.p, ns, (const BlockDesc *)c->d_blocks.p,
                      (const BlockResult *)c->d_bres.p, (const LmdRec *)c->d_lmds.p, (const uint8_t *)c->d_lits.p,
                      (uint8_t *)d_dst, (StreamResult *)c->d_sres.p, c->lane_share <= 1 ? c->mirror : lzmi::OutMirror(), st);
        if (!mlist.empty()) {
            // (tickets of two LMDs per thread when the streams are long enough to keep their workgroups in tickets: dec_lzp_kernel)
            uint64_t pipe_raw = 0;
            for (uint32_t i : mlist) pipe_raw += h_walk[i].raw_total;
            const int pipe_lpt = pipe_raw >= ((uint64_t)mlist.size() << 20) ? 2 : 1;
            launch_dec_lzp(pipe_variant, pipe_k, pipe_lpt, (const uint8_t *)d_src, (const StreamIn *)c->d_streams.p,
                           (const StreamPlan *)c->d_plan.p, d_mlist, (uint32_t)mlist.size(), (const BlockDesc *)c->d_blocks.p,
                           (uint32_t)nb, (const BlockResult *)c->d_bres.p, (const LmdRec *)c->d_lmds.p, (const uint8_t *)c->d_lits.p,
                           (uint2 *)c->d_ck.p, (uint8_t *)d_dst, (StreamResult *)c->d_sres.p, d_lzp_state, (uint32_t)c->diag_pipe_scatter, st);
        }
    }
    if (nj) {
        launch_dec_jump((const uint8_t *)d_src, (const StreamIn *)c->d_streams.p, (const StreamPlan *)c->d_plan.p,
                        (const StreamWalk *)c->d_walk.p, ns, (const BlockDesc *)c->d_blocks.p, (uint32_t)nb,
                        (const BlockResult *)c->d_bres.p, (const LmdRec *)c->d_lmds.p, (const uint8_t *)c->d_lits.p, (uint8_t *)d_dst,
                        (uint32_t *)c->d_origin.p, nj, d_jerr, d_jflags, (StreamResult *)c->d_sres.p, c, jump_fused, st);
    }
    mark(3);
    lzmi::CtlArray<StreamResult> h_sres(c->h_ctl[DH_SRES], ns);
    lzmi::CtlArray<uint32_t> h_state(c->h_ctl[DH_STATE], mlist.empty() ? 0 : (size_t)LZP_STATE_WORDS * ns);
    HIP_TRY(hipMemcpyAsync(h_sres.data(), c->d_sres.p, ns * sizeof(StreamResult), hipMemcpyDeviceToHost, st));
    mark(4);
    if (!mlist.empty()) HIP_TRY(hipMemcpyAsync(h_state.data(), d_lzp_state, h_state.size() * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    mark(5);
#ifdef LZFSE_MI_DIAG
    if (trace_host) {
        auto ms = [&](int a, int b) { return std::chrono::duration<double, std::milli>(tp[b] - tp[a]).count(); };
        fprintf(stderr, "dec_host ctx=%p streams=%u total=%.3f upload+walk_launch=%.3f walk_wait=%.3f plan+launches=%.3f results_copy_call=%.3f wait=%.3f\n", (void *)c, ns, ms(0, 5),
                ms(0, 1), ms(1, 2), ms(2, 3), ms(3, 4), ms(4, 5));
    }
#endif
    if (hipGetLastError() != hipSuccess) return LZFSE_MI_IO;
    {
        // The pipelined LZ kernel hands data from workgroup to workgroup through one XCD's L2 and checks that it may
        // (XCC id of every workgroup that touches a stream). If a launch ever says no, its streams are decoded again by
        // the one-workgroup kernel, and this context stops using the pipelined one.
        bool again = false;
        for (uint32_t i : mlist) again |= h_state[(size_t)LZP_STATE_WORDS * i + LZP_BAD] != 0;   // the word a misplaced workgroup sets
        if (again) {
            c->pipe_broken = true;
            c->pipe_refusals++;
            for (uint32_t i : mlist) h_plan[i].pipe = 0;
            HIP_TRY(hipMemcpyAsync(c->d_plan.p, h_plan.data(), ns * sizeof(StreamPlan), hipMemcpyHostToDevice, st));
            {
                StageTimer t(c, "dec_lz_again");
                launch_dec_lz(1, (const uint8_t *)d_src, (const StreamIn *)c->d_streams.p, (const StreamPlan *)c->d_plan.p, ns,
                              (const BlockDesc *)c->d_blocks.p, (const BlockResult *)c->d_bres.p, (const LmdRec *)c->d_lmds.p,
                              (const uint8_t *)c->d_lits.p, (uint8_t *)d_dst, (StreamResult *)c->d_sres.p, c->lane_share <= 1 ? c->mirror : lzmi::OutMirror(), st);
            }
            HIP_TRY(hipMemcpyAsync(h_sres.data(), c->d_sres.p, ns * sizeof(StreamResult), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (hipGetLastError() != hipSuccess) return LZFSE_MI_IO;
        }
    }
    if ((c->diag_stats & 4) && nj) {
        uint32_t hf[20] = {};
        if (hipMemcpy(hf, d_jflags, sizeof hf, hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "jump rounds that still moved bytes:");
            for (int r2 = 0; r2 < 17; r2++) fprintf(stderr, " %u", hf[r2]);
            fprintf(stderr, "\n");
        }
    }
    if ((c->diag_stats & 4) && !mlist.empty()) {
        for (size_t k = 0; k < mlist.size() && k < 4; k++) {
            const uint64_t *q = (const uint64_t *)(h_state.data() + (size_t)LZP_STATE_WORDS * mlist[k] + LZP_DIAG);
            fprintf(stderr, "lzp[%u] K=%u tickets=%llu cycles: setup=%llu ahead=%llu wait=%llu turn=%llu (copies from earlier output %llu, gather %llu, write-back and hand-over %llu)\n", mlist[k], pipe_k,
                    (unsigned long long)q[4], (unsigned long long)q[0], (unsigned long long)q[1], (unsigned long long)q[2],
                    (unsigned long long)q[3], (unsigned long long)q[5], (unsigned long long)q[6], (unsigned long long)q[7]);
        }
    }
    if (c->diag_stats & 4) {
        for (uint32_t i = 0; i < ns && i < 16; i++) {
            const StreamResult &q = h_sres[i];
            fprintf(stderr, "lz[%u] out=%llu groups=%u dep=%u long=%u cyc scan=%llu short=%llu long=%llu dep=%llu wb=%llu total=%llu\n", i,
                    (unsigned long long)q.out_len, q.groups, q.n_dep, q.n_long, (unsigned long long)q.cyc[0],
                    (unsigned long long)q.cyc[1], (unsigned long long)q.cyc[2], (unsigned long long)q.cyc[3],
                    (unsigned long long)q.cyc[4], (unsigned long long)q.cyc[5]);
        }
    }
    for (uint32_t i = 0; i < ns; i++) {
        if (h_plan[i].skip) continue;
        statuses[i] = h_sres[i].status ? h_sres[i].status : h_walk[i].status;
        out_lens[i] = statuses[i] ? 0 : h_sres[i].out_len;
    }
    c->detail.assign(ns, 0u);
    for (uint32_t i = 0; i < ns; i++)
        if (statuses[i] && statuses[i] == h_walk[i].status &&
            (statuses[i] == LZFSE_MI_BAD_BLOCK || statuses[i] == LZFSE_MI_FSE_BAD_LMD_COUNT || statuses[i] == LZFSE_MI_FSE_BAD_LITERAL_COUNT))
            c->detail[i] = h_walk[i].detail;
    timing_end(c);
    return LZFSE_MI_OK;
}

#ifdef LZFSE_MI_DIAG
// Debug hook of the diagnostic build for stage-level parity tests (tests/test_gpu_decode.py): the LMD records dec_fse_kernel left
// for the LZ stage in the last decode pass on this context -- pairs (l | m << 16, d with D = 0 substituted: lmd_type.rs:153-160),
// streams in call order, blocks in stream order. Not part of the ABI; the product library does not export it.
extern "C" LZFSE_MI_API int lzfse_mi_debug_last_lmds(lzfse_mi_ctx *c, uint32_t *h_pairs, size_t cap_lmds, size_t *n_lmds) {
    if (!c || !n_lmds) return LZFSE_MI_BAD_ARGUMENT;
    *n_lmds = (size_t)c->diag_last_lmds;
    if (!h_pairs || cap_lmds < c->diag_last_lmds) return LZFSE_MI_BUFFER_OVERFLOW;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->diag_last_lmds) HIP_TRY(hipMemcpy(h_pairs, c->d_lmds.p, c->diag_last_lmds * sizeof(LmdRec), hipMemcpyDeviceToHost));
    return LZFSE_MI_OK;
}
#endif

static int encode_batch_device_one(lzfse_mi_ctx *c, size_t count, const void *d_src, const uint64_t *src_off,
                                   const uint64_t *src_len, void *d_dst, const uint64_t *dst_off,
                                   const uint64_t *dst_cap, uint64_t *out_lens, int *statuses) {
    if (!c || (count && (!src_off || !src_len || !dst_off || !dst_cap || !out_lens || !statuses)))
        return LZFSE_MI_BAD_ARGUMENT;
    if (count == 0) return LZFSE_MI_OK;
    if (count > 0x7FFFFFFFu) return LZFSE_MI_BAD_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    c->detail.assign(count, 0u);
    // ---- inputs <= 4096 bytes: the reference's host-side size classes (raw / LZVN, frontend_bytes.rs:63-111). All of
    // them travel together: one gather kernel + one D2H, the host encoder, one H2D + one scatter kernel, queued in
    // front of the device pipeline of the large streams (whose final synchronisation covers the copies back).
    std::vector<size_t> small;
    for (size_t i = 0; i < count; i++)
        if (src_len[i] <= VN_CUTOFF) small.push_back(i);
    std::vector<int> small_status(small.size(), 0);
    std::vector<uint64_t> small_len(small.size(), 0);
    if (!small.empty()) {
        constexpr size_t SLOT = 4096 + 256;   // a stream of <= 4096 bytes never encodes to more than n + 24 bytes
        const size_t ns = small.size();
        if (!c->d_small.ensure(2 * ns * SLOT + ns * sizeof(SmallDesc)) || !c->h_small.ensure(2 * ns * SLOT + ns * sizeof(SmallDesc)))
            return LZFSE_MI_IO;
        uint8_t *d_in = (uint8_t *)c->d_small.p, *d_out = d_in + ns * SLOT;
        SmallDesc *d_desc = (SmallDesc *)(d_out + ns * SLOT);
        uint8_t *h_in = (uint8_t *)c->h_small.p, *h_out = h_in + ns * SLOT;
        SmallDesc *hd = (SmallDesc *)(h_out + ns * SLOT);   // descriptors in pinned memory too: the copies are asynchronous
        for (size_t k = 0; k < ns; k++) hd[k] = {src_off[small[k]], (uint64_t)k * SLOT, (uint32_t)src_len[small[k]], 0u};
        HIP_TRY(hipMemcpyAsync(d_desc, hd, ns * sizeof(SmallDesc), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(small_copy_kernel, dim3((unsigned)ns), dim3(256), 0, c->stream, (const uint8_t *)d_src, d_in, d_desc, (uint32_t)ns);
        HIP_TRY(hipMemcpyAsync(h_in, d_in, ns * SLOT, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (size_t k = 0; k < ns; k++) {
            size_t n = 0;
            const size_t i = small[k];
            small_status[k] = lzmi::encode_small(h_in + k * SLOT, (size_t)src_len[i], h_out + k * SLOT, SLOT, &n, c->parse_ring);
            if (small_status[k] == 0 && n > dst_cap[i]) small_status[k] = LZFSE_MI_BUFFER_OVERFLOW;
            small_len[k] = small_status[k] ? 0 : n;
            hd[k] = {(uint64_t)k * SLOT, dst_off[i], (uint32_t)small_len[k], 0u};
        }
        HIP_TRY(hipMemcpyAsync(d_out, h_out, ns * SLOT, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_desc, hd, ns * sizeof(SmallDesc), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(small_copy_kernel, dim3((unsigned)ns), dim3(256), 0, c->stream, (const uint8_t *)d_out, (uint8_t *)d_dst, d_desc, (uint32_t)ns);
    }
    timing_begin(c);
    int r = enc_batch_device(c, (uint32_t)count, (const uint8_t *)d_src, src_off, src_len, (uint8_t *)d_dst,
                             dst_off, dst_cap, out_lens, statuses);
    timing_end(c);
    if (r) return r;
    if (!small.empty()) {
        HIP_TRY(hipStreamSynchronize(c->stream));   // (a batch of small inputs only: nothing else has waited for the copies)
        for (size_t k = 0; k < small.size(); k++) { statuses[small[k]] = small_status[k]; out_lens[small[k]] = small_len[k]; }
    }
    return LZFSE_MI_OK;
}

// ---------------------------------------------------------------------------- split batches

typedef int (*batch_dev_fn)(lzfse_mi_ctx *, size_t, const void *, const uint64_t *, const uint64_t *, void *,
                            const uint64_t *, const uint64_t *, uint64_t *, int *);

// Several stages of both directions are bound by latency or by a serial chain per stream/block and leave most of
// the chip idle (chains per tile, segment walkers, FSE blocks). A large batch is therefore cut into sub-batches of
// about equal size that run side by side: the caller's thread drives one on the context's stream, helper threads
// drive the others on shadow contexts (own stream and scratch each). Streams are independent, so the results are
// those of one call. Measured best: 2 lanes in both directions (3 encode lanes gain 2 % without stage timing, lose 4 % with it) (LZFSE_MI_LANES_ENC / _DEC override,
// LZFSE_MI_NO_SPLIT=1 turns it off).
static int split_batch(lzfse_mi_ctx *c, batch_dev_fn one, int lanes, bool stagger, size_t count, const void *d_src, const uint64_t *src_off,
                       const uint64_t *src_len, void *d_dst, const uint64_t *dst_off, const uint64_t *dst_cap,
                       uint64_t *out_lens, int *statuses) {
    if (!c) return LZFSE_MI_BAD_ARGUMENT;
    c->last_split = 0;
    uint64_t total = 0;
    if (count && src_len)
        for (size_t i = 0; i < count; i++) total += src_len[i];
    const bool no_split = false;
    if (lanes > LZFSE_MI_MAX_LANES) lanes = LZFSE_MI_MAX_LANES;
    while (lanes > 1 && (count < (size_t)4 * lanes || total < ((uint64_t)lanes << 21))) lanes--;
    auto unsplit = [&]() {
        c->lane_share = 1;
        c->detail.clear();
        const int r = one(c, count, d_src, src_off, src_len, d_dst, dst_off, dst_cap, out_lens, statuses);
        c->detail_out = c->detail;
        return r;
    };
    if (no_split || lanes < 2 || !src_off || !dst_off || !dst_cap || !out_lens || !statuses) return unsplit();
    if (hipSetDevice(c->device) != hipSuccess) return LZFSE_MI_IO;
    if (!c->split_ev && hipEventCreateWithFlags(&c->split_ev, hipEventDisableTiming) != hipSuccess) c->split_ev = nullptr;
    for (int k = 0; k + 1 < lanes; k++)
        if (!c->shadow[k]) {
            if (lzfse_mi_create(c->device, &c->shadow[k]) != LZFSE_MI_OK) { c->shadow[k] = nullptr; lanes = k + 1; break; }
            // (std::thread's constructor throws when no thread can be started: nothrow only covers the allocation)
            try { c->worker[k] = new (std::nothrow) LaneWorker(); } catch (...) { c->worker[k] = nullptr; }
            if (!c->worker[k]) { lzfse_mi_destroy(c->shadow[k]); c->shadow[k] = nullptr; lanes = k + 1; break; }
        }
    if (lanes < 2 || !c->split_ev) return unsplit();
    struct Part {
        std::vector<size_t> idx;
        std::vector<uint64_t> so, sl, dof, dc, ol;
        std::vector<int> st;
        uint64_t bytes = 0;
        int rc = 0;
    };
    std::vector<Part> part((size_t)lanes);
    for (size_t i = 0; i < count; i++) {
        size_t best = 0;
        for (size_t k = 1; k < part.size(); k++)
            if (part[k].bytes < part[best].bytes) best = k;
        Part &p = part[best];
        p.idx.push_back(i); p.so.push_back(src_off[i]); p.sl.push_back(src_len[i]);
        p.dof.push_back(dst_off[i]); p.dc.push_back(dst_cap[i]);
        p.bytes += src_len[i];
    }
    for (Part &p : part) { p.ol.assign(p.idx.size(), 0); p.st.assign(p.idx.size(), 0); }
    // the helper lanes start after everything already queued on the caller's stream
    bool ok = hipEventRecord(c->split_ev, c->stream) == hipSuccess;
    for (int k = 0; ok && k + 1 < lanes; k++) {
        ok = hipStreamWaitEvent(c->shadow[k]->stream, c->split_ev, 0) == hipSuccess;
        c->shadow[k]->timing = c->timing;
        c->shadow[k]->opt_pipe = c->opt_pipe;
        c->shadow[k]->parse_ring = c->parse_ring;
        c->shadow[k]->diag_lz_jump = c->diag_lz_jump; c->shadow[k]->diag_lz_variant = c->diag_lz_variant;
        c->shadow[k]->diag_stats = c->diag_stats; c->shadow[k]->diag_chain = c->diag_chain; c->shadow[k]->diag_walk = c->diag_walk;
        c->shadow[k]->diag_pipe_scatter = c->diag_pipe_scatter;
    }
    if (!ok) return unsplit();
    // staggered start (encode): lane k + 1 begins when lane k has queued its candidate kernel
    for (int k = 0; stagger && k + 1 < lanes; k++) {
        lzmi::LaneGate &g = c->gates[k];
        if (!g.ev && hipEventCreateWithFlags(&g.ev, hipEventDisableTiming) != hipSuccess) { g.ev = nullptr; stagger = false; }
    }
    for (int k = 0; k < lanes; k++) {
        lzfse_mi_ctx *cx = k == 0 ? c : c->shadow[k - 1];
        cx->lane_share = lanes;
        cx->gate_in = (stagger && k > 0) ? &c->gates[k - 1] : nullptr;
        cx->gate_out = (stagger && k + 1 < lanes) ? &c->gates[k] : nullptr;
        if (cx->gate_out) cx->gate_out->arm();
    }
    c->detail_out.assign(count, 0u);
    auto run = [&](lzfse_mi_ctx *cx, Part &p) {
        cx->detail.clear();
        p.rc = one(cx, p.idx.size(), d_src, p.so.data(), p.sl.data(), d_dst, p.dof.data(), p.dc.data(), p.ol.data(), p.st.data());
        for (size_t k = 0; k < p.idx.size() && k < cx->detail.size(); k++) c->detail_out[p.idx[k]] = cx->detail[k];
    };
    const size_t started = (size_t)lanes - 1;
    for (int k = 0; k + 1 < lanes; k++) {
        lzfse_mi_ctx *cx = c->shadow[k];
        Part *pp = &part[(size_t)k + 1];
        c->worker[k]->submit([&run, cx, pp] { run(cx, *pp); });
    }
    run(c, part[0]);
    for (int k = 0; k + 1 < lanes; k++) c->worker[k]->wait();
    for (int k = 0; k < lanes; k++) {
        lzfse_mi_ctx *cx = k == 0 ? c : c->shadow[k - 1];
        cx->gate_in = cx->gate_out = nullptr;
    }
    int rc = 0;
    for (Part &p : part) {
        for (size_t k = 0; k < p.idx.size(); k++) { out_lens[p.idx[k]] = p.ol[k]; statuses[p.idx[k]] = p.st[k]; }
        if (p.rc && !rc) rc = p.rc;
    }
    c->last_split = (int)started;
    return rc;
}

int lzfse_mi_decode_batch_device(lzfse_mi_ctx *c, size_t count, const void *d_src, const uint64_t *src_off,
                                 const uint64_t *src_len, void *d_dst, const uint64_t *dst_off,
                                 const uint64_t *dst_cap, uint64_t *out_lens, int *statuses) {
    if (!c) return LZFSE_MI_BAD_ARGUMENT;
    // One pass by default. (Round 2 ran two sub-batches side by side here: the entropy stage was bound by instruction issue and
    // a second lane filled its gaps. With the round-3 decode steps a single pass is as fast or faster at every batch size
    // measured -- Snappy x 64 / 256 / 512: 72.5 / 124.5 / 137.5 GB/s against 69.6 / 119.3 / 136.4 -- and few large streams
    // want the chip for one pass of the pipelined LZ kernel anyway. LZFSE_MI_OPT_DECODE_LANES still splits on request.)
    const int lanes = c->opt_lanes_dec ? c->opt_lanes_dec : 1;
    return split_batch(c, decode_batch_device_one, lanes, false, count, d_src, src_off, src_len, d_dst,
                       dst_off, dst_cap, out_lens, statuses);
}

int lzfse_mi_encode_batch_device(lzfse_mi_ctx *c, size_t count, const void *d_src, const uint64_t *src_off,
                                 const uint64_t *src_len, void *d_dst, const uint64_t *dst_off,
                                 const uint64_t *dst_cap, uint64_t *out_lens, int *statuses) {
    if (!c) return LZFSE_MI_BAD_ARGUMENT;
    // One pass below 448 MiB, two lanes up to 700 MiB, three beyond (round 5, profiles/r05_lanes_sweep.txt: Snappy x 16 / 64 / 128 /
    // 192 / 256 / 512 encode 17.6 / 27.3 / 30.5 / 32.2 / 30.0 / 30.6 GB/s as one pass, 9.6 / 26.8 / 29.8 / 32.9 / 34.3 / 35.5 with two
    // lanes, - / - / - / 31.6 / 34.9 / 36.5 with three; 256 MiB in 64 streams 31.8 against 30.4). Rounds 2 to 4 cut every call of
    // 8 streams and 4 MiB in two: the candidate stage fills every wave slot by itself, and what a second lane hides of the
    // latency-bound stages only pays once those are a large batch's worth.
    int lanes = c->opt_lanes_enc;
    if (!lanes) {
        uint64_t total = 0;
        for (size_t i = 0; src_len && i < count; i++) total += src_len[i];
        lanes = total >= (700ull << 20) ? 3 : total >= (448ull << 20) ? 2 : 1;
        // ... except a call of many small streams, 24 .. 100 MiB in 128 streams and more: every stage of such a call is one round of
        // resident workgroups, and two halves run their rounds side by side (each Snappy file x 256: 26 .. 47 MB, encode 19.1 / 17.3 /
        // 21.9 / 22.8 GB/s as one pass against 22.7 / 21.6 / 25.3 / 26.0 in two lanes; Snappy x 16 / x 32: 17.8 / 22.5 against 18.1 / 22.7;
        // 64 streams are better off as one pass up to 26 MB: profiles/r05_per_file_lanes.txt, r05_lanes_small.txt)
        if (lanes == 1 && count >= 128 && total >= (24ull << 20) && total < (100ull << 20)) lanes = 2;
    }
    return split_batch(c, encode_batch_device_one, lanes, c->opt_stagger != 0, count, d_src, src_off, src_len, d_dst, dst_off, dst_cap,
                       out_lens, statuses);
}

int lzfse_mi_get_info(lzfse_mi_ctx *c, int what, int64_t *value) {
    if (!c || !value) return LZFSE_MI_BAD_ARGUMENT;
    switch (what) {
    case LZFSE_MI_INFO_PIPE_REFUSALS: {
        // this context, the helper contexts of its split calls and of its large host calls
        int64_t n = c->pipe_refusals;
        for (lzfse_mi_ctx *sh : c->shadow) if (sh) n += sh->pipe_refusals;
        if (c->host_peer) {
            n += c->host_peer->pipe_refusals;
            for (lzfse_mi_ctx *sh : c->host_peer->shadow) if (sh) n += sh->pipe_refusals;
        }
        *value = n;
        return LZFSE_MI_OK;
    }
    default: return LZFSE_MI_BAD_ARGUMENT;
    }
}

int lzfse_mi_set_option(lzfse_mi_ctx *c, int option, int64_t value) {
    if (!c) return LZFSE_MI_BAD_ARGUMENT;
    switch (option) {
    case LZFSE_MI_OPT_ENCODE_LANES:
    case LZFSE_MI_OPT_DECODE_LANES:
        if (value < 0 || value > LZFSE_MI_MAX_LANES) return LZFSE_MI_BAD_ARGUMENT;
        (option == LZFSE_MI_OPT_ENCODE_LANES ? c->opt_lanes_enc : c->opt_lanes_dec) = (int)value;
        return LZFSE_MI_OK;
    case LZFSE_MI_OPT_STAGGER: c->opt_stagger = value != 0; return LZFSE_MI_OK;
    case LZFSE_MI_OPT_STREAM_SPARE:
        // 0: free what finished stream objects left with the context and keep nothing from now on; 1 (default): keep
        if (value != 0 && value != 1) return LZFSE_MI_BAD_ARGUMENT;
        c->spare.keep = value != 0;
        if (!c->spare.keep) {
            for (auto &b : c->spare.b) b.release();
        }
        return LZFSE_MI_OK;
    case LZFSE_MI_OPT_DECODE_PIPE:
        if (value < 0 || value > 0x1FF || (value > 1 && (value & 0xFF) > 64) || (value > 1 && (value & 0xFF) == 0)) return LZFSE_MI_BAD_ARGUMENT;
        c->opt_pipe = (int)value;
        return LZFSE_MI_OK;
#ifdef LZFSE_MI_DIAG
    case LZFSE_MI_OPT_DIAG_LZ_PATH: c->diag_lz_jump = (int)value; return LZFSE_MI_OK;
    case LZFSE_MI_OPT_DIAG_LZ_TILE: c->diag_lz_variant = (int)value; return LZFSE_MI_OK;
    case LZFSE_MI_OPT_DIAG_STATS: c->diag_stats = (int)value; return LZFSE_MI_OK;
    case LZFSE_MI_OPT_DIAG_CHAIN: c->diag_chain = (int)value; return LZFSE_MI_OK;
    case LZFSE_MI_OPT_DIAG_WALK: c->diag_walk = (int)value; return LZFSE_MI_OK;
    case LZFSE_MI_OPT_DIAG_PIPE_SCATTER: c->diag_pipe_scatter = (int)value; c->pipe_broken = false; return LZFSE_MI_OK;
    case LZFSE_MI_OPT_DIAG_GUIDE: {
        // (the reference's own conditions on the two, frontend_bytes.rs:166-168,359: the limit of a block lies MAX_MATCH_DISTANCE or more into it)
        const uint64_t g = (uint64_t)value & 0xFFFFFFFFull, sl = (uint64_t)value >> 32;
        if (value && (sl < 256 || 2 * sl > g || g > 0x7FFFFFFFull || g - sl - 3 < MAX_D_VALUE)) return LZFSE_MI_BAD_ARGUMENT;
        c->diag_guide = (uint64_t)value;
        return LZFSE_MI_OK;
    }
#else
    case LZFSE_MI_OPT_DIAG_LZ_PATH:
    case LZFSE_MI_OPT_DIAG_LZ_TILE:
    case LZFSE_MI_OPT_DIAG_STATS:
    case LZFSE_MI_OPT_DIAG_CHAIN:
    case LZFSE_MI_OPT_DIAG_WALK:
    case LZFSE_MI_OPT_DIAG_PIPE_SCATTER:
    case LZFSE_MI_OPT_DIAG_GUIDE: return LZFSE_MI_UNSUPPORTED;
#endif
    default: return LZFSE_MI_BAD_ARGUMENT;
    }
}

// ---------------------------------------------------------------------------- host-pointer API

// Host buffers in, host buffers out (what a binding hands over: Vec<u8> / &[u8]). Pageable memory cannot be the end of a
// DMA, so everything passes through pinned staging; one memcpy thread moves ~10 GB/s, less than the link, so the staging
// copies are spread over a few threads (large calls only) and run group by group beside the DMA of the group before /
// after. Encoded streams are a third of their capacity: they are packed on the device before they travel.
struct CopyJob {
    uint8_t *dst;
    const uint8_t *src;
    size_t len;
};

// helper threads that live as long as the context (a thread per granule and call costs as much as the copy it does);
// returns how many of the `want` there are
static unsigned copy_helpers(lzfse_mi_ctx *c, unsigned want) {
    while (c->copy_workers.size() < want) {
        LaneWorker *w = nullptr;
        try { w = new (std::nothrow) LaneWorker(); } catch (...) { w = nullptr; }   // (no thread to be had: fewer helpers)
        if (!w) break;
        c->copy_workers.push_back(w);
    }
    return (unsigned)std::min<size_t>(want, c->copy_workers.size());
}

// A destination that has never been touched -- the Vec<u8> a binding has just made: a large malloc is an mmap -- is faulted in
// page by page by whoever writes it first. The helper threads do that while the kernels run: MADV_POPULATE_WRITE maps the
// pages writable without changing what they hold (Linux 5.14; anything it refuses is simply left to the copy). 256 MiB
// decoded into fresh memory: 25.8 -> 17.3 ms, the time of a buffer used before (scripts/fresh_dst.py, profiles/r04_fresh_dst.txt).
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
struct Populate {
    lzfse_mi_ctx *c = nullptr;
    unsigned busy = 0;
    std::vector<std::pair<uintptr_t, uintptr_t>> spans;   // page-aligned [lo, hi)
    void add(const void *p, uint64_t len) {
        if (len < ((uint64_t)1 << 20)) return;
        const uintptr_t lo = (uintptr_t)p & ~(uintptr_t)4095, hi = ((uintptr_t)p + len + 4095) & ~(uintptr_t)4095;
        spans.push_back({lo, hi});
    }
    void start(lzfse_mi_ctx *ctx) {
        uint64_t total = 0;
        for (auto &s : spans) total += s.second - s.first;
        if (total < ((uint64_t)4 << 20)) return;
        c = ctx;
        const unsigned hc = std::thread::hardware_concurrency();
        unsigned T = std::min<unsigned>(8u, std::max<unsigned>(1u, hc / 2));
        T = (unsigned)std::min<uint64_t>(T, total >> 21);
        busy = copy_helpers(c, T);
        for (unsigned t = 0; t < busy; t++) {
            const uint64_t a = total * t / busy, b = total * (t + 1) / busy;
            c->copy_workers[t]->submit([this, a, b] {
                uint64_t at = 0;
                for (auto &s : spans) {
                    const uint64_t len = s.second - s.first;
                    const uint64_t x = std::max(a, at), y = std::min(b, at + len);
                    if (x < y) {
                        // (whole pages: a page that two threads share is asked for twice, which is harmless)
                        const uintptr_t lo = (s.first + (x - at)) & ~(uintptr_t)4095, hi = (s.first + (y - at) + 4095) & ~(uintptr_t)4095;
                        for (uintptr_t q = lo; q < hi; q += (uintptr_t)8 << 20)
                            if (madvise((void *)q, std::min<uintptr_t>(hi - q, (uintptr_t)8 << 20), MADV_POPULATE_WRITE)) break;
                    }
                    at += len;
                }
            });
        }
    }
    void finish() {
        for (unsigned t = 0; t < busy; t++) c->copy_workers[t]->wait();
        busy = 0;
    }
    ~Populate() { finish(); }
};

// bytes [lo, hi) of the concatenation of the jobs, shared by a few threads
static void par_copy(lzfse_mi_ctx *c, const std::vector<CopyJob> &jobs, const std::vector<uint64_t> &pre, uint64_t lo, uint64_t hi) {
    if (hi <= lo) return;
    const uint64_t total = hi - lo;
    unsigned T = 1;
    if (total >= ((uint64_t)8 << 20)) {
        const unsigned hc = std::thread::hardware_concurrency();
        T = std::min<unsigned>(8u, std::max<unsigned>(1u, hc / 2));
        T = (unsigned)std::min<uint64_t>(T, total >> 21);
    }
    auto work = [&](unsigned t) {
        const uint64_t a = lo + total * t / T, b = lo + total * (t + 1) / T;
        size_t k = (size_t)(std::upper_bound(pre.begin(), pre.end(), a) - pre.begin()) - 1;   // the job that holds byte a
        for (; k < jobs.size() && pre[k] < b; k++) {
            const CopyJob &j = jobs[k];
            const uint64_t x = std::max(a, pre[k]), y = std::min(b, pre[k] + j.len);
            if (x < y) memcpy(j.dst + (x - pre[k]), j.src + (x - pre[k]), (size_t)(y - x));
        }
    };
    T = copy_helpers(c, T - 1) + 1;
    if (T <= 1) { T = 1; work(0); return; }
    for (unsigned t = 1; t < T; t++) c->copy_workers[t - 1]->submit([&work, t] { work(t); });
    work(0);
    for (unsigned t = 1; t < T; t++) c->copy_workers[t - 1]->wait();
}

// The host's side of OutMirror (internal.h): helper threads that watch the streams' words and copy a stream out of the pinned image
// as soon as the device says it is whole.
struct MirrorOut {
    lzfse_mi_ctx *c = nullptr;
    unsigned busy = 0;
    std::atomic<bool> stop{false};
    volatile unsigned long long *done = nullptr;
    std::vector<uint8_t> ok;    // [stream]: copied out of the image (and how long it was)
    std::vector<uint64_t> len;
    void start(lzfse_mi_ctx *ctx, size_t count, const std::vector<uint8_t> &big_out, const std::vector<uint64_t> &dof, uint8_t *const *dsts,
               uint64_t span) {
        if (!ctx->h_out.ensure(span + 256)) return;
        void *dv = nullptr, *dd = nullptr;
        unsigned long long *flags = (unsigned long long *)ctx->h_done.get(count * sizeof(unsigned long long));
        if (!flags || hipHostGetDevicePointer(&dv, ctx->h_out.p, 0) != hipSuccess || hipHostGetDevicePointer(&dd, flags, 0) != hipSuccess) return;
        std::memset(flags, 0, count * sizeof(unsigned long long));
        const unsigned hc = std::thread::hardware_concurrency();
        unsigned T = std::min<unsigned>(8u, std::max<unsigned>(1u, hc / 2));
        T = (unsigned)std::min<size_t>(T, std::max<size_t>(1, count / 4));
        T = copy_helpers(ctx, T);
        if (!T) return;
        c = ctx; done = flags; busy = T;
        ok.assign(count, 0); len.assign(count, 0);
        stop.store(false);
        ctx->mirror.base = (uint8_t *)dv; ctx->mirror.span = span; ctx->mirror.done = (unsigned long long *)dd;
        const uint8_t *img = (const uint8_t *)ctx->h_out.p;
        for (unsigned t = 0; t < T; t++) {
            c->copy_workers[t]->submit([this, t, T, count, &big_out, &dof, dsts, img] {
                std::vector<size_t> mine;
                for (size_t i = t; i < count; i += T) if (!big_out[i]) mine.push_back(i);
                for (;;) {
                    const bool last = stop.load(std::memory_order_acquire);   // (read BEFORE the sweep: after it, every word is final)
                    size_t keep = 0;
                    for (size_t k = 0; k < mine.size(); k++) {
                        const size_t i = mine[k];
                        const unsigned long long w = __atomic_load_n(&done[i], __ATOMIC_ACQUIRE);
                        if (w == 0) { mine[keep++] = i; continue; }
                        if (w != ~0ull) { std::memcpy(dsts[i], img + dof[i], (size_t)(w - 1)); len[i] = w - 1; ok[i] = 1; }
                    }
                    mine.resize(keep);
                    if (mine.empty() || last) return;
                    if (keep) std::this_thread::yield();
                }
            });
        }
    }
    void finish() {
        if (!busy) return;
        stop.store(true, std::memory_order_release);
        for (unsigned t = 0; t < busy; t++) c->copy_workers[t]->wait();
        busy = 0;
        c->mirror = lzmi::OutMirror();
    }
    // the stream's bytes are in the caller's buffer already (all `n` of them)
    bool copied(size_t i, uint64_t n) const { return i < ok.size() && ok[i] && len[i] == n; }
    ~MirrorOut() { finish(); }
};

static constexpr uint64_t HOST_GROUP = (uint64_t)16 << 20;   // staging granule: copied by the threads while the DMA moves the granule before / after

// The staged image of byte `pos` of the concatenation: jobs are in staging order, stage[k] = staging offset of job k.
static uint64_t staged_at(const std::vector<CopyJob> &jobs, const std::vector<uint64_t> &pre, const std::vector<uint64_t> &stage, uint64_t pos) {
    const size_t k = (size_t)(std::upper_bound(pre.begin(), pre.end(), pos) - pre.begin()) - 1;
    (void)jobs;
    return stage[k] + (pos - pre[k]);
}

// Streams of this size and more travel by ONE DMA each straight between the caller's buffer and device memory: the
// runtime's own transfer of pageable memory runs at the link's rate (56 GB/s both ways for 8 MiB .. 1 GiB on this pool,
// pinned or not, scripts/pcie_probe.py -> profiles/r03_pcie_probe.txt), so a staging copy through this library's pinned
// buffers only adds a pass over the bytes. A transfer call costs 16 (in) / 27 (out) microseconds, which is the time of
// half a megabyte: smaller streams are packed into the pinned granules as before (3 072 x 245 KiB moved one by one: 15 / 9 GB/s).
#ifndef HOST_DIRECT_IN
#define HOST_DIRECT_IN ((uint64_t)1 << 20)
#endif
// Outputs below 512 MiB do NOT travel that way (round 4): a transfer straight into pageable memory pins what it lands in, and for a
// destination that is the untouched tail of a large allocation -- Vec::with_capacity(encode_bound) + 31 MB of stream -- that
// took 27 - 38 ms where the staged form takes 10.8 (128 MiB of text encoded; scripts/fresh_dst.py); on a buffer used before the
// two forms are within 7 % of each other either way.
#ifndef HOST_DIRECT_OUT
#define HOST_DIRECT_OUT ((uint64_t)512 << 20)   // (beyond that a second copy of the output in pinned memory is the larger evil)
#endif
// ... and all staged outputs of a call together stay below this (round 5): the pinned staging is kept until the context goes, and a batch of
// 64 streams of 256 MiB would otherwise pin 20 GiB of host memory for good. Outputs beyond the budget travel straight into the caller's
// buffers, as all of them do when the staging cannot be had at all.
#ifndef HOST_MIRROR_MAX
#define HOST_MIRROR_MAX ((uint64_t)80 << 20)
#endif
#ifndef HOST_STAGE_BUDGET
#define HOST_STAGE_BUDGET ((uint64_t)512 << 20)
#endif

static int host_batch_one(lzfse_mi_ctx *c, batch_dev_fn fn, bool pack_outputs, size_t count, const uint8_t *const *srcs,
                          const size_t *lens, uint8_t *const *dsts, const size_t *caps, size_t *out_lens,
                          int *statuses, const std::function<void()> *on_staged) {
    if (!c || (count && (!srcs || !lens || !dsts || !caps || !out_lens || !statuses))) return LZFSE_MI_BAD_ARGUMENT;
    if (count == 0) { if (on_staged) (*on_staged)(); return LZFSE_MI_OK; }
    HIP_TRY(hipSetDevice(c->device));
    std::vector<uint64_t> so(count), sl(count), dof(count), dc(count), ol(count);
    const uint64_t direct_out = c->pinned_out ? (uint64_t)1 << 20 : HOST_DIRECT_OUT;   // (a stream object's window: plain DMA into its pinned buffer)
    // device layout: the staged (small) streams first, in the caller's order, then the direct (large) ones
    uint64_t in_total = 0, out_total = 0, in_staged = 0, out_staged = 0;
    for (int big = 0; big < 2; big++) {
        for (size_t i = 0; i < count; i++)
            if ((lens[i] >= HOST_DIRECT_IN) == (big != 0)) { so[i] = in_total; sl[i] = lens[i]; in_total += (lens[i] + 255) & ~(uint64_t)255; }
        if (!big) in_staged = in_total;
    }
    // which outputs travel by a transfer of their own: by capacity, and whatever would take the staged ones beyond the budget
    // (encoded streams are packed first: there the budget is applied to what they really hold, below)
    std::vector<uint8_t> big_out(count);
    {
        uint64_t acc = 0;
        for (size_t i = 0; i < count; i++) {
            const uint64_t padded = (caps[i] + 255) & ~(uint64_t)255;
            big_out[i] = caps[i] >= direct_out || (!pack_outputs && acc + padded > HOST_STAGE_BUDGET);
            if (!big_out[i]) acc += padded;
        }
    }
    for (int big = 0; big < 2; big++) {
        for (size_t i = 0; i < count; i++)
            if ((big_out[i] != 0) == (big != 0)) { dof[i] = out_total; dc[i] = caps[i]; out_total += (caps[i] + 255) & ~(uint64_t)255; }
        if (!big) out_staged = out_total;
    }
    // (a deferred transfer, see below: this call's device output buffer is the one of the call before last, whose transfer is over)
    const bool defer = c->defer_out && count == 1 && !pack_outputs;
    const int ob = defer ? c->out_flip : 0;
    DevBuf &dout = ob ? c->d_out2 : c->d_out;
    c->defer_last = nullptr;
    if (defer) {
        if (c->defer_pending[ob]) { HIP_TRY(hipEventSynchronize(c->defer_ev[ob])); c->defer_pending[ob] = false; }
        c->out_flip ^= 1;
    } else if (c->defer_pending[0]) { HIP_TRY(hipEventSynchronize(c->defer_ev[0])); c->defer_pending[0] = false; }
    if (!c->d_in.ensure(in_total + 256) || !dout.ensure(out_total + 256) || !c->h_in.ensure(in_staged + 256))
        return LZFSE_MI_IO;
    std::vector<CopyJob> jobs;
    std::vector<uint64_t> pre, stage;   // per job: first byte in the concatenation, offset in the staging buffer
    jobs.reserve(count); pre.reserve(count + 1); stage.reserve(count);
    // ---- in: granules of HOST_GROUP bytes; the DMA of one runs under the staging copy of the next ----
    uint64_t n_in = 0;
    for (size_t i = 0; i < count; i++)
        if (lens[i] && lens[i] < HOST_DIRECT_IN) { jobs.push_back({(uint8_t *)c->h_in.p + so[i], srcs[i], lens[i]}); pre.push_back(n_in); stage.push_back(so[i]); n_in += lens[i]; }
    for (uint64_t lo = 0; lo < n_in; lo += HOST_GROUP) {
        const uint64_t hi = std::min(n_in, lo + HOST_GROUP);
        par_copy(c, jobs, pre, lo, hi);
        const uint64_t a = staged_at(jobs, pre, stage, lo), b = staged_at(jobs, pre, stage, hi - 1) + 1;
        HIP_TRY(hipMemcpyAsync((uint8_t *)c->d_in.p + a, (uint8_t *)c->h_in.p + a, b - a, hipMemcpyHostToDevice, c->stream));
    }
    for (size_t i = 0; i < count; i++)
        if (lens[i] >= HOST_DIRECT_IN) HIP_TRY(hipMemcpyAsync((uint8_t *)c->d_in.p + so[i], srcs[i], lens[i], hipMemcpyHostToDevice, c->stream));
    if (on_staged) (*on_staged)();   // (the inputs are on their way: the other half of a split call may start staging)
    // while the kernels run, the helper threads fault in the large destinations: all of a decoded stream's (its capacity is
    // its size unless the caller was generous: then what the headers say), 3/8 of the input for an encoded one
    Populate pop;
    for (size_t i = 0; i < count; i++) {
        if (caps[i] < ((uint64_t)1 << 20) || c->pinned_out) continue;
        uint64_t want = caps[i];
        if (pack_outputs) want = std::min<uint64_t>(want, (uint64_t)lens[i] / 8 * 3);
        else {
            uint64_t promised = 0;
            (void)lzfse_mi_decode_size(srcs[i], lens[i], &promised);
            want = std::min<uint64_t>(want, promised);
        }
        pop.add(dsts[i], want);
    }
    pop.start(c);
    // Decode (round 5): the LZ stage writes every finished tile of a staged stream into the device buffer AND into the pinned
    // staging image (OutMirror), and a word per stream when it is through; helper threads copy finished streams from the image into
    // the caller's buffers while the rest is still decoded. What the transfer after the kernels cost (94 MB: 1.7 ms at the link's
    // rate, then the copies) now runs under the stage. Streams that go another way on the device (pointer jumping, several
    // workgroups per stream) and failed ones leave no word: they travel as before.
    // (Up to 80 MiB of staged output per call, i.e. per half of a large call: beyond that the halves' pipeline does better with a
    // stage that is over quickly and a copy engine that moves its output under the other half's kernels, than with a stage that
    // holds its wave slots for as long as the link takes -- A/B on one box, 47 / 71 / 94 / 141 / 188 MB: 17.3 / 19 / 21.5 / 21 / 19.7
    // GB/s with the image against 14.2 / 12.5 / 14 / 20.6 / 23 without, scripts/pcie_sample.py.)
    MirrorOut mo;
    if (!pack_outputs && !defer && !c->pinned_out && out_staged && out_staged <= HOST_MIRROR_MAX && c->opt_lanes_dec <= 1 && !pop.busy)
        mo.start(c, count, big_out, dof, dsts, out_staged);
    int r = fn(c, count, c->d_in.p, so.data(), sl.data(), dout.p, dof.data(), dc.data(), ol.data(), statuses);
    mo.finish();
    pop.finish();
    if (r) return r;
    // ---- out: where each stream's bytes are on the device (packed first when they fill little of their capacity) ----
    std::vector<uint64_t> at(count);        // offset of stream i's output in the buffer that travels
    uint64_t produced = 0, hi_off = 0;
    bool fits32 = count <= 0x7FFFFFFFu;
    for (size_t i = 0; i < count; i++) {
        out_lens[i] = statuses[i] == 0 ? (size_t)ol[i] : 0;
        at[i] = dof[i];
        fits32 &= out_lens[i] <= 0xFFFFFFFFull;
        if (out_lens[i]) { produced += out_lens[i]; hi_off = std::max(hi_off, dof[i] + ol[i]); }
    }
    if (!produced) return LZFSE_MI_OK;
    const uint8_t *d_from = (const uint8_t *)dout.p;
    uint64_t staged_span = out_staged;      // the staged outputs lie in [0, staged_span) of the buffer that travels
    const std::vector<uint8_t> big_layout = big_out;   // (by capacity: the layout)
    if (pack_outputs && fits32 && produced + (produced >> 2) < hi_off) {
        uint64_t pk = 0;
        std::vector<SmallDesc> desc;
        std::vector<uint64_t> packed(count, 0);
        desc.reserve(count);
        uint64_t pk_staged = 0;
        // ... or, when the outputs are packed anyway (encoded streams: a third of their capacity), by what they really hold
        {
            uint64_t acc = 0;
            for (size_t i = 0; i < count; i++) {
                const uint64_t padded = (out_lens[i] + 15) & ~(uint64_t)15;
                big_out[i] = out_lens[i] >= direct_out || acc + padded > HOST_STAGE_BUDGET;
                if (!big_out[i]) acc += padded;
            }
        }
        for (int big = 0; big < 2; big++) {   // (staged streams first)
            for (size_t i = 0; i < count; i++) {
                if ((big_out[i] != 0) != (big != 0)) continue;
                packed[i] = pk;
                if (!out_lens[i]) continue;
                desc.push_back({dof[i], pk, (uint32_t)out_lens[i], 0u});
                pk += (out_lens[i] + 15) & ~(uint64_t)15;
            }
            if (!big) pk_staged = pk;
        }
        // (the inputs are dead: the packed copy takes their place, its descriptors behind it)
        const uint64_t desc_at = (pk + 255) & ~(uint64_t)255;
        if (c->d_in.ensure((size_t)desc_at + desc.size() * sizeof(SmallDesc) + 256)) {
            SmallDesc *d_desc = (SmallDesc *)((uint8_t *)c->d_in.p + desc_at);
            HIP_TRY(hipMemcpyAsync(d_desc, desc.data(), desc.size() * sizeof(SmallDesc), hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(small_copy_kernel, dim3((uint32_t)desc.size()), dim3(256), 0, c->stream, (const uint8_t *)dout.p,
                               (uint8_t *)c->d_in.p, (const SmallDesc *)d_desc, (uint32_t)desc.size());
            HIP_TRY(hipStreamSynchronize(c->stream));   // (`desc` is pageable memory of this scope)
            d_from = (const uint8_t *)c->d_in.p;
            at = packed;
            staged_span = pk_staged;
        } else {
            big_out = big_layout;
        }
    } else if (pack_outputs) {
        // (encoded streams that are not packed: the staging would hold their capacities' span)
        uint64_t acc = 0;
        for (size_t i = 0; i < count; i++)
            if (!big_out[i]) {
                const uint64_t padded = (caps[i] + 255) & ~(uint64_t)255;
                if (acc + padded > HOST_STAGE_BUDGET) { staged_span = std::min(staged_span, dof[i]); for (size_t k = i; k < count; k++) if (!big_out[k]) big_out[k] = 1; break; }
                acc += padded;
            }
    }
    if (!c->h_out.ensure(staged_span + 256)) {
        // no pinned staging to be had: every output travels straight into the caller's buffer
        for (size_t i = 0; i < count; i++) big_out[i] = 1;
        staged_span = 0;
    }
    // ---- granules again: every DMA is queued, and a granule is copied to its destinations as soon as it has arrived ----
    jobs.clear(); pre.clear(); stage.clear();
    uint64_t n_out = 0;
    for (size_t i = 0; i < count; i++)
        if (out_lens[i] && !big_out[i] && !mo.copied(i, out_lens[i])) { jobs.push_back({dsts[i], (const uint8_t *)c->h_out.p + at[i], out_lens[i]}); pre.push_back(n_out); stage.push_back(at[i]); n_out += out_lens[i]; }
    const size_t n_gran = (size_t)((n_out + HOST_GROUP - 1) / HOST_GROUP);
    while (c->host_ev.size() < n_gran) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return LZFSE_MI_IO;
        c->host_ev.push_back(e);
    }
    for (size_t g = 0; g < n_gran; g++) {
        const uint64_t lo = g * HOST_GROUP, hi = std::min(n_out, lo + HOST_GROUP);
        const uint64_t a = staged_at(jobs, pre, stage, lo), b = staged_at(jobs, pre, stage, hi - 1) + 1;
        HIP_TRY(hipMemcpyAsync((uint8_t *)c->h_out.p + a, d_from + a, b - a, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipEventRecord(c->host_ev[g], c->stream));
    }
    // the large streams go straight into the caller's buffers (behind the granules on the stream: those are unpacked by the
    // helper threads while these travel)
    bool any_direct = false;
    if (defer && out_lens[0] && big_out[0] && n_gran == 0) {
        // The stream decoder's window in the background: the kernels are through (fn has waited for them). The last bytes of the
        // output -- the next window's history -- come at once; the rest travels on a stream of its own while this call returns
        // and the next window's call uploads and decodes: whoever wants the bytes waits for the event first (ctx_deferred_event).
        if (!c->xfer_stream) {
            if (hipStreamCreateWithFlags(&c->xfer_stream, hipStreamNonBlocking) != hipSuccess) c->xfer_stream = nullptr;
            else
                for (auto &e : c->defer_ev)
                    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
        }
        if (c->xfer_stream && c->defer_ev[ob]) {
            const size_t tail = out_lens[0] < ((size_t)1 << 18) ? out_lens[0] : (size_t)1 << 18, body = out_lens[0] - tail;
            HIP_TRY(hipMemcpyAsync(dsts[0] + body, d_from + at[0] + body, tail, hipMemcpyDeviceToHost, c->stream));
            bool ok = true;
            if (body) {
                HIP_TRY(hipMemcpyAsync(dsts[0], d_from + at[0], body, hipMemcpyDeviceToHost, c->xfer_stream));
                // (from here on a transfer into the caller's buffer is in flight: whatever fails, it is waited for before this
                // call returns an error -- the caller frees or reuses that buffer on one)
                if (hipEventRecord(c->defer_ev[ob], c->xfer_stream) == hipSuccess) {
                    c->defer_pending[ob] = true;
                    c->defer_last = c->defer_ev[ob];
                } else ok = false;
            }
            if (hipStreamSynchronize(c->stream) != hipSuccess) ok = false;
            if (!ok) {
                (void)hipStreamSynchronize(c->xfer_stream);
                c->defer_pending[ob] = false;
                c->defer_last = nullptr;
                return LZFSE_MI_IO;
            }
            return LZFSE_MI_OK;
        }
    }
    for (size_t i = 0; i < count; i++)
        if (out_lens[i] && big_out[i]) {
            HIP_TRY(hipMemcpyAsync(dsts[i], d_from + at[i], out_lens[i], hipMemcpyDeviceToHost, c->stream));
            any_direct = true;
        }
    for (size_t g = 0; g < n_gran; g++) {
        HIP_TRY(hipEventSynchronize(c->host_ev[g]));
        const uint64_t lo = g * HOST_GROUP;
        par_copy(c, jobs, pre, lo, std::min(n_out, lo + HOST_GROUP));
    }
    if (any_direct) HIP_TRY(hipStreamSynchronize(c->stream));
    return LZFSE_MI_OK;
}

// A large call is cut in two halves that run on two contexts, the second one step behind the first: its inputs travel
// while the first half's kernels run, its kernels run while the first half's outputs travel and are copied out. (Not while
// stage timings are collected: they are per context.)
static int host_batch(lzfse_mi_ctx *c, batch_dev_fn fn, bool pack_outputs, size_t count, const uint8_t *const *srcs,
                      const size_t *lens, uint8_t *const *dsts, const size_t *caps, size_t *out_lens,
                      int *statuses) {
    if (!c || (count && (!srcs || !lens || !dsts || !caps || !out_lens || !statuses))) return LZFSE_MI_BAD_ARGUMENT;
    uint64_t bytes = 0;
    for (size_t i = 0; i < count; i++) bytes += (uint64_t)lens[i] + caps[i];
    size_t nA = 0;
    // (measured: 1 GiB of 4 MiB streams +18 % encode, +10 % decode. Round 3 found 94 MB in 384 streams 10 - 25 % SLOWER cut in two,
    // with each half cut once more into its two encode lanes: four sub-batches of 24 MB. Round 4: below 512 MiB the halves ARE the
    // two lanes, each half one pass: 94 MB +8 % encode, +4 % decode (A/B on one box, scripts/pcie_sample.py); 29 MB: no difference)
#ifndef HOST_SPLIT_MIN
#define HOST_SPLIT_MIN ((uint64_t)64 << 20)
#endif
    if (count >= 2 && bytes >= HOST_SPLIT_MIN && !c->timing && !c->is_peer) {
        uint64_t acc = 0;
        while (nA + 1 < count && acc + lens[nA] + caps[nA] <= bytes / 2) { acc += (uint64_t)lens[nA] + caps[nA]; nA++; }
        if (nA == 0) nA = 1;
        if (!c->host_peer) {
            if (lzfse_mi_create(c->device, &c->host_peer) != LZFSE_MI_OK) c->host_peer = nullptr;
            else c->host_peer->is_peer = true;
        }
        if (c->host_peer && !c->host_worker) {
            try { c->host_worker = new (std::nothrow) LaneWorker(); } catch (...) { c->host_worker = nullptr; }
        }
        if (!c->host_peer || !c->host_worker) nA = 0;
    }
    if (nA == 0 || nA >= count) {
        const int r = host_batch_one(c, fn, pack_outputs, count, srcs, lens, dsts, caps, out_lens, statuses, nullptr);
        return r;
    }
    lzfse_mi_ctx *p = c->host_peer;
    p->opt_lanes_enc = c->opt_lanes_enc; p->opt_lanes_dec = c->opt_lanes_dec; p->opt_stagger = c->opt_stagger; p->opt_pipe = c->opt_pipe;
    // (a call below 512 MiB that is cut in two: the halves ARE the two sub-batches, each runs as one pass)
    const int keep_lanes = c->opt_lanes_enc;
    if (bytes < ((uint64_t)512 << 20) && !c->opt_lanes_enc) { c->opt_lanes_enc = 1; p->opt_lanes_enc = 1; }
    p->diag_lz_jump = c->diag_lz_jump; p->diag_lz_variant = c->diag_lz_variant; p->diag_stats = c->diag_stats; p->diag_chain = c->diag_chain;
    p->diag_walk = c->diag_walk; p->diag_pipe_scatter = c->diag_pipe_scatter;
    p->parse_ring = c->parse_ring;
    int rB = 0;
    bool started = false;
    const size_t nB = count - nA;
    const std::function<void()> kick = [&] {
        started = true;
        c->host_worker->submit([&] { rB = host_batch_one(p, fn, pack_outputs, nB, srcs + nA, lens + nA, dsts + nA, caps + nA, out_lens + nA, statuses + nA, nullptr); });
    };
    const int rA = host_batch_one(c, fn, pack_outputs, nA, srcs, lens, dsts, caps, out_lens, statuses, &kick);
    if (!started) kick();   // (the first half failed before its inputs were staged: the second still has to give its answers)
    c->host_worker->wait();
    c->opt_lanes_enc = keep_lanes;
    // the detail words of both halves, in the caller's order
    std::vector<uint32_t> det(count, 0u);
    for (size_t i = 0; i < nA && i < c->detail_out.size(); i++) det[i] = c->detail_out[i];
    for (size_t i = 0; i < nB && i < p->detail_out.size(); i++) det[nA + i] = p->detail_out[i];
    c->detail_out.swap(det);
    return rA ? rA : rB;
}

int lzfse_mi_decode_batch(lzfse_mi_ctx *c, size_t count, const uint8_t *const *srcs, const size_t *lens,
                          uint8_t *const *dsts, const size_t *caps, size_t *out_lens, int *statuses) {
    return host_batch(c, lzfse_mi_decode_batch_device, false, count, srcs, lens, dsts, caps, out_lens, statuses);
}

int lzfse_mi_encode_batch(lzfse_mi_ctx *c, size_t count, const uint8_t *const *srcs, const size_t *lens,
                          uint8_t *const *dsts, const size_t *caps, size_t *out_lens, int *statuses) {
    return host_batch(c, lzfse_mi_encode_batch_device, true, count, srcs, lens, dsts, caps, out_lens, statuses);
}

int lzfse_mi_last_error_detail(lzfse_mi_ctx *c, size_t stream_index, uint32_t *detail) {
    if (!c || !detail) return LZFSE_MI_BAD_ARGUMENT;
    *detail = stream_index < c->detail_out.size() ? c->detail_out[stream_index] : 0u;
    return LZFSE_MI_OK;
}

int lzfse_mi_decode(lzfse_mi_ctx *c, const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_len) {
    if (!out_len) return LZFSE_MI_BAD_ARGUMENT;
    int st = 0;
    int r = lzfse_mi_decode_batch(c, 1, &src, &n, &dst, &cap, out_len, &st);
    return r ? r : st;
}

// A slice of more than BLOCK_GUIDE + 3 bytes (E20; frontend_bytes.rs:160-211 match_any, :348-375 reposition; test/src/big_mem.rs encodes
// 0x8000_0003 .. 0x2_0000_0000 bytes this way). The reference matches it in blocks of BLOCK_GUIDE bytes: positions below a block's limit
// (BLOCK_GUIDE - SLACK - 3) are visited, matches may run to the block's end, then the source slice is moved up to
// MAX_MATCH_DISTANCE below where the walk stands and the next block begins. The device takes ONE BLOCK PER CALL; between the calls
// this function does what reposition does to the walk's state, and carries what the FSE back end had not written yet:
//   * the walk's literal index and pending match go into the next call as its start state (EncStream::start);
//   * the positions the reference never pushed into its history -- those between the position the block's last match was found at
//     and the limit, when that match ran past the limit -- are kept out of the next call's chains (EncTile::skip_lo);
//   * the bvx2 block the back end was filling when the walk ended is not emitted: its events lead the next call's match list, and
//     that call's stream begins at that block's first raw byte (at most a block's worth, 27 MB, below the reference's own slice), so
//     that its literals are where the entropy stage looks for them. The complete blocks are final: their bytes stay in `dst`.
static int encode_slice_blocks(lzfse_mi_ctx *c, const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_len, uint32_t guide, uint32_t slack) {
    uint64_t base = 0, out = 0, lit = 0, p_idx = 0, p_midx = 0, raw0 = 0, sk_lo = 0, sk_hi = 0;
    uint32_t p_len = 0, carry_skip = 0;
    std::vector<lzmi::RepoEvent> carry;
    std::vector<uint64_t> carry_pos;
    bool first = true;
    try {
        for (;;) {
            const uint64_t remaining = n - base;
            const bool is_short = remaining <= (uint64_t)guide + 3;
            const uint64_t block_len = is_short ? remaining : guide;
            const uint64_t w0 = first ? 0 : std::min(base, raw0);
            if (base - w0 > ((uint64_t)64 << 20) || lit < w0) return LZFSE_MI_IO;   // (a bvx2 block spans 27 MB at most)
            lzmi::RepoWindow w;
            w.first = first; w.final = is_short;
            w.rel0 = (uint32_t)(base - w0);
            w.stop = is_short ? 0u : (uint32_t)(base - w0 + block_len - slack - 3);
            w.st_lit = (uint32_t)(lit - w0);
            w.st_plen = p_len; w.st_pidx = p_len ? (uint32_t)(p_idx - w0) : 0u; w.st_pmidx = p_len ? (uint32_t)(p_midx - w0) : 0u;
            w.st_skip = carry_skip; w.st_raw = (uint32_t)(raw0 - w0);
            if (sk_hi > sk_lo && sk_hi > w0) { w.skip_lo = (uint32_t)(std::max(sk_lo, w0) - w0); w.skip_hi = (uint32_t)(sk_hi - w0); }
            for (size_t k = 0; k < carry.size(); k++) {
                if (carry_pos[k] < w0) return LZFSE_MI_IO;
                carry[k].lit_pos = (uint32_t)(carry_pos[k] - w0);
            }
            w.carry = carry.data(); w.n_carry = (uint32_t)carry.size();
            const uint8_t *ws = src + w0;
            uint8_t *wd = dst + out;
            size_t n_win = (size_t)(base + block_len - w0), wcap = cap - (size_t)out, got = 0;
            int st = 0;
            c->repo = &w;
            const int r = lzfse_mi_encode_batch(c, 1, &ws, &n_win, &wd, &wcap, &got, &st);
            c->repo = nullptr;
            if (r || st) return r ? r : st;
            out += got;
            if (is_short) break;
            // ---- reposition (:348-375) ----
            lit = w0 + w.e_lit;
            p_len = w.e_plen; p_idx = w0 + w.e_pidx; p_midx = w0 + w.e_pmidx;
            const uint64_t limit = base + block_len - slack - 3;
            const uint64_t idx_end = std::max(limit, lit);   // self.index after sync_history (:357)
            sk_lo = sk_hi = 0;
            if (w.e_cross && w0 + w.e_cross < limit) { sk_lo = w0 + w.e_cross; sk_hi = limit; }
            carry = w.left;
            if (!carry.empty() && w.left_skip) {
                // (the front of the unclosed block's first event lies in blocks that are written: the event is carried without it)
                lzmi::RepoEvent &f = carry[0];
                const uint32_t sk = w.left_skip, sl = sk < f.l ? sk : f.l;
                f.lit_pos += sk; f.l -= sl; f.m -= sk - sl;
            }
            carry_pos.resize(carry.size());
            for (size_t k = 0; k < carry.size(); k++) carry_pos[k] = w0 + carry[k].lit_pos;
            carry_skip = 0;
            raw0 = w0 + w.left_raw;
            base = idx_end - MAX_D_VALUE;   // delta = self.index - MAX_MATCH_DISTANCE (:359); the literals that have passed it left with this call
            first = false;
        }
    } catch (...) { c->repo = nullptr; return LZFSE_MI_IO; }
    *out_len = (size_t)out;
    return LZFSE_MI_OK;
}

int lzfse_mi_encode(lzfse_mi_ctx *c, const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_len) {
    if (!out_len) return LZFSE_MI_BAD_ARGUMENT;
    if (c && src && dst && !c->parse_ring) {
        uint32_t guide, slack;
        lzmi::ctx_diag_guide(c, guide, slack);
        if (n > (size_t)guide + 3) return encode_slice_blocks(c, src, n, dst, cap, out_len, guide, slack);
    }
    int st = 0;
    int r = lzfse_mi_encode_batch(c, 1, &src, &n, &dst, &cap, out_len, &st);
    return r ? r : st;
}

// ---- the ring / stream encoder's parse (LzfseRingEncoder::encode, LzfseWriter: encode/frontend_ring.rs) ----
namespace {
struct RingScope {   // the encode entry points below run the ordinary pipeline with the context marked for the ring parse
    lzfse_mi_ctx *c;
    explicit RingScope(lzfse_mi_ctx *ctx) : c(ctx) { if (c) c->parse_ring = true; }
    ~RingScope() { if (c) c->parse_ring = false; }
};
}  // namespace

int lzfse_mi_encode_ring_batch_device(lzfse_mi_ctx *c, size_t count, const void *d_src, const uint64_t *src_off,
                                      const uint64_t *src_len, void *d_dst, const uint64_t *dst_off,
                                      const uint64_t *dst_cap, uint64_t *out_lens, int *statuses) {
    RingScope ring(c);
    return lzfse_mi_encode_batch_device(c, count, d_src, src_off, src_len, d_dst, dst_off, dst_cap, out_lens, statuses);
}

int lzfse_mi_encode_ring_batch(lzfse_mi_ctx *c, size_t count, const uint8_t *const *srcs, const size_t *lens,
                               uint8_t *const *dsts, const size_t *caps, size_t *out_lens, int *statuses) {
    RingScope ring(c);
    return lzfse_mi_encode_batch(c, count, srcs, lens, dsts, caps, out_lens, statuses);
}

int lzfse_mi_encode_ring(lzfse_mi_ctx *c, const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_len) {
    RingScope ring(c);
    return lzfse_mi_encode(c, src, n, dst, cap, out_len);
}

}  // extern "C"
