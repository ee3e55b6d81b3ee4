// Internal (non-ABI) declarations shared by api.hip, decode.hip and encode.hip.
#pragma once
#include "common.h"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <condition_variable>
#include <mutex>
#include <vector>

struct lzfse_mi_ctx;

namespace lzmi {

// One helper thread per shadow context, alive as long as the context: a split batch call hands it a sub-batch and
// waits for it, instead of creating and joining threads per call.
class LaneWorker {
  public:
    LaneWorker() : th_([this] { loop(); }) {}
    ~LaneWorker() {
        { std::lock_guard<std::mutex> g(m_); quit_ = true; }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
    }
    void submit(std::function<void()> job) {
        { std::lock_guard<std::mutex> g(m_); job_ = std::move(job); busy_ = true; }
        cv_.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return !busy_; });
    }
  private:
    void loop() {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return quit_ || (busy_ && job_); });
                if (quit_) return;
                job = std::move(job_);
                job_ = nullptr;
            }
            job();
            { std::lock_guard<std::mutex> g(m_); busy_ = false; }
            cv_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::function<void()> job_;
    bool busy_ = false, quit_ = false;
    std::thread th_;   // last member: the thread starts when everything above exists
};

// brackets kernel launches with HIP events on the context's stream (lzfse_mi_get_timings)
struct StageTimer {
    lzfse_mi_ctx *ctx;
    int idx = -1;
    const char *stage;
    StageTimer(lzfse_mi_ctx *c, const char *name);
    ~StageTimer();
};
hipStream_t ctx_stream(lzfse_mi_ctx *c);

// Hand-over point between two lanes of a split encode call: the later lane starts its kernels when the earlier one
// has queued its match-candidate kernel, so that one lane's latency-bound stages run under the other lane's
// throughput-bound stage instead of both lanes doing the same thing at the same time.
struct LaneGate {
    hipEvent_t ev = nullptr;
    std::mutex m;
    std::condition_variable cv;
    int state = 0;  // 0: pending, 1: ev recorded, 2: released without an event
    void arm() { std::lock_guard<std::mutex> g(m); state = 0; }
    void open(int s) { { std::lock_guard<std::mutex> g(m); if (state == 0) state = s; } cv.notify_all(); }
    int wait() {  // bounded: a lane that never gets there (an error path) must not hang its successor
        std::unique_lock<std::mutex> g(m);
        cv.wait_for(g, std::chrono::seconds(2), [&] { return state != 0; });
        return state;
    }
};
LaneGate *ctx_gate_in(lzfse_mi_ctx *c);
LaneGate *ctx_gate_out(lzfse_mi_ctx *c);
int ctx_diag_stats(lzfse_mi_ctx *c);  // LZFSE_MI_OPT_DIAG_STATS bits: 1 block encode, 2 parse, 4 LZ decode
bool ctx_parse_ring(lzfse_mi_ctx *c);  // this call encodes with the ring / stream encoder's parse (lzfse_mi_encode_ring*, lzfse_mi_estream_*)
// One window of a longer stream (the stream encoder, stream.hip): set on the context for the duration of ONE single-stream
// lzfse_mi_encode_ring call. The stream's bytes are those of the window, positions relative to its first byte.
struct EncWindow {
    bool start = false;       // the parse goes on from `st` (the window before was cut there) instead of (0, 0, nothing pending)
    bool beyond = false;      // the window's first byte is not the stream's: a multiple of 16 KiB beyond the first ring (RING_CONT, enc_common.h)
    bool final = false;       // the input ends with this window: everything is emitted, bvx$ included
    uint32_t st[5] = {};      // index, literal_index, pending (idx, match idx, len)
    uint32_t skip = 0;        // bytes of the first event the parse will make from `st` that are in the stream already
    // out (final = false): where the window was cut (as `st` and `skip` of the next one); found = 0: no block of it is final yet
    uint32_t found = 0, index = 0, lit = 0, p_idx = 0, p_midx = 0, p_len = 0, skip_out = 0;
};
EncWindow *ctx_window(lzfse_mi_ctx *c);
// One block of the reference's slice front end beyond what a single call takes (frontend_bytes.rs:160-211,348-375; a slice of more
// than BLOCK_GUIDE + 3 bytes): what the device call over it is told and what it tells (encode_slice_blocks in api.hip drives the
// blocks, enc_batch_device fills EncStream from this). Positions are relative to the call's stream, which begins `rel0` bytes before
// the block: at the first raw byte of the bvx2 block the front end had not closed when the block before ended.
struct RepoEvent { uint32_t lit_pos, l, m, d; };   // (MatchRec, enc_common.h)
struct RepoWindow {
    // in
    bool first = true;            // the slice's first block: the parse starts at (0, 0, nothing pending)
    bool final = false;           // its last block: flush_pending, flush_literals, bvx$
    uint32_t rel0 = 0;            // the block's first byte (self.src of the reference)
    uint32_t stop = 0;            // the block's limit (not the last block)
    uint32_t st_lit = 0, st_pidx = 0, st_pmidx = 0, st_plen = 0;   // literal index and pending match at the block's first visited position, rel0 + MAX_MATCH_DISTANCE
    uint32_t st_skip = 0, st_raw = 0;   // bytes of the first carried event that have left already; first raw byte of the first bvx2 block of this call
    uint32_t skip_lo = 0, skip_hi = 0;  // positions the reference never pushed into its history
    const RepoEvent *carry = nullptr;   // events of the unclosed bvx2 block
    uint32_t n_carry = 0;
    // out (not the last block)
    uint32_t e_lit = 0, e_pidx = 0, e_pmidx = 0, e_plen = 0, e_cross = 0;   // EncStreamOut::e_*
    uint64_t final_bytes = 0;     // bytes of the bvx2 blocks that are complete
    uint32_t left_skip = 0, left_raw = 0;   // the unclosed block: bytes of its first event that lie before it, its first raw byte
    std::vector<RepoEvent> left;  // ... and its events
};
RepoWindow *ctx_repo(lzfse_mi_ctx *c);
void ctx_diag_guide(lzfse_mi_ctx *c, uint32_t &guide, uint32_t &slack);   // BLOCK_GUIDE, SLACK (frontend_bytes.rs:19-23); the diagnostic build can set small ones
// A growable byte buffer in PINNED host memory (hipHostMalloc): what the stream objects (stream.hip) keep their windows in, so
// that a window travels by plain DMA in both directions -- a transfer from or to pageable memory is pinned page by page by the
// runtime first, every time (64 MiB windows: stream decode 10.0 -> see profiles/r04_stream_bench.txt).
struct PinBuf {
    uint8_t *p = nullptr;
    size_t size = 0, cap = 0;
    PinBuf() = default;
    PinBuf(const PinBuf &) = delete;
    PinBuf &operator=(const PinBuf &) = delete;
    ~PinBuf() { release(); }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; size = cap = 0; }
    void swap(PinBuf &o) { std::swap(p, o.p); std::swap(size, o.size); std::swap(cap, o.cap); }
    // room for n bytes; keep: the bytes held stay (false: they may be dropped -- no copy into the new room)
    bool reserve(size_t n, bool keep = true) {
        if (n <= cap) return true;
        void *q = nullptr;
        if (hipHostMalloc(&q, n, hipHostMallocDefault) != hipSuccess) return false;
        if (keep && size) std::memcpy(q, p, size);
        if (!keep) size = 0;
        if (p) (void)hipHostFree(p);
        p = (uint8_t *)q; cap = n;
        return true;
    }
    bool append(const uint8_t *src, size_t n) {
        if (!n) return true;
        if (size + n > cap && !reserve(std::max<size_t>(std::max(size + n, 2 * cap), (size_t)256 << 10))) return false;
        std::memcpy(p + size, src, n);
        size += n;
        return true;
    }
    void erase_front(size_t k) { std::memmove(p, p + k, size - k); size -= k; }
};
// Host buffers of the stream objects that outlive them: a stream object's window buffers are tens of MiB, and pinned pages
// cost more to get than the window's decode -- the next stream object of the context takes over what the last one left
// (0: decoder input, 1: decoder output of a window decoded in the call, 2: encoder input, 3: encoder output, 4: the copy of a
// window decoded in the background, 5 and 6: its two output buffers).
struct StreamSpare {
    PinBuf b[7];
    bool keep = true;   // LZFSE_MI_OPT_STREAM_SPARE
};
StreamSpare &ctx_spare(lzfse_mi_ctx *c);
// A stream object names its context through ONE pointer, registered here: lzfse_mi_destroy of the context sets the pointers of
// the stream objects still alive to null, so that a stream object destroyed (or fed) AFTER its context neither hands buffers
// to freed memory nor runs a call on it (include/lzfse_mi.h: either order of destruction is allowed).
void ctx_attach(lzfse_mi_ctx *c, lzfse_mi_ctx **ref);
void ctx_detach(lzfse_mi_ctx *c, lzfse_mi_ctx **ref);
// The context a stream object's windows run on when they run in the BACKGROUND (a window of the stream encoder is encoded by a
// helper thread while the caller goes on feeding): a context of its own beside the caller's, made when first needed and shared by
// the stream objects of that context, one window at a time (`m`). The box outlives whichever of context and stream objects goes
// first: lzfse_mi_destroy takes `m`, destroys `peer` and marks the box dead.
struct StreamBox {
    std::mutex m;
    lzfse_mi_ctx *peer = nullptr;
    int device = 0;
    bool dead = false;
};
std::shared_ptr<StreamBox> ctx_stream_box(lzfse_mi_ctx *c);
lzfse_mi_ctx *box_peer(StreamBox &b);   // under b.m: the peer context (created here), null when dead or not to be had
void ctx_set_window(lzfse_mi_ctx *c, EncWindow *w);
// the destination of the host-pointer call that follows is pinned memory: its output travels straight there (stream.hip)
void ctx_set_pinned_out(lzfse_mi_ctx *c, bool on);
// ... and may be left to travel while the call returns (a single decoded stream; all but its last 256 KiB): what
// ctx_deferred_event gives after the call (a hipEvent_t, or null) is to be waited for before the bytes are read
void ctx_set_defer_out(lzfse_mi_ctx *c, bool on);
void *ctx_deferred_event(lzfse_mi_ctx *c);
int ctx_diag_chain(lzfse_mi_ctx *c);  // LZFSE_MI_OPT_DIAG_CHAIN: 1 = every chain tile through the ballot kernel

// ---- decode.hip ----
void launch_dec_walk(bool emit, const uint8_t *src, const StreamIn *streams, uint32_t n_streams,
                     StreamWalk *walk, const StreamPlan *plan, BlockDesc *blocks, const uint32_t *settled, hipStream_t st);
void launch_dec_fills(void *const *ptrs, const uint64_t *bytes, const uint32_t *values, int count, hipStream_t st);
uint32_t fastwalk_cap();
void launch_dec_fastwalk(const uint8_t *src, const StreamIn *streams, const uint32_t *elig, uint32_t n_elig, uint64_t max_len,
                         uint32_t *count, uint2 *cand, StreamWalk *walk, BlockDesc *cache, uint32_t *settled, hipStream_t st);
void launch_dec_emit(const StreamIn *streams, uint32_t n_streams, const StreamPlan *plan, const BlockDesc *cache, uint64_t cache_total,
                     BlockDesc *blocks, hipStream_t st);
// jf != null: every stream of the call takes the pointer-jumping LZ path, and the entropy kernel's workgroups do that path's first
// step (literals out, origins initialised) for their own block behind their entropy stage; launch_dec_jump is told (init_done)
struct JumpFuse {
    const StreamIn *streams;
    const StreamPlan *plan;
    uint8_t *dst;
    uint32_t *origin, *jerr;
    uint64_t total;
};
void launch_dec_fse(const uint8_t *src, uint64_t src_total, const BlockDesc *blocks, uint32_t n_blocks,
                    uint8_t *lit_out, LmdRec *lmd_out, BlockResult *results, uint32_t *order_hist, uint32_t *order,
                    const JumpFuse *jf, hipStream_t st);
// Where dec_lz_kernel stores a second copy of what it writes (the host-pointer decode calls): the pinned image of the output buffer
// (streams whose capacity ends below `span` have one, at their offset), and a word per stream for "its bytes are there".
struct OutMirror {
    uint8_t *base = nullptr;
    uint64_t span = 0;
    unsigned long long *done = nullptr;   // [stream]: 0 until the stream is through; then 1 + its length, or all ones (failed)
};
void launch_dec_lz(int variant, const uint8_t *src, const StreamIn *streams, const StreamPlan *plan,
                   uint32_t n_streams, const BlockDesc *blocks, const BlockResult *bres, const LmdRec *lmds,
                   const uint8_t *lits, uint8_t *dst, StreamResult *sres, const OutMirror &mirror, hipStream_t st);

// dec_lzp_kernel's words per stream, one 128-byte line per KIND of access: [LZP_NEXT] next ticket and [LZP_HOME] home XCC + 1
// (device-scope atomics), [LZP_DONE] the published ticket (plain stores and L2-scope loads inside one XCD: that line is dirty
// in that XCD's L2), [LZP_BAD] the word a workgroup on ANOTHER XCD raises through memory (nobody near writes its line, so
// device-scope loads of it are not served from a dirty local copy), [LZP_DIAG ..] eight u64 cycle counters.
constexpr uint32_t LZP_STATE_WORDS = 160, LZP_NEXT = 0, LZP_HOME = 1, LZP_DONE = 32, LZP_BAD = 64, LZP_DIAG = 96;
// (diagnostic build) [LZP_SUMS + 4 * (ticket & 1) ..]: checksum, first position (low 32 bits) and length of what the ticket wrote
constexpr uint32_t LZP_SUMS = 128;
void launch_dec_lzp(int variant, uint32_t K, int lpt, const uint8_t *src, const StreamIn *streams, const StreamPlan *plan,
                    const uint32_t *mlist, uint32_t n_multi, const BlockDesc *blocks, uint32_t n_blocks, const BlockResult *bres,
                    const LmdRec *lmds, const uint8_t *lits, uint2 *ck, uint8_t *dst, StreamResult *sres, uint32_t *state,
                    uint32_t scatter, hipStream_t st);
void launch_dec_lzp_selftest(uint32_t *buf, uint32_t *out, hipStream_t st);   // buf: 1088 zeroed dwords, out: 2 zeroed dwords
void launch_dec_jump(const uint8_t *src, const StreamIn *streams, const StreamPlan *plan, const StreamWalk *walk, uint32_t n_streams,
                     const BlockDesc *blocks, uint32_t n_blocks, const BlockResult *bres, const LmdRec *lmds, const uint8_t *lits,
                     uint8_t *dst, uint32_t *origin, uint64_t total, uint32_t *jerr, uint32_t *flags, StreamResult *sres,
                     lzfse_mi_ctx *c, bool init_done, hipStream_t st);

// A host array that travels to or from the device in every call (stream descriptors, plans, per-stream results): pinned and kept
// with the context. Round 5: these were std::vectors, i.e. pageable memory, and a transfer of pageable memory is neither
// asynchronous -- the call returns when every kernel queued before it is through -- nor cheap beyond a size the runtime
// keeps to itself: the 221 KB of walk results of a 4 608-stream decode call took 14.5 ms to arrive where the 196 KB of a
// 4 080-stream call took 0.06 (profiles/r05_host_phases.txt).
struct PinVec {
    void *p = nullptr;
    size_t cap = 0;
    void *get(size_t bytes) {   // nullptr: no pinned memory to be had (the caller falls back to a vector)
        if (bytes <= cap) return p;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        const size_t nc = bytes + bytes / 2 + 4096;
        if (hipHostMalloc(&p, nc, hipHostMallocDefault) != hipSuccess) { p = nullptr; return nullptr; }
        cap = nc;
        return p;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};
template <class T> struct CtlArray {
    T *p = nullptr;
    size_t n = 0;
    std::vector<T> fallback;
    CtlArray(PinVec &pv, size_t count) : n(count) {
        p = (T *)pv.get((count ? count : 1) * sizeof(T));
        if (!p) { fallback.resize(count ? count : 1); p = fallback.data(); }
    }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    T *data() { return p; }
    size_t size() const { return n; }
};

// ---- encode.hip ----
struct EncScratch {
    void *bufs[32] = {};
    size_t caps[32] = {};
    PinVec host[8];   // control arrays of a call (EH_*)
};
EncScratch &ctx_enc(lzfse_mi_ctx *c);
void enc_scratch_release(EncScratch &s);
int enc_batch_device(lzfse_mi_ctx *c, uint32_t count, const uint8_t *d_src, const uint64_t *src_off,
                     const uint64_t *src_len, uint8_t *d_dst, const uint64_t *dst_off, const uint64_t *dst_cap,
                     uint64_t *out_lens, int *statuses);

// host_small.cpp: the host-side size classes (<= 4096 bytes); ring = as LzfseRingEncoder::encode / LzfseWriter emit them
int encode_small(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_len, bool ring);

}  // namespace lzmi
