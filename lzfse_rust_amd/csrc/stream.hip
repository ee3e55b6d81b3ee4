// Streaming decode surface (decode half): what LzfseRingDecoder::decode(reader, writer) does
// (decode/ring_decoder.rs:58-68) -- the same block loop as the slice path (decode/decoder.rs:73-99: blocks until bvx$,
// which must be the last 4 bytes of the input) over a source that arrives in pieces and a sink that takes the output in
// pieces. Host code over the public entry points: the input is cut at block boundaries, every window of complete blocks
// (LZFSE_MI_STREAM_WINDOW = 64 MiB of raw bytes unless the caller says otherwise) is decoded as one stream on the device, and the 262 139 bytes a match may
// reach back (fse/constants.rs:42) travel with it as a leading raw block (bvx-), so no kernel knows about windows. The
// first error in stream order is reported with the slice path's code: the windows before it decoded cleanly, and the
// window that holds it is decoded by the same kernels.
//
// The encode half of the rank is at the end of this file (lzfse_mi_estream_*): the ring front end's parse
// (encode/frontend_ring.rs) is done by the device kernels (encode_parse.hip, st_ring_find and the round ends of the
// stitcher), a window of input at a time; this file collects the pieces, cuts the windows and hands the stream out.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "common.h"
#include "internal.h"

using namespace lzmi;

struct lzfse_mi_dstream {
    lzfse_mi_ctx *ctx = nullptr;
    size_t window = 0;
    PinBuf in;                     // bytes fed; in[in_pos..] are not yet decoded (pinned memory, like `dst`: a window is two plain DMAs)
    size_t in_pos = 0;
    size_t scan_span = 0;          // the blocks in[in_pos .. in_pos + scan_span) are known to be complete (the scan goes on from there) ...
    uint64_t scan_raw = 0;         // ... and hold this many raw bytes
    std::vector<uint8_t> hist;     // the last <= MAX_D_VALUE bytes of output
    std::vector<uint8_t> tmp_src;
    PinBuf dst;                    // one window's output ...
    uint8_t *far_dst = nullptr;    // ... unless a (damaged) header promises more than a GiB: malloc'd, never touched beyond what is written
    size_t far_cap = 0;
    uint8_t *out_p() const { return far_dst ? far_dst : dst.p; }
    // A window that is full while more input is to come is decoded in the BACKGROUND (a helper thread, the stream objects' own
    // context: StreamBox), from a copy of its blocks (`wsrc`: history block, blocks, bvx$) into the other of two output buffers,
    // while the caller feeds the next window and takes the output of the one before: it is handed to `write` -- on the caller's
    // thread, in stream order -- right after the next window has been sent off (dstream_advance).
    PinBuf wsrc, adst[2];
    int cur = 0;                   // adst[cur] takes the next window
    std::shared_ptr<StreamBox> box;
    LaneWorker *worker = nullptr;
    bool pending = false;
    int pend_st = 0, pend_buf = 0;
    size_t pend_got = 0, pend_nh = 0, pend_len = 0;
    uint64_t pend_cap = 0;
    void *pend_ev = nullptr;       // the window's bytes are still on their way (all but the last 256 KiB): wait for this hipEvent_t first
    size_t reserved = 0;           // bytes lzfse_mi_dstream_reserve has promised behind in.size
    // (the helper copies the window's blocks out of `in` itself: until it has, `in` must neither move nor lose its front)
    std::atomic<int> copied{1};
    size_t pend_from = 0, pend_span = 0;
    void wait_copied() { while (!copied.load(std::memory_order_acquire)) std::this_thread::yield(); }
    void wait_idle() { if (worker && pending) worker->wait(); }
    ~lzfse_mi_dstream() {
        // (what this object's windows grew stays with the context for the next stream object -- if the context is still there
        // and keeps such buffers: ctx is null once lzfse_mi_destroy has run, LZFSE_MI_OPT_STREAM_SPARE)
        wait_idle();
        // (a window's bytes may still be travelling into adst[]: not to be freed under them. Once the context is gone -- ctx is null --
        // so is the stream they travelled on, which lzfse_mi_destroy has waited for)
        if (ctx && pend_ev) (void)hipEventSynchronize((hipEvent_t)pend_ev);
        delete worker;
        std::free(far_dst);
        if (!ctx) return;
        ctx_detach(ctx, &ctx);
        StreamSpare &sp = ctx_spare(ctx);
        if (sp.keep && in.cap > sp.b[0].cap) { in.size = 0; sp.b[0].swap(in); }
        if (sp.keep && dst.cap > sp.b[1].cap) { dst.size = 0; sp.b[1].swap(dst); }
        if (sp.keep && wsrc.cap > sp.b[4].cap) { wsrc.size = 0; sp.b[4].swap(wsrc); }
        for (int k = 0; k < 2; k++)
            if (sp.keep && adst[k].cap > sp.b[5 + k].cap) { adst[k].size = 0; sp.b[5 + k].swap(adst[k]); }
    }
    uint64_t total_in = 0, total_out = 0;
    int status = 0;                // sticky
    bool eos_seen = false;         // bvx$ consumed: any further byte is PayloadOverflow (decoder.rs:93-95)
};

extern "C" LZFSE_MI_API size_t lzfse_mi_decode_headroom(const uint8_t *src, size_t n);

namespace {

// extent of the block at p: 0 = known (len, n_raw, eos), 1 = more input needed, 2 = cannot be told here (damaged or
// unknown: the device decides, with the slice path's error)
int block_extent(const uint8_t *p, size_t avail, uint64_t &len, uint64_t &n_raw, bool &eos) {
    eos = false; len = 0; n_raw = 0;
    if (avail < 4) return 1;
    const uint32_t magic = ld_u32(p);
    if (magic == MAGIC_EOS) { eos = true; len = 4; return 0; }
    if (magic == MAGIC_VX2) {
        if (avail < V2_HEADER_SIZE) return 1;
        FseHeader h;
        if (fse_load_v2(p, h)) return 2;
        len = (uint64_t)h.hdr_size + h.lit_payload + h.lmd_payload; n_raw = h.n_raw;
        return avail < len ? 1 : 0;
    }
    if (magic == MAGIC_VX1) {
        if (avail < V1_HEADER_SIZE) return 1;
        FseHeader h;
        if (fse_load_v1(p, h)) return 2;
        len = (uint64_t)h.hdr_size + h.lit_payload + h.lmd_payload; n_raw = h.n_raw;
        return avail < len ? 1 : 0;
    }
    if (magic == MAGIC_VXN) {
        if (avail < 12) return 1;
        n_raw = ld_u32(p + 4); len = 12ull + ld_u32(p + 8);
        if (n_raw > 136ull * (len - 12)) n_raw = 136ull * (len - 12);   // (as lzfse_mi_decode_size: what a sound block can yield at most)
        return avail < len ? 1 : 0;
    }
    if (magic == MAGIC_RAW) {
        if (avail < 8) return 1;
        n_raw = ld_u32(p + 4); len = 8ull + n_raw;
        return avail < len ? 1 : 0;
    }
    return 2;
}

bool grow_dst(lzfse_mi_dstream *s, uint64_t cap) {
    std::free(s->far_dst); s->far_dst = nullptr; s->far_cap = 0;
    if (cap > ((uint64_t)1 << 30)) {
        s->far_dst = (uint8_t *)std::malloc((size_t)cap + 64);
        s->far_cap = s->far_dst ? (size_t)cap + 64 : 0;
        return s->far_dst != nullptr;
    }
    if (s->dst.cap >= cap + 64) return true;
    // (with room to spare: windows end at block boundaries, so their sizes differ by a block or two, and a buffer that fits
    // the largest one so far exactly is thrown away by every window that is a little larger -- fresh pinned pages cost more than the decode)
    const uint64_t want = cap + cap / 8 + ((uint64_t)1 << 20);
    return s->dst.reserve((size_t)want + 64, false) || s->dst.reserve((size_t)cap + 64, false);
}

// decode in[0 .. span) as one stream behind the history block; with_eos: the span's blocks are complete and bvx$ is added
int decode_span(lzfse_mi_dstream *s, size_t span, uint64_t raw, bool with_eos, lzfse_mi_write_fn write, void *user) {
    const size_t nh = s->hist.size();
    s->scan_span = 0; s->scan_raw = 0;
    const uint8_t *src = nullptr;
    size_t src_len = 0;
    // A window of complete blocks is decoded where it lies: the history block (bvx- header + the last 262 139 bytes of
    // output) is written over the consumed input in front of it, bvx$ over the 4 bytes behind it (put back afterwards).
    // Only a tail that could not be delimited is assembled in a buffer of its own.
    const size_t head = nh ? nh + 8 : 0;
    const bool in_place = with_eos && s->in_pos >= head;
    uint8_t saved[4] = {0, 0, 0, 0};
    size_t old_size = 0, eos_at = 0;
    if (in_place) {
        old_size = s->in.size;
        eos_at = s->in_pos + span;
        if (old_size < eos_at + 4) {
            if (!s->in.reserve(eos_at + 4)) return LZFSE_MI_IO;
            s->in.size = eos_at + 4;
        }
        uint8_t *p0 = s->in.p + s->in_pos - head;
        if (nh) {
            const uint32_t m = MAGIC_RAW, n32 = (uint32_t)nh;
            std::memcpy(p0, &m, 4); std::memcpy(p0 + 4, &n32, 4);
            std::memcpy(p0 + 8, s->hist.data(), nh);
        }
        std::memcpy(saved, s->in.p + eos_at, 4);
        const uint32_t m = MAGIC_EOS;
        std::memcpy(s->in.p + eos_at, &m, 4);
        src = p0; src_len = head + span + 4;
    } else {
        s->tmp_src.clear();
        if (nh) {
            uint8_t hd[8];
            const uint32_t m = MAGIC_RAW, n32 = (uint32_t)nh;
            std::memcpy(hd, &m, 4); std::memcpy(hd + 4, &n32, 4);
            s->tmp_src.insert(s->tmp_src.end(), hd, hd + 8);
            s->tmp_src.insert(s->tmp_src.end(), s->hist.begin(), s->hist.end());
        }
        s->tmp_src.insert(s->tmp_src.end(), s->in.p + s->in_pos, s->in.p + s->in_pos + span);
        if (with_eos) { const uint32_t m = MAGIC_EOS; const uint8_t *q = (const uint8_t *)&m; s->tmp_src.insert(s->tmp_src.end(), q, q + 4); }
        src = s->tmp_src.data(); src_len = s->tmp_src.size();
    }
    auto put_back = [&] {
        if (!in_place) return;
        std::memcpy(s->in.p + eos_at, saved, 4);
        s->in.size = old_size;
    };
    uint64_t cap64 = nh + raw;
    if (!with_eos) {   // a tail the parser could not delimit: what its headers promise, as the slice path's caller would size it
        uint64_t promised = 0;
        (void)lzfse_mi_decode_size(src, src_len, &promised);
        cap64 = std::max<uint64_t>(cap64, promised);
    }
    if (!grow_dst(s, cap64)) { put_back(); return LZFSE_MI_IO; }
    size_t got = 0;
    ctx_set_pinned_out(s->ctx, s->far_dst == nullptr);
    int st = lzfse_mi_decode(s->ctx, src, src_len, s->out_p(), (size_t)cap64, &got);
    if (st == LZFSE_MI_BUFFER_OVERFLOW) {   // the sink is unbounded: find the error the reference's Vec would have met
        cap64 += lzfse_mi_decode_headroom(src, src_len);
        if (!grow_dst(s, cap64)) { ctx_set_pinned_out(s->ctx, false); put_back(); return LZFSE_MI_IO; }
        ctx_set_pinned_out(s->ctx, s->far_dst == nullptr);
        st = lzfse_mi_decode(s->ctx, src, src_len, s->out_p(), (size_t)cap64, &got);
    }
    ctx_set_pinned_out(s->ctx, false);
    put_back();
    if (st) return st;
    if (got < nh) return LZFSE_MI_IO;
    const size_t fresh = got - nh;
    if (fresh && write && write(user, s->out_p() + nh, fresh)) return LZFSE_MI_IO;
    s->total_out += fresh;
    // history for the next window
    s->hist.assign(s->out_p() + (got >= MAX_D_VALUE ? got - MAX_D_VALUE : 0), s->out_p() + got);
    s->in_pos += span;
    s->total_in += span;
    return 0;
}

// ---- windows in the background ----

// the window in flight: its status; ptr / len: what it decoded (behind the history block), still to be handed to the sink
int ds_complete(lzfse_mi_dstream *s, const uint8_t **ptr, size_t *len, void **ev) {
    *ptr = nullptr; *len = 0;
    s->wait_idle();
    s->pending = false;
    *ev = s->pend_ev;     // (the next window's helper writes the field again)
    s->pend_ev = nullptr;
    if (s->pend_st || s->pend_got < s->pend_nh) {
        // (the caller of a failed window never looks at its bytes, and may free or reuse their buffer: nothing may still be travelling into it)
        if (*ev) (void)hipEventSynchronize((hipEvent_t)*ev);
        *ev = nullptr;
        return s->pend_st ? s->pend_st : LZFSE_MI_IO;
    }
    const uint8_t *base = s->adst[s->pend_buf].p;
    const size_t got = s->pend_got;
    *ptr = base + s->pend_nh; *len = got - s->pend_nh;
    s->hist.assign(base + (got >= MAX_D_VALUE ? got - MAX_D_VALUE : 0), base + got);   // history for the next window
    return 0;
}

int ds_sink(lzfse_mi_dstream *s, const uint8_t *ptr, size_t len, void *ev, lzfse_mi_write_fn write, void *user) {
    if (ev && hipEventSynchronize((hipEvent_t)ev) != hipSuccess) return LZFSE_MI_IO;   // (the window's bytes: over by now, as a rule)
    if (len && write && write(user, ptr, len)) return LZFSE_MI_IO;
    s->total_out += len;
    return 0;
}

// nothing in flight afterwards
int ds_drain(lzfse_mi_dstream *s, lzfse_mi_write_fn write, void *user) {
    if (!s->pending) return 0;
    const uint8_t *p; size_t n; void *ev;
    if (const int st = ds_complete(s, &p, &n, &ev)) return st;
    return ds_sink(s, p, n, ev, write, user);
}

// in[in_pos .. in_pos + span) -- complete blocks holding `raw` bytes -- go to the device in the background (nothing is in
// flight). -1: not this way (a window of more than a GiB, no helper to be had): the caller decodes it as before.
int ds_launch(lzfse_mi_dstream *s, size_t span, uint64_t raw) {
    const size_t nh = s->hist.size(), head = nh ? nh + 8 : 0, src_len = head + span + 4;
    const uint64_t cap64 = nh + raw;
    if (cap64 > ((uint64_t)1 << 30) || !s->box) return -1;
    {
        // (the background path runs on a context of the library's own: when that cannot be had -- no memory for a second set of
        // scratch -- the window is decoded in the call, on the caller's context, as before round 4)
        StreamBox *b = s->box.get();
        std::lock_guard<std::mutex> g(b->m);
        if (!b->dead && !box_peer(*b)) return -1;
    }
    if (!s->worker) {
        try { s->worker = new (std::nothrow) LaneWorker(); } catch (...) { s->worker = nullptr; }
        if (!s->worker) return -1;
    }
    if (!s->wsrc.reserve(src_len + 64, false)) return LZFSE_MI_IO;
    uint8_t *q = s->wsrc.p;
    if (nh) {
        const uint32_t m = MAGIC_RAW, n32 = (uint32_t)nh;
        std::memcpy(q, &m, 4); std::memcpy(q + 4, &n32, 4);
        std::memcpy(q + 8, s->hist.data(), nh);
    }
    { const uint32_t m = MAGIC_EOS; std::memcpy(q + head + span, &m, 4); }
    s->pend_from = s->in_pos; s->pend_span = span;
    s->copied.store(0, std::memory_order_relaxed);
    PinBuf &d = s->adst[s->cur];
    if (d.cap < cap64 + 64) {
        // (with room to spare, as grow_dst)
        const uint64_t want = cap64 + cap64 / 8 + ((uint64_t)1 << 20);
        if (!d.reserve((size_t)want + 64, false) && !d.reserve((size_t)cap64 + 64, false)) return LZFSE_MI_IO;
    }
    s->pend_st = 0; s->pend_got = 0; s->pend_nh = nh; s->pend_buf = s->cur; s->pend_len = src_len; s->pend_cap = cap64;
    s->pending = true;
    s->worker->submit([s] {
        std::memcpy(s->wsrc.p + (s->pend_nh ? s->pend_nh + 8 : 0), s->in.p + s->pend_from, s->pend_span);
        s->copied.store(1, std::memory_order_release);
        StreamBox *b = s->box.get();
        std::lock_guard<std::mutex> g(b->m);
        lzfse_mi_ctx *c = box_peer(*b);
        if (!c) { s->pend_st = b->dead ? LZFSE_MI_BAD_ARGUMENT : LZFSE_MI_IO; return; }
        PinBuf &dd = s->adst[s->pend_buf];
        uint64_t cap = s->pend_cap;
        size_t got = 0;
        // (the output may still be travelling when the call returns -- on a stream of its own, under the next window's call:
        // ds_complete's caller waits for pend_ev before it hands the bytes on)
        ctx_set_pinned_out(c, true);
        ctx_set_defer_out(c, true);
        int st = lzfse_mi_decode(c, s->wsrc.p, s->pend_len, dd.p, (size_t)cap, &got);
        if (st == LZFSE_MI_BUFFER_OVERFLOW) {   // the sink is unbounded: find the error the reference's Vec would have met (decode_span)
            cap += lzfse_mi_decode_headroom(s->wsrc.p, s->pend_len);
            if (!dd.reserve((size_t)cap + 64, false)) st = LZFSE_MI_IO;
            else st = lzfse_mi_decode(c, s->wsrc.p, s->pend_len, dd.p, (size_t)cap, &got);
        }
        s->pend_ev = st ? nullptr : ctx_deferred_event(c);
        ctx_set_defer_out(c, false);
        ctx_set_pinned_out(c, false);
        s->pend_st = st; s->pend_got = got;
    });
    s->cur ^= 1;
    s->scan_span = 0; s->scan_raw = 0;
    s->in_pos += span;
    s->total_in += span;
    return 0;
}

// A full window with more input to come: the one in flight is taken back, this one sent off, and what the former decoded handed
// to the sink while the device works on this one.
int ds_advance(lzfse_mi_dstream *s, size_t span, uint64_t raw, lzfse_mi_write_fn write, void *user) {
    const uint8_t *p = nullptr; size_t n = 0; void *ev = nullptr;
    if (s->pending)
        if (const int st = ds_complete(s, &p, &n, &ev)) return st;
    const int e = ds_launch(s, span, raw);
    if (const int st = ds_sink(s, p, n, ev, write, user)) return st;
    if (e > 0) return e;
    if (e < 0) return decode_span(s, span, raw, true, write, user);
    return 0;
}

}  // namespace

extern "C" {

// The reference writes into a Vec: a damaged block that produces MORE than its header's n_raw_bytes is decoded to its last
// LMD and then fails with BadLmdPayload / a bad D on the way (fse/fse_core.rs:103-140), where a destination of exactly
// lzfse_mi_decode_size bytes ends in BUFFER_OVERFLOW first. This is how much more such a block can write: 40 000 literals
// + 10 000 matches of 2 359 bytes for a bvx1/bvx2 block (fse/constants.rs), 136 bytes per payload byte for a bvxn block.
LZFSE_MI_API size_t lzfse_mi_decode_headroom(const uint8_t *src, size_t n) {
    if (!src) return 0;
    size_t pos = 0;
    uint64_t vn = 0;
    bool fse = false;
    while (n - pos >= 4) {
        uint64_t len, nr; bool eos;
        const uint32_t magic = ld_u32(src + pos);
        const int why = block_extent(src + pos, n - pos, len, nr, eos);
        if (magic == MAGIC_VX1 || magic == MAGIC_VX2) fse = true;
        if (magic == MAGIC_VXN) vn += std::min<uint64_t>(why == 0 ? len : n - pos, n - pos);
        if (why || eos) break;
        pos += (size_t)len;
    }
    return (size_t)((fse ? (uint64_t)LITERALS_PER_BLOCK + (uint64_t)LMDS_PER_BLOCK * MAX_M_VALUE : 0) + 136 * vn);
}

LZFSE_MI_API int lzfse_mi_dstream_create(lzfse_mi_ctx *ctx, size_t window, lzfse_mi_dstream **out) {
    if (!ctx || !out) return LZFSE_MI_BAD_ARGUMENT;
    lzfse_mi_dstream *s = new (std::nothrow) lzfse_mi_dstream;
    if (!s) return LZFSE_MI_IO;
    s->ctx = ctx;
    ctx_attach(ctx, &s->ctx);
    s->box = ctx_stream_box(ctx);
    s->window = window ? window : (size_t)LZFSE_MI_STREAM_WINDOW;
    {
        StreamSpare &sp = ctx_spare(ctx);
        s->in.swap(sp.b[0]);
        s->dst.swap(sp.b[1]);
        s->wsrc.swap(sp.b[4]);
        s->adst[0].swap(sp.b[5]); s->adst[1].swap(sp.b[6]);
        s->in.size = 0; s->dst.size = 0; s->wsrc.size = 0; s->adst[0].size = 0; s->adst[1].size = 0;
    }
    *out = s;
    return LZFSE_MI_OK;
}

LZFSE_MI_API void lzfse_mi_dstream_destroy(lzfse_mi_dstream *s) { delete s; }

LZFSE_MI_API int lzfse_mi_dstream_totals(const lzfse_mi_dstream *s, uint64_t *bytes_in, uint64_t *bytes_out) {
    if (!s) return LZFSE_MI_BAD_ARGUMENT;
    if (bytes_in) *bytes_in = s->total_in;
    if (bytes_out) *bytes_out = s->total_out;
    return LZFSE_MI_OK;
}

// Where the next `want` input bytes go (LzfseRingDecoder::decode reads straight into its ring, decode/ring_decoder.rs:57-67):
// read into *ptr, then commit what came. No decoding happens here.
LZFSE_MI_API int lzfse_mi_dstream_reserve(lzfse_mi_dstream *s, size_t want, uint8_t **ptr) {
    if (!s || !ptr || !want) return LZFSE_MI_BAD_ARGUMENT;
    *ptr = nullptr;
    s->reserved = 0;
    if (s->status) return s->status;
    if (!s->ctx) return s->status = LZFSE_MI_BAD_ARGUMENT;   // its context has been destroyed
    // (consumed input is dropped now and then; MAX_D_VALUE + 8 bytes of it stay in front of the rest: the next window's
    // history block is written there, decode_span)
    constexpr size_t KEEP = (size_t)MAX_D_VALUE + 8;
    if (s->in_pos > KEEP && s->in_pos - KEEP >= s->in.size / 2) { s->wait_copied(); s->in.erase_front(s->in_pos - KEEP); s->in_pos = KEEP; }
    if (s->in.size + want > s->in.cap) {
        s->wait_copied();   // (the buffer is about to move)
        if (!s->in.reserve(std::max<size_t>(std::max(s->in.size + want, 2 * s->in.cap), (size_t)256 << 10))) return s->status = LZFSE_MI_IO;
    }
    *ptr = s->in.p + s->in.size;
    s->reserved = want;
    return LZFSE_MI_OK;
}

static int ds_process(lzfse_mi_dstream *s, int finish, lzfse_mi_write_fn write, void *user);

// n <= want bytes have been stored where lzfse_mi_dstream_reserve said; then as lzfse_mi_dstream_feed does after taking its bytes
LZFSE_MI_API int lzfse_mi_dstream_commit(lzfse_mi_dstream *s, size_t n, int finish, lzfse_mi_write_fn write, void *user) {
    if (!s || n > s->reserved) return LZFSE_MI_BAD_ARGUMENT;
    if (s->status) return s->status;
    if (!s->ctx) return s->status = LZFSE_MI_BAD_ARGUMENT;
    s->in.size += n;
    s->reserved = 0;
    return ds_process(s, finish, write, user);
}

LZFSE_MI_API int lzfse_mi_dstream_feed(lzfse_mi_dstream *s, const uint8_t *src, size_t n, int finish, lzfse_mi_write_fn write, void *user) {
    if (!s || (!src && n)) return LZFSE_MI_BAD_ARGUMENT;
    if (s->status) return s->status;
    if (!s->ctx) return s->status = LZFSE_MI_BAD_ARGUMENT;   // its context has been destroyed
    if (n) {
        uint8_t *p;
        if (const int st = lzfse_mi_dstream_reserve(s, n, &p)) return st;
        std::memcpy(p, src, n);
        s->in.size += n;
        s->reserved = 0;
    }
    return ds_process(s, finish, write, user);
}

static int ds_process(lzfse_mi_dstream *s, int finish, lzfse_mi_write_fn write, void *user) {
    for (;;) {
        if (s->eos_seen) {
            if (s->in.size > s->in_pos) return s->status = LZFSE_MI_PAYLOAD_OVERFLOW;   // bytes behind bvx$ (decoder.rs:93-95)
            return LZFSE_MI_OK;
        }
        // the longest run of complete blocks at the front of the input, up to a window of raw bytes
        size_t span = s->scan_span;      // (what earlier feeds have scanned is not scanned again)
        uint64_t raw = s->scan_raw;
        int why = 1;   // why the run ended: 0 bvx$, 1 input exhausted, 2 undecidable block, 3 window full
        while (true) {
            uint64_t len, nr; bool eos;
            why = block_extent(s->in.p + s->in_pos + span, s->in.size - s->in_pos - span, len, nr, eos);
            if (why) break;
            if (eos) break;
            span += (size_t)len; raw += nr;
            if (raw >= s->window) { why = 3; break; }
        }
        if (why == 0) {
            // bvx$ follows the run: decode the run, consume the magic; whether it ends the input shows at once or later
            if (const int st = ds_drain(s, write, user)) return s->status = st;
            if (span) { const int st = decode_span(s, span, raw, true, write, user); if (st) return s->status = st; }
            s->in_pos += 4;
            s->total_in += 4;
            s->eos_seen = true;
            continue;
        }
        if (why == 3) { const int st = ds_advance(s, span, raw, write, user); if (st) return s->status = st; continue; }
        if (why == 1 && !finish) {
            // wait for more input (a window is decoded when it is full, or when the input ends)
            s->scan_span = span; s->scan_raw = raw;
            return LZFSE_MI_OK;
        }
        // the input ends inside a block, or a block cannot be delimited: the rest goes to the device as it is, which
        // reports what the slice path reports for it (PayloadUnderflow, BadBlock, a header error ...)
        // The sound blocks in front of it go first, as a window of their own: what the sink holds and what the totals say
        // when the error comes must not depend on how the input was cut into feeds (the reference has written those
        // blocks by then: decoder.rs:76-99 decodes block by block).
        if (const int st = ds_drain(s, write, user)) return s->status = st;
        if (span) { const int st = decode_span(s, span, raw, true, write, user); if (st) return s->status = st; }
        {
            const int st = decode_span(s, s->in.size - s->in_pos, 0, false, write, user);
            return s->status = st ? st : LZFSE_MI_PAYLOAD_UNDERFLOW;   // (no bvx$: cannot have decoded cleanly)
        }
    }
}

}  // extern "C"

// ------------------------------------------------------------------------------------ streaming encode (LzfseWriter)

struct lzfse_mi_estream {
    lzfse_mi_ctx *ctx = nullptr;
    size_t window = 0;            // new input bytes per device call
    PinBuf buf;                   // buf[head ..]: the input from position `base` on: what the parse may still look at, and what it has not seen yet (pinned, like `out`)
    size_t head = 0;              // (what lies below it is dropped when the buffer is next put in order: never while a window is in flight)
    size_t live() const { return buf.size - head; }
    uint64_t base = 0;            // position of buf[head] in the stream, a multiple of 16 KiB
    bool have_state = false;      // a window has been cut: the parse goes on from `st` (positions relative to base)
    uint32_t st[5] = {};          // index, literal_index, pending (idx, match idx, len): encode/frontend_ring.rs' idx, literal_idx, pending
    uint32_t skip = 0;            // bytes of the first event the parse makes from there that have left already (a block ended inside it)
    size_t next_at = 0;           // live() at which the next window is tried
    PinBuf out;                   // one window's bytes
    uint64_t total_in = 0, total_out = 0;
    int status = 0;               // sticky
    bool finished = false;
    // A window that is not the last is encoded in the BACKGROUND, on the stream objects' own context (StreamBox), while the caller
    // goes on feeding: the copy of a window's input into `buf` takes about as long as the device takes for the window before.
    // What the window made leaves through `write` at the caller's next call that gets that far (es_complete); `buf` neither moves
    // nor shrinks while a window is in flight, and bytes behind the window's end are the caller's to append.
    std::shared_ptr<StreamBox> box;
    LaneWorker *worker = nullptr;
    bool pending = false;
    size_t pend_n = 0, pend_len = 0;
    int pend_st = 0;
    EncWindow pend_w;
    size_t reserved = 0;          // bytes lzfse_mi_estream_reserve has promised behind buf.size
    void wait_idle() { if (worker && pending) worker->wait(); }
    ~lzfse_mi_estream() {
        wait_idle();
        delete worker;
        if (!ctx) return;   // (the context went first)
        ctx_detach(ctx, &ctx);
        StreamSpare &sp = ctx_spare(ctx);
        if (sp.keep && buf.cap > sp.b[2].cap) { buf.size = 0; sp.b[2].swap(buf); }
        if (sp.keep && out.cap > sp.b[3].cap) { out.size = 0; sp.b[3].swap(out); }
    }
};

namespace {

// what a position of the parse may still reach back to: 262 139 bytes of match distance (fse/constants.rs:42) below the
// literal index (a backward extension stops there on one side and moves the candidate down with it on the other), and a
// compare that runs past the ring's tail reads one ring earlier, at most 272 KiB below the position
constexpr size_t E_KEEP = 262139 + 65536;
constexpr size_t E_MIN_WINDOW = (size_t)1 << 20;

int es_out_room(lzfse_mi_estream *s, size_t cap) { return s->out.reserve(cap, false) ? 0 : LZFSE_MI_IO; }

int es_write(lzfse_mi_estream *s, lzfse_mi_write_fn write, void *user, size_t len) {
    // the sink takes the stream in pieces, as the reference's 8 KiB output ring does (encode/constants.rs:36-48); larger here
    for (size_t o = 0; o < len; o += (size_t)1 << 20)
        if (write(user, s->out.p + o, len - o < ((size_t)1 << 20) ? len - o : (size_t)1 << 20)) return LZFSE_MI_IO;
    s->total_out += len;
    return 0;
}

// One device call over the first n bytes on hand, on context c. w: the window's state in, where it was cut out.
int es_call(lzfse_mi_ctx *c, lzfse_mi_estream *s, size_t n, EncWindow *w, size_t cap, size_t *len) {
    ctx_set_pinned_out(c, true);   // (the window's bytes land in `out`: pinned memory)
    ctx_set_window(c, w);
    const int st = lzfse_mi_encode_ring(c, s->buf.p + s->head, n, s->out.p, cap, len);
    ctx_set_window(c, nullptr);
    ctx_set_pinned_out(c, false);
    return st;
}

// The input ends here: one call over everything on hand, on the caller's context (nothing is in flight).
int es_final(lzfse_mi_estream *s, lzfse_mi_write_fn write, void *user) {
    const size_t n = s->live();
    const size_t cap = lzfse_mi_encode_bound(n);
    if (es_out_room(s, cap ? cap : 1)) return LZFSE_MI_IO;
    size_t len = 0;
    if (!s->have_state) {
        // the whole input in one call (all size classes)
        const int st = es_call(s->ctx, s, n, nullptr, cap, &len);
        return st ? st : es_write(s, write, user, len);
    }
    if (n > (size_t)0x7FFFFFFFu) return LZFSE_MI_UNSUPPORTED;   // positions are 31 bits on the device
    EncWindow w;
    w.start = true; w.beyond = s->base != 0; w.final = true;
    for (int k = 0; k < 5; k++) w.st[k] = s->st[k];
    w.skip = s->skip;
    const int st = es_call(s->ctx, s, n, &w, cap, &len);
    return st ? st : es_write(s, write, user, len);
}

// A window that is not the last: everything on hand goes to the device as one call, in the background. The window will be
// cut behind its last block that no later byte can change (enc_cut_kernel, encode_parse.hip).
int es_launch(lzfse_mi_estream *s) {
    const size_t n = s->live();
    if (n > (size_t)0x7FFFFFFFu) return LZFSE_MI_UNSUPPORTED;   // positions are 31 bits on the device
    const size_t cap = lzfse_mi_encode_bound(n);
    if (es_out_room(s, cap)) return LZFSE_MI_IO;
    // The caller appends behind the window while it is in flight, so `buf` must neither move nor run out meanwhile: room for
    // another window's worth behind what is on hand. The bytes of earlier windows that nothing needs any more (below `head`) are
    // dropped here, when the room is short -- every other window with three windows of room -- and not after every window.
    const size_t slack = s->window + ((size_t)1 << 16);
    if (s->buf.size + slack > s->buf.cap) {
        if (s->head) { s->buf.erase_front(s->head); s->head = 0; }
        if (s->buf.size + slack > s->buf.cap && !s->buf.reserve(s->buf.size + 2 * slack)) return LZFSE_MI_IO;
    }
    if (!s->worker) {
        try { s->worker = new (std::nothrow) LaneWorker(); } catch (...) { s->worker = nullptr; }
    }
    EncWindow &w = s->pend_w;
    w = EncWindow();
    w.start = s->have_state; w.beyond = s->base != 0; w.final = false;
    for (int k = 0; k < 5; k++) w.st[k] = s->st[k];
    w.skip = s->skip;
    s->pend_n = n; s->pend_len = 0; s->pend_st = 0;
    // (no second context to be had -- the peer is a full set of device scratch -- and the box alive: the window runs in this call on
    // the caller's own context, as before round 4, instead of failing the stream)
    bool own = false;
    if (StreamBox *b = s->box.get()) {
        std::lock_guard<std::mutex> g(b->m);
        own = !b->dead && !box_peer(*b) && s->ctx;
    }
    auto job = [s, n, cap, own] {
        if (own) { s->pend_st = es_call(s->ctx, s, n, &s->pend_w, cap, &s->pend_len); return; }
        StreamBox *b = s->box.get();
        if (!b) { s->pend_st = LZFSE_MI_IO; return; }
        std::lock_guard<std::mutex> g(b->m);
        lzfse_mi_ctx *c = box_peer(*b);
        s->pend_st = c ? es_call(c, s, n, &s->pend_w, cap, &s->pend_len) : (b->dead ? LZFSE_MI_BAD_ARGUMENT : LZFSE_MI_IO);
    };
    s->pending = true;
    if (s->worker && !own) s->worker->submit(job);
    else job();   // (no thread to be had, or the caller's own context: the window runs here)
    return 0;
}

// What the window in flight made: its final blocks leave through `write`, and the buffer keeps what the parse still needs:
// E_KEEP bytes below the literal index of the cut, and everything from there on.
int es_complete(lzfse_mi_estream *s, lzfse_mi_write_fn write, void *user) {
    s->wait_idle();
    s->pending = false;
    if (s->pend_st) return s->pend_st;
    const EncWindow &w = s->pend_w;
    const size_t n = s->pend_n;
    if (!w.found) {
        // no block of this window is final yet (few, very long matches: a block of 10 000 LMDs can span many MiB): more input first
        s->next_at = n + (n > s->window ? n : s->window);
        return 0;
    }
    if (const int e = es_write(s, write, user, s->pend_len)) return e;
    size_t keep = w.lit > E_KEEP ? (size_t)w.lit - E_KEEP : 0;
    keep &= ~(size_t)(0x4000 - 1);
    s->head += keep;
    s->base += keep;
    const uint32_t k32 = (uint32_t)keep;
    s->st[0] = w.index - k32; s->st[1] = w.lit - k32;
    s->st[2] = w.p_len ? w.p_idx - k32 : 0; s->st[3] = w.p_len ? w.p_midx - k32 : 0; s->st[4] = w.p_len;
    s->skip = w.skip_out;
    s->have_state = true;
    s->next_at = (n - keep) + s->window;
    return 0;
}

}  // namespace

extern "C" {

int lzfse_mi_estream_create(lzfse_mi_ctx *ctx, size_t window, lzfse_mi_estream **out) {
    if (!ctx || !out) return LZFSE_MI_BAD_ARGUMENT;
    lzfse_mi_estream *s = new (std::nothrow) lzfse_mi_estream();
    if (!s) return LZFSE_MI_IO;
    s->ctx = ctx;
    ctx_attach(ctx, &s->ctx);
    s->box = ctx_stream_box(ctx);
    s->window = window ? window : (size_t)LZFSE_MI_STREAM_WINDOW;
    if (s->window < E_MIN_WINDOW) s->window = E_MIN_WINDOW;
    if (s->window > ((size_t)1 << 30)) s->window = (size_t)1 << 30;
    s->next_at = s->window + ((size_t)1 << 19);   // (the last 256 KiB + 16 KiB of a window are never final: one ring on top)
    {
        StreamSpare &sp = ctx_spare(ctx);
        s->buf.swap(sp.b[2]);
        s->out.swap(sp.b[3]);
        s->buf.size = 0; s->out.size = 0;
    }
    *out = s;
    return LZFSE_MI_OK;
}

void lzfse_mi_estream_destroy(lzfse_mi_estream *s) { delete s; }

// Where the next input bytes go, and how many fit there (1 <= *room <= want): LzfseRingEncoder::encode's copy(reader)
// (encode/ring_encoder.rs:55-67) reads straight into its ring, and so can a binding -- read into *ptr, then commit what came.
// Making room may take a finished window back (its blocks leave through `write`) and send the next one off.
int lzfse_mi_estream_reserve(lzfse_mi_estream *s, size_t want, uint8_t **ptr, size_t *room, lzfse_mi_write_fn write, void *user) {
    if (!s || !ptr || !room || !want || !write || s->finished) return LZFSE_MI_BAD_ARGUMENT;
    *ptr = nullptr; *room = 0;
    s->reserved = 0;
    if (s->status) return s->status;
    if (!s->ctx) return s->status = LZFSE_MI_BAD_ARGUMENT;   // its context has been destroyed
    for (;;) {
        // room: up to where the next window is tried; with a window in flight also no further than `buf` reaches without moving
        size_t r = s->next_at > s->live() ? s->next_at - s->live() : 0;
        if (s->pending && r > s->buf.cap - s->buf.size) r = s->buf.cap - s->buf.size;
        if (r == 0) {
            if (s->pending) { if (const int st = es_complete(s, write, user)) return s->status = st; continue; }
            if (const int st = es_launch(s)) return s->status = st;
            s->next_at = s->live() + s->window;        // (until the window says where it was cut)
            continue;
        }
        if (r > want) r = want;
        // (room for the whole window at once, as soon as the input shows that it will be needed: growing step by step copies
        // the window's bytes again and again and pins twice the pages)
        if (!s->pending && s->buf.cap < s->head + s->next_at && s->buf.size + r > ((size_t)4 << 20) && !s->buf.reserve(s->head + s->next_at + ((size_t)1 << 16))) return s->status = LZFSE_MI_IO;
        if (s->buf.size + r > s->buf.cap && !s->buf.reserve(std::max<size_t>(std::max(s->buf.size + r, 2 * s->buf.cap), (size_t)256 << 10))) return s->status = LZFSE_MI_IO;
        *ptr = s->buf.p + s->buf.size; *room = r;
        s->reserved = r;
        return LZFSE_MI_OK;
    }
}

// n <= the room last reserved bytes have been stored where lzfse_mi_estream_reserve said
int lzfse_mi_estream_commit(lzfse_mi_estream *s, size_t n) {
    if (!s || s->finished || n > s->reserved) return LZFSE_MI_BAD_ARGUMENT;
    if (s->status) return s->status;
    s->buf.size += n; s->total_in += n;
    s->reserved = 0;
    return LZFSE_MI_OK;
}

// Write::write of LzfseWriter (encode/writer.rs:59-63): takes all of buf; whenever a window's worth of input is on hand the
// device encodes it -- in the background, while this call and the next ones take more input -- and the blocks that are final
// leave through `write` as soon as a call finds the window done
int lzfse_mi_estream_feed(lzfse_mi_estream *s, const uint8_t *src, size_t n, lzfse_mi_write_fn write, void *user) {
    if (!s || (!src && n) || !write || s->finished) return LZFSE_MI_BAD_ARGUMENT;
    if (s->status) return s->status;
    if (!s->ctx) return s->status = LZFSE_MI_BAD_ARGUMENT;   // its context has been destroyed
    while (n) {
        uint8_t *p; size_t room;
        if (const int st = lzfse_mi_estream_reserve(s, n, &p, &room, write, user)) return st;
        std::memcpy(p, src, room);
        if (const int st = lzfse_mi_estream_commit(s, room)) return st;
        src += room; n -= room;
    }
    return LZFSE_MI_OK;
}

// LzfseWriter::finalize (encode/writer.rs:50-53) = FrontendRing::flush (frontend_ring.rs:275-295) + the writer's flush
int lzfse_mi_estream_finish(lzfse_mi_estream *s, lzfse_mi_write_fn write, void *user, uint64_t *bytes_in, uint64_t *bytes_out) {
    if (!s || !write || s->finished) return LZFSE_MI_BAD_ARGUMENT;
    s->finished = true;
    if (s->status) { s->wait_idle(); s->pending = false; return s->status; }
    if (!s->ctx) { s->wait_idle(); s->pending = false; return s->status = LZFSE_MI_BAD_ARGUMENT; }   // its context has been destroyed
    int st = 0;
    if (s->pending) st = es_complete(s, write, user);
    if (!st) st = es_final(s, write, user);
    if (bytes_in) *bytes_in = s->total_in;
    if (bytes_out) *bytes_out = st ? 0 : s->total_out;
    s->buf.size = 0; s->head = 0;   // (its room goes back to the context with the object)
    return s->status = st;
}

}  // extern "C"
