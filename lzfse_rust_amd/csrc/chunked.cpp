// Chunked container + one-process multi-device entry points (BASELINE config 5).
//
// A large input is cut into fixed-size chunks; every chunk is encoded as its own, complete LZFSE stream (the reference
// resets its tables per call, encode/frontend_bytes.rs:113-119, so chunks are independent) and chunk c goes to context
// c mod n_ctx -- one context per device, no collective, the host gathers. The framing is this build's own; a plain
// LzfseDecoder decodes each chunk stream but not the container (EOS rule, decode/decoder.rs:93-95).
//
//   frame := "LZMC" u16 version(1) u16 flags(0) u32 chunk_size u32 n_chunks u64 raw_total
//            n_chunks x { u32 raw_len, u32 stream_len }  streams back to back            (all little-endian)
//
// Built on the public batch entry points only (lzfse_mi_encode_batch / lzfse_mi_decode_batch): one host thread per
// context drives that context's share of the chunks.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "../../include/lzfse_mi.h"

namespace {

constexpr uint32_t kMagic = 0x434D5A4Cu;  // "LZMC"
constexpr size_t kFixed = 24;             // bytes in front of the chunk table

inline void put32(uint8_t *p, uint32_t v) { std::memcpy(p, &v, 4); }
inline void put64(uint8_t *p, uint64_t v) { std::memcpy(p, &v, 8); }
inline uint32_t get32(const uint8_t *p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
inline uint64_t get64(const uint8_t *p) { uint64_t v; std::memcpy(&v, p, 8); return v; }

struct Frame {
    uint32_t chunk = 0, n_chunks = 0;
    uint64_t raw_total = 0;
    const uint8_t *table = nullptr, *payload = nullptr;
    size_t payload_len = 0;
};

int parse(const uint8_t *src, size_t n, Frame &f) {
    if (!src || n < kFixed) return LZFSE_MI_PAYLOAD_UNDERFLOW;
    if (get32(src) != kMagic) return LZFSE_MI_BAD_BLOCK;
    uint16_t version;
    std::memcpy(&version, src + 4, 2);
    if (version != 1) return LZFSE_MI_UNSUPPORTED;
    f.chunk = get32(src + 8);
    f.n_chunks = get32(src + 12);
    f.raw_total = get64(src + 16);
    if (f.chunk == 0 && f.n_chunks != 0) return LZFSE_MI_BAD_ARGUMENT;
    if ((n - kFixed) / 8 < f.n_chunks) return LZFSE_MI_PAYLOAD_UNDERFLOW;
    f.table = src + kFixed;
    f.payload = f.table + (size_t)f.n_chunks * 8;
    f.payload_len = n - kFixed - (size_t)f.n_chunks * 8;
    uint64_t raw = 0, enc = 0;
    for (uint32_t c = 0; c < f.n_chunks; c++) {
        const uint32_t r = get32(f.table + 8 * (size_t)c), e = get32(f.table + 8 * (size_t)c + 4);
        if (r > f.chunk || (r != f.chunk && c + 1 != f.n_chunks)) return LZFSE_MI_BAD_ARGUMENT;
        raw += r;
        enc += e;
    }
    if (raw != f.raw_total) return LZFSE_MI_BAD_ARGUMENT;
    if (enc > f.payload_len) return LZFSE_MI_PAYLOAD_UNDERFLOW;
    if (enc < f.payload_len) return LZFSE_MI_PAYLOAD_OVERFLOW;
    return LZFSE_MI_OK;
}

// Runs work(1) .. work(n - 1) on threads of their own and work(0) here; a thread that cannot be had has its share done here too.
// Nothing is thrown across the C ABI: the callers catch what is left (allocation failures of their vectors).
template <class F> void fan_out(int n, F &work) {
    std::vector<std::thread> th;
    std::vector<int> here;
    for (int k = 1; k < n; k++) {
        try { th.emplace_back([&work, k] { work(k); }); } catch (...) { here.push_back(k); }
    }
    work(0);
    for (int k : here) work(k);
    for (auto &t : th) t.join();
}

}  // namespace

extern "C" {

size_t lzfse_mi_chunked_bound(size_t n, size_t chunk) {
    if (chunk == 0) chunk = LZFSE_MI_CHUNK_DEFAULT;
    const size_t n_chunks = (n + chunk - 1) / chunk;
    return kFixed + n_chunks * 8 + n_chunks * lzfse_mi_encode_bound(chunk < n ? chunk : n) + 64;
}

int lzfse_mi_encode_chunked(lzfse_mi_ctx *const *ctxs, int n_ctx, const uint8_t *src, size_t n, size_t chunk, uint8_t *dst,
                            size_t cap, size_t *out_len) {
    if (!ctxs || n_ctx < 1 || (!src && n) || !dst || !out_len) return LZFSE_MI_BAD_ARGUMENT;
    for (int k = 0; k < n_ctx; k++)
        if (!ctxs[k]) return LZFSE_MI_BAD_ARGUMENT;
    if (chunk == 0) chunk = LZFSE_MI_CHUNK_DEFAULT;
    if (chunk > 0x7FFFFFFFu) return LZFSE_MI_BAD_ARGUMENT;
    const size_t n_chunks = (n + chunk - 1) / chunk;
    if (n_chunks > 0xFFFFFFFFu) return LZFSE_MI_UNSUPPORTED;
    const size_t head = kFixed + n_chunks * 8;
    if (cap < head) return LZFSE_MI_BUFFER_OVERFLOW;
    try {
        // Every context encodes its chunks (c mod n_ctx). Round 5: when the destination has the room lzfse_mi_chunked_bound asks
        // for, a chunk is encoded INTO the frame, at the place it would have if every stream before it filled its bound, and the
        // streams are then moved down to lie back to back (one pass over the compressed bytes, in place). Round 4 encoded every
        // chunk into a zero-filled vector of its own (1.75 x the input touched before a byte was encoded) and copied it over.
        // A smaller destination gets private buffers, uninitialised, as a caller that sized `dst` by guesswork would need.
        const size_t slot = lzfse_mi_encode_bound(std::min(chunk, n));
        const bool in_frame = n_chunks == 0 || (cap - head) / (n_chunks ? n_chunks : 1) >= slot;
        std::vector<std::unique_ptr<uint8_t[]>> priv(in_frame ? 0 : n_chunks);
        std::vector<size_t> enc_len(n_chunks, 0);
        std::vector<int> rc((size_t)n_ctx, LZFSE_MI_OK);
        auto work = [&](int k) {
            try {
                std::vector<const uint8_t *> srcs;
                std::vector<size_t> lens, caps, outs;
                std::vector<uint8_t *> dsts;
                std::vector<size_t> ids;
                for (size_t c = (size_t)k; c < n_chunks; c += (size_t)n_ctx) {
                    const size_t off = c * chunk, len = std::min(chunk, n - off);
                    const size_t bound = lzfse_mi_encode_bound(len);
                    uint8_t *to = in_frame ? dst + head + c * slot : (priv[c].reset(new uint8_t[bound]), priv[c].get());
                    srcs.push_back(src + off); lens.push_back(len);
                    dsts.push_back(to); caps.push_back(bound);
                    ids.push_back(c);
                }
                if (ids.empty()) return;
                outs.assign(ids.size(), 0);
                std::vector<int> st(ids.size(), 0);
                int r = lzfse_mi_encode_batch(ctxs[k], ids.size(), srcs.data(), lens.data(), dsts.data(), caps.data(), outs.data(), st.data());
                for (size_t j = 0; j < ids.size() && !r; j++) {
                    if (st[j]) r = st[j];
                    enc_len[ids[j]] = outs[j];
                }
                rc[(size_t)k] = r;
            } catch (...) { rc[(size_t)k] = LZFSE_MI_IO; }
        };
        fan_out(n_ctx, work);
        for (int r : rc)
            if (r) return r;
        size_t total = head;
        for (size_t c = 0; c < n_chunks; c++) {
            if (enc_len[c] > 0xFFFFFFFFu) return LZFSE_MI_UNSUPPORTED;
            total += enc_len[c];
        }
        if (total > cap) return LZFSE_MI_BUFFER_OVERFLOW;
        put32(dst, kMagic);
        const uint16_t version = 1, flags = 0;
        std::memcpy(dst + 4, &version, 2);
        std::memcpy(dst + 6, &flags, 2);
        put32(dst + 8, (uint32_t)chunk);
        put32(dst + 12, (uint32_t)n_chunks);
        put64(dst + 16, (uint64_t)n);
        uint8_t *p = dst + head;
        for (size_t c = 0; c < n_chunks; c++) {
            put32(dst + kFixed + 8 * c, (uint32_t)std::min(chunk, n - c * chunk));
            put32(dst + kFixed + 8 * c + 4, (uint32_t)enc_len[c]);
            const uint8_t *from = in_frame ? dst + head + c * slot : priv[c].get();
            if (from != p) std::memmove(p, from, enc_len[c]);   // (in the frame: downwards, stream by stream, never over one not yet moved)
            p += enc_len[c];
        }
        *out_len = total;
        return LZFSE_MI_OK;
    } catch (...) { return LZFSE_MI_IO; }
}

int lzfse_mi_decode_chunked_size(const uint8_t *src, size_t n, uint64_t *raw_len) {
    if (!raw_len) return LZFSE_MI_BAD_ARGUMENT;
    Frame f;
    const int r = parse(src, n, f);
    if (r) return r;
    *raw_len = f.raw_total;
    return LZFSE_MI_OK;
}

int lzfse_mi_decode_chunked(lzfse_mi_ctx *const *ctxs, int n_ctx, const uint8_t *src, size_t n, uint8_t *dst, size_t cap,
                            size_t *out_len) {
    if (!ctxs || n_ctx < 1 || !out_len || (!dst && cap)) return LZFSE_MI_BAD_ARGUMENT;
    for (int k = 0; k < n_ctx; k++)
        if (!ctxs[k]) return LZFSE_MI_BAD_ARGUMENT;
    Frame f;
    int r = parse(src, n, f);
    if (r) return r;
    if (f.raw_total > cap) return LZFSE_MI_BUFFER_OVERFLOW;
    try {
    std::vector<size_t> enc_off(f.n_chunks), raw_off(f.n_chunks);
    size_t eo = 0, ro = 0;
    for (uint32_t c = 0; c < f.n_chunks; c++) {
        enc_off[c] = eo; raw_off[c] = ro;
        ro += get32(f.table + 8 * (size_t)c);
        eo += get32(f.table + 8 * (size_t)c + 4);
    }
    std::vector<int> rc((size_t)n_ctx, LZFSE_MI_OK);
    auto work = [&](int k) {
        try {
        std::vector<const uint8_t *> srcs;
        std::vector<size_t> lens, caps, outs;
        std::vector<uint8_t *> dsts;
        for (size_t c = (size_t)k; c < f.n_chunks; c += (size_t)n_ctx) {
            srcs.push_back(f.payload + enc_off[c]); lens.push_back(get32(f.table + 8 * c + 4));
            dsts.push_back(dst + raw_off[c]); caps.push_back(get32(f.table + 8 * c));
        }
        if (srcs.empty()) return;
        outs.assign(srcs.size(), 0);
        std::vector<int> st(srcs.size(), 0);
        int e = lzfse_mi_decode_batch(ctxs[k], srcs.size(), srcs.data(), lens.data(), dsts.data(), caps.data(), outs.data(), st.data());
        for (size_t j = 0; j < srcs.size() && !e; j++) {
            if (st[j]) e = st[j];
            else if (outs[j] != caps[j]) e = LZFSE_MI_BAD_ARGUMENT;   // the chunk table promised another size
        }
        rc[(size_t)k] = e;
        } catch (...) { rc[(size_t)k] = LZFSE_MI_IO; }
    };
    fan_out(n_ctx, work);
    for (int e : rc)
        if (e) return e;
    *out_len = (size_t)f.raw_total;
    return LZFSE_MI_OK;
    } catch (...) { return LZFSE_MI_IO; }
}

}  // extern "C"
