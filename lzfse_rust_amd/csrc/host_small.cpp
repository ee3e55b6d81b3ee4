// Host-side size classes of LzfseEncoder::encode_bytes that the reference itself keeps on the CPU
// (encode/frontend_bytes.rs:63-111): inputs of 0..=20 bytes become one bvx- block, inputs of
// 21..=4096 bytes one bvxn (LZVN) block - replaced by bvx- when that is not larger - followed by
// bvx$. The device encoder (encode.hip) only takes inputs > 4096 bytes (bvx2), exactly like
// FrontendBytes::flush_select; nothing here is a fallback for the device path.
//
// LZVN emission follows vn/backend.rs:37-136 and vn/opc.rs; the parse is the same lazy matcher as
// the bvx2 path (frontend_bytes.rs:160-344, match_object.rs:12-33) instantiated for the Vn match
// unit: 3-byte hash, 3- or 4+-byte matches, distances <= 65 535 (vn/object.rs:9-60).
//
// The ring / stream encoder (LzfseRingEncoder::encode, LzfseWriter: encode/frontend_ring.rs) makes the same choice of
// block kinds below one ring (flush_select :297-312) but runs its own loop over them, match_short (:401-450), which is
// not the slice loop for the Vn unit: it visits one more position (idx < tail - MATCH_UNIT + 1 = n - 2, so the last
// 4-byte load reads one byte past the input) and it measures candidates with the ring's coarse compare, which runs past
// the end of the input (st. ring/object.rs:39-84: what lies there in a fresh ring is zeros) before the winner is cut
// back to the input's end. `ring` selects that loop; 16 % of random inputs of 21..4096 bytes come out differently.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/lzfse_mi.h"

namespace {

constexpr uint32_t kRawCutoff = 20, kVnCutoff = 4096, kGoodMatch = 40, kVnMaxDist = 65535;
constexpr uint32_t kHashBits = 14;

inline uint32_t load32(const uint8_t *p) { uint32_t v; std::memcpy(&v, p, 4); return v; }

struct Candidate { uint32_t idx = 0, midx = 0, len = 0; };

// 2^14 buckets x 4 (value, position) pairs, newest first (encode/history.rs:15-31,99-118)
class VnHistory {
  public:
    VnHistory() : val_(size_t(4) << kHashBits, 0), pos_(size_t(4) << kHashBits, 0xC0000000u) {}
    // returns the bucket as it was before inserting (val, pos)
    void push(uint32_t val, uint32_t pos, uint32_t out_val[4], uint32_t out_pos[4]) {
        size_t b = size_t(((val & 0x00FFFFFFu) * 0x9E3779B1u) >> (32 - kHashBits)) * 4;
        for (int k = 0; k < 4; k++) { out_val[k] = val_[b + k]; out_pos[k] = pos_[b + k]; }
        for (int k = 3; k > 0; k--) { val_[b + k] = val_[b + k - 1]; pos_[b + k] = pos_[b + k - 1]; }
        val_[b] = val; pos_[b] = pos;
    }
  private:
    std::vector<uint32_t> val_, pos_;
};

class VnWriter {  // vn/backend.rs:57-136
  public:
    explicit VnWriter(std::vector<uint8_t> &out) : out_(out) {}
    void literals(const uint8_t *p, uint32_t n) {
        n_lit_ += n;
        while (n >= 16) { uint32_t c = n < 271 ? n : 271; op2(0xE0, c - 16); bytes(p, c); p += c; n -= c; }
        if (n) { op1(0xE0 | n); bytes(p, n); }
    }
    void match(const uint8_t *lit, uint32_t n_lit, uint32_t m, uint32_t d) {
        n_lit_ += n_lit; n_match_ += m;
        while (n_lit >= 16) { uint32_t c = n_lit < 271 ? n_lit : 271; op2(0xE0, c - 16); bytes(lit, c); lit += c; n_lit -= c; }
        if (n_lit >= 4) { op1(0xE0 | n_lit); bytes(lit, n_lit); lit += n_lit; n_lit = 0; }
        const uint32_t l = n_lit;
        uint32_t first = 10 - 2 * l;  // opc.rs:229-232
        if (first > m) first = m;
        m -= first;
        if (d == prev_d_) {
            if (l == 0) op1(0xF0 | first);                                         // SmlM
            else { op1(0x06 | ((first - 3) << 3) | (l << 6)); bytes(lit, l); }      // PreD
        } else if (d < 0x600) {                                                     // SmlD
            op1(((d >> 8) & 7) | ((first - 3) << 3) | (l << 6)); op1(d & 0xFF); bytes(lit, l);
        } else if (d >= 0x4000 || m == 0 || first + m > 0x22) {                     // LrgD
            op1(0x07 | ((first - 3) << 3) | (l << 6)); op1(d & 0xFF); op1(d >> 8); bytes(lit, l);
        } else {                                                                    // MedD carries `first` only (backend.rs:110-114)
            const uint32_t mm = first - 3;
            const uint32_t opu = ((mm >> 2) & 7) | (l << 3) | (5u << 5) | ((mm & 3) << 8) | (d << 10);
            op1(opu & 0xFF); op1((opu >> 8) & 0xFF); op1((opu >> 16) & 0xFF); bytes(lit, l);
        }
        prev_d_ = d;
        while (m > 15) { uint32_t c = m < 271 ? m : 271; op2(0xF0, c - 16); m -= c; }
        if (m) op1(0xF0 | m);
    }
    uint32_t n_raw() const { return n_lit_ + n_match_; }
  private:
    void op1(uint32_t b) { out_.push_back(uint8_t(b)); }
    void op2(uint32_t a, uint32_t b) { out_.push_back(uint8_t(a)); out_.push_back(uint8_t(b)); }
    void bytes(const uint8_t *p, uint32_t n) { out_.insert(out_.end(), p, p + n); }
    std::vector<uint8_t> &out_;
    uint32_t prev_d_ = 0, n_lit_ = 0, n_match_ = 0;
};

void put32(std::vector<uint8_t> &o, uint32_t v) { for (int k = 0; k < 4; k++) o.push_back(uint8_t(v >> (8 * k))); }

void raw_block(std::vector<uint8_t> &o, const uint8_t *src, uint32_t n) {  // raw/ops.rs:19-29
    put32(o, 0x2D787662u); put32(o, n); o.insert(o.end(), src, src + n);
}

// match_inc_coarse::<4> (ring/object.rs:39-84) on a flat, zero-padded copy of the input: the true common length while
// it is below 12 + 32 (K + 1), K = ceil((max - 12) / 32), else max
uint32_t coarse_inc4(const uint8_t *b, uint32_t i, uint32_t c, uint32_t max) {
    const uint32_t K = max > 12 ? (max - 12 + 31) / 32 : 0, thr = 12 + 32 * (K + 1);
    uint32_t len = 4;
    while (len < thr && b[i + len] == b[c + len]) len++;
    return len < thr ? len : max;
}

void vn_block(std::vector<uint8_t> &o, const uint8_t *input, uint32_t n, bool ring) {
    // (ring: the loads and compares past the end of the input see zeros, as in a fresh RingBox, ring/ring_box.rs:9-17)
    std::vector<uint8_t> padded;
    const uint8_t *src = input;
    if (ring) { padded.assign(size_t(n) + 128, 0); std::memcpy(padded.data(), input, n); src = padded.data(); }
    const size_t mark = o.size();
    o.insert(o.end(), 12, 0);
    VnWriter w(o);
    VnHistory hist;
    Candidate pending;
    uint32_t lit = 0;
    const uint32_t end = ring ? n - 2 : n - 3;   // frontend_ring.rs:412 (tail - MATCH_UNIT + 1) / frontend_bytes.rs:172
    auto emit = [&](const Candidate &c) { w.match(src + lit, c.idx - lit, c.len, c.idx - c.midx); lit = c.idx + c.len; };
    for (uint32_t i = 0;;) {
        const uint32_t v = load32(src + i);
        uint32_t qv[4], qp[4];
        hist.push(v, i, qv, qp);
        Candidate in;  // find_match (frontend_bytes.rs:214-244) for the Vn match unit
        for (int k = 0; k < 4; k++) {
            if (i - qp[k] > kVnMaxDist) break;
            uint32_t x = v ^ qv[k], len = 0;
            if (x == 0) {
                if (ring) len = coarse_inc4(src, i, qp[k], n - i);   // frontend_ring.rs:493-501
                else { len = 4; while (len < n - i && src[i + len] == src[qp[k] + len]) len++; }
            }
            else if ((x & 0x00FFFFFFu) == 0) len = 3;
            if (len > in.len) { in.len = len; in.midx = qp[k]; }
        }
        if (ring && in.len > n - i) in.len = n - i;   // frontend_ring.rs:479-481
        bool emitted = false;
        if (in.len) {
            in.idx = i;
            uint32_t room = i - lit, b = 0;
            if (room > in.midx) room = in.midx;
            while (b < room && src[i - b - 1] == src[in.midx - b - 1]) b++;
            in.idx -= b; in.midx -= b; in.len += b;
            // Match::select::<40> (match_object.rs:12-33)
            if (in.len >= kGoodMatch) { emit(in); pending.len = 0; emitted = true; }
            else if (pending.len == 0) pending = in;
            else if (pending.idx + pending.len <= in.idx) { emit(pending); pending = in; emitted = true; }
            else if (in.len > pending.len) { emit(in); pending.len = 0; emitted = true; }
            else { emit(pending); pending.len = 0; emitted = true; }
        }
        if (emitted) {
            if (lit >= end) break;
            i++;
            for (; i < lit; i++) { uint32_t a[4], bq[4]; hist.push(load32(src + i), i, a, bq); }  // sync_history
            if (i >= end) break;
        } else {
            if (++i == end) break;
        }
    }
    if (pending.len) emit(pending);
    if (n - lit) w.literals(src + lit, n - lit);
    const uint8_t eos[8] = {0x06, 0, 0, 0, 0, 0, 0, 0};
    o.insert(o.end(), eos, eos + 8);
    const uint32_t payload = uint32_t(o.size() - mark) - 12;
    uint8_t hdr[12];
    const uint32_t magic = 0x6E787662u, raw = w.n_raw();
    std::memcpy(hdr, &magic, 4); std::memcpy(hdr + 4, &raw, 4); std::memcpy(hdr + 8, &payload, 4);
    std::memcpy(o.data() + mark, hdr, 12);
    if (size_t(n) + 8 <= o.size() - mark) {  // not smaller than a raw block (frontend_bytes.rs:92-99, frontend_ring.rs:323-330)
        o.resize(mark);
        raw_block(o, input, n);
    }
}

}  // namespace

namespace lzmi {
int encode_small(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_len, bool ring);
}

extern "C" int lzfse_mi_encode_small(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_len) {
    return lzmi::encode_small(src, n, dst, cap, out_len, false);
}

int lzmi::encode_small(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_len, bool ring) {
    if (!out_len || (!src && n) || n > kVnCutoff) return LZFSE_MI_BAD_ARGUMENT;
    std::vector<uint8_t> o;
    o.reserve(n + n / 4 + 64);
    if (n > kRawCutoff) vn_block(o, src, uint32_t(n), ring);
    else raw_block(o, src, uint32_t(n));
    put32(o, 0x24787662u);
    if (o.size() > cap) return LZFSE_MI_BUFFER_OVERFLOW;
    std::memcpy(dst, o.data(), o.size());
    *out_len = o.size();
    return LZFSE_MI_OK;
}
