/*
 * lzfse_oracle.c -- TEST INFRASTRUCTURE ONLY (see lzfse_oracle.h).
 *
 * Plain-C restatement of lzfse_rust v0.2.0's slice codec. Every function cites the
 * reference file:line (relative to /root/reference/) whose behaviour it restates.
 * Single-threaded, scalar, no SIMD: it is the checker, never the product.
 */
#include "lzfse_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ constants */
/* src/fse/constants.rs:22-69 */
#define LMDS_PER_BLOCK 10000u
#define LITERALS_PER_BLOCK 40000u
#define L_SYMBOLS 20
#define M_SYMBOLS 20
#define D_SYMBOLS 64
#define U_SYMBOLS 256
#define L_STATES 64u
#define M_STATES 64u
#define D_STATES 256u
#define U_STATES 1024u
#define MAX_L_VALUE 315u
#define MAX_M_VALUE 2359u
#define MAX_D_VALUE 262139u
#define N_WEIGHTS 360
#define V1_HEADER_SIZE 50u
#define V2_HEADER_SIZE 32u
#define V1_WEIGHT_PAYLOAD_BYTES 722u
#define V2_WEIGHT_PAYLOAD_BYTES_MAX 630u
/* src/encode/constants.rs:3-10 */
#define GOOD_MATCH_LEN 40u
#define RAW_CUTOFF 20u
#define RAW_LIMIT 0x4000u
#define VN_CUTOFF 0x1000u
/* src/encode/history.rs:10-13 */
#define HASH_BITS 14
#define HASH_WIDTH 4
/* src/base/magic_bytes.rs:3-7 */
#define MAGIC_EOS 0x24787662u
#define MAGIC_RAW 0x2D787662u
#define MAGIC_VX1 0x31787662u
#define MAGIC_VX2 0x32787662u
#define MAGIC_VXN 0x6E787662u
/* src/vn/constants.rs:1-7 */
#define VN_MAX_D 65535u
#define VN_HEADER_SIZE 12u

/* src/fse/constants.rs:127-134,159-166 */
static const uint8_t L_EXTRA_BITS[L_SYMBOLS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                                0, 0, 0, 0, 0, 0, 2, 3, 5, 8};
static const uint32_t L_BASE_VALUE[L_SYMBOLS] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,
                                                 10, 11, 12, 13, 14, 15, 16, 20, 28, 60};
static const uint8_t M_EXTRA_BITS[M_SYMBOLS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                                0, 0, 0, 0, 0, 0, 3, 5, 8, 11};
static const uint32_t M_BASE_VALUE[M_SYMBOLS] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,
                                                 10, 11, 12, 13, 14, 15, 16, 24, 56, 312};
/* src/fse/constants.rs:305-321: D extra bits = symbol/4, bases cumulative from 0. */
static uint8_t D_EXTRA_BITS[D_SYMBOLS];
static uint32_t D_BASE_VALUE[D_SYMBOLS];
/* value -> symbol (src/fse/constants.rs:137-157,169-303,323-353): the largest symbol
 * whose base <= value; the reference uses dense lookup tables with the same content. */
static uint8_t L_SYM_OF[MAX_L_VALUE + 1];
static uint8_t M_SYM_OF[MAX_M_VALUE + 1];
static int g_tables_ready;

static void init_tables(void) {
    if (g_tables_ready) return;
    uint32_t base = 0;
    for (int i = 0; i < D_SYMBOLS; i++) {
        D_EXTRA_BITS[i] = (uint8_t)(i / 4);
        D_BASE_VALUE[i] = base;
        base += 1u << (i / 4);
    }
    for (uint32_t v = 0; v <= MAX_L_VALUE; v++) {
        int s = 0;
        for (int i = 0; i < L_SYMBOLS; i++)
            if (L_BASE_VALUE[i] <= v) s = i;
        L_SYM_OF[v] = (uint8_t)s;
    }
    for (uint32_t v = 0; v <= MAX_M_VALUE; v++) {
        int s = 0;
        for (int i = 0; i < M_SYMBOLS; i++)
            if (M_BASE_VALUE[i] <= v) s = i;
        M_SYM_OF[v] = (uint8_t)s;
    }
    g_tables_ready = 1;
}

/* src/fse/constants.rs:323-353 (d_index + D_BASE_FROM_VALUE) */
static int d_sym_of(uint32_t v) {
    int lo = 0, hi = D_SYMBOLS - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (D_BASE_VALUE[mid] <= v)
            lo = mid;
        else
            hi = mid - 1;
    }
    return lo;
}

static inline uint32_t ld32(const uint8_t *p) {
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}
static inline uint64_t ld64(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}
static inline uint16_t ld16(const uint8_t *p) {
    uint16_t v;
    memcpy(&v, p, 2);
    return v;
}
static inline void st32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }
static inline void st64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); }
static inline int clz32(uint32_t v) { return v ? __builtin_clz(v) : 32; }

/* growable output used by the encoder side (the reference appends to a Vec<u8>) */
typedef struct {
    uint8_t *p;
    size_t len, cap;
} vec_t;

static int vec_reserve(vec_t *v, size_t extra) {
    if (v->len + extra <= v->cap) return 1;
    size_t nc = v->cap ? v->cap : 4096;
    while (nc < v->len + extra) nc *= 2;
    uint8_t *np = (uint8_t *)realloc(v->p, nc);
    if (!np) return 0;
    v->p = np;
    v->cap = nc;
    return 1;
}
static int vec_put(vec_t *v, const void *src, size_t n) {
    if (!vec_reserve(v, n)) return 0;
    memcpy(v->p + v->len, src, n);
    v->len += n;
    return 1;
}
static int vec_put32(vec_t *v, uint32_t x) { return vec_put(v, &x, 4); }

/* ================================================================== FSE weights */

/* src/fse/weights.rs:240-278 (normalize_m1, _coarse, _trim) */
void lzo_normalize_m1(uint16_t *w, uint32_t n, uint32_t in_total, uint32_t out_total) {
    if (in_total == 0) return; /* coarse returns (0,0); -0 < w[0]/4 is false when w[0]==0,
                                  trim(0) is a no-op; else w[0] += 0 */
    uint32_t shift = (uint32_t)clz32(out_total);
    uint32_t multiply = (1u << 31) / in_total;
    uint32_t round = 1u << (shift - 1);
    uint32_t max_weight = 0, max_index = 0;
    int32_t remaining = (int32_t)out_total;
    for (uint32_t i = 0; i < n; i++) {
        if (w[i] == 0) continue;
        uint32_t f = ((uint32_t)w[i] * multiply + round) >> shift;
        if (f == 0) f = 1;
        w[i] = (uint16_t)f;
        remaining -= (int32_t)f;
        if (f > max_weight) {
            max_weight = f;
            max_index = i;
        }
    }
    if (-remaining < (int32_t)w[max_index] / 4) {
        w[max_index] = (uint16_t)((int32_t)w[max_index] + remaining);
    } else {
        uint32_t overflow = (uint32_t)(-remaining);
        for (int s = 3; s >= 0; s--) {
            for (uint32_t i = 0; i < n; i++) {
                if (overflow == 0) break;
                if (w[i] == 0) continue;
                uint32_t k = ((uint32_t)w[i] - 1) >> s;
                if (k > overflow) k = overflow;
                w[i] = (uint16_t)(w[i] - k);
                overflow -= k;
            }
        }
    }
}

/* src/fse/weight_encoder.rs:23-37 + src/fse/weights.rs:139-163 */
uint32_t lzo_weights_store_v2(const uint16_t *weights, uint8_t *dst) {
    uint64_t accum = 0;
    uint32_t accum_bits = 0, i = 0;
    for (int k = 0; k < N_WEIGHTS; k++) {
        uint32_t w = weights[k], u, nb;
        switch (w) {
        case 0: u = 0, nb = 2; break;
        case 1: u = 2, nb = 2; break;
        case 2: u = 1, nb = 3; break;
        case 3: u = 5, nb = 3; break;
        case 4: u = 3, nb = 5; break;
        case 5: u = 11, nb = 5; break;
        case 6: u = 19, nb = 5; break;
        case 7: u = 27, nb = 5; break;
        default:
            if (w < 24)
                u = ((w - 8) << 4) + 7, nb = 8;
            else
                u = ((w - 24) << 4) + 15, nb = 14;
        }
        accum |= (uint64_t)u << accum_bits;
        accum_bits += nb;
        while (accum_bits >= 8) {
            dst[i++] = (uint8_t)accum;
            accum >>= 8;
            accum_bits -= 8;
        }
    }
    if (accum_bits > 0) dst[i++] = (uint8_t)accum;
    return i;
}

/* src/fse/constants.rs:115-124 */
static const uint8_t WEIGHTS_BITS_TABLE[32] = {2, 3, 2, 5, 2, 3, 2, 8, 2, 3, 2, 5, 2, 3, 2, 14,
                                               2, 3, 2, 5, 2, 3, 2, 8, 2, 3, 2, 5, 2, 3, 2, 14};
static const int8_t WEIGHTS_VALUE_TABLE[32] = {0, 2, 1, 4, 0, 3, 1, -1, 0, 2, 1, 5, 0, 3, 1, -1,
                                               0, 2, 1, 6, 0, 3, 1, -1, 0, 2, 1, 7, 0, 3, 1, -1};

/* src/fse/weights.rs:189-200 */
static int weights_check_totals(const uint16_t *w) {
    uint32_t t = 0;
    int i = 0;
    for (t = 0; i < 20; i++) t += w[i];
    if (t > L_STATES) return 0;
    for (t = 0; i < 40; i++) t += w[i];
    if (t > M_STATES) return 0;
    for (t = 0; i < 104; i++) t += w[i];
    if (t > D_STATES) return 0;
    for (t = 0; i < 360; i++) t += w[i];
    if (t > U_STATES) return 0;
    return 1;
}

/* src/fse/weights.rs:83-105 + src/fse/weight_encoder.rs:10-20 */
int lzo_weights_load_v2(const uint8_t *src, uint32_t n, uint16_t *weights) {
    uint64_t accum = 0;
    int64_t accum_bits = 0;
    uint32_t i = 0;
    for (int k = 0; k < N_WEIGHTS; k++) {
        while (i != n && accum_bits <= 24) {
            accum |= (uint64_t)src[i] << accum_bits;
            accum_bits += 8;
            i++;
        }
        uint32_t index = (uint32_t)accum & 0x1F;
        uint32_t nb = WEIGHTS_BITS_TABLE[index], w;
        if (nb == 8)
            w = 8 + (((uint32_t)accum >> 4) & 0xF);
        else if (nb == 14)
            w = 24 + (((uint32_t)accum >> 4) & 0x3FF);
        else
            w = (uint32_t)WEIGHTS_VALUE_TABLE[index];
        weights[k] = (uint16_t)w;
        accum >>= nb;
        accum_bits -= nb;
    }
    if (accum_bits < 0) return LZO_FSE_WEIGHT_PAYLOAD_UNDERFLOW;
    if (accum_bits >= 8 || i != n) return LZO_FSE_WEIGHT_PAYLOAD_OVERFLOW;
    if (!weights_check_totals(weights)) return LZO_FSE_BAD_WEIGHT_PAYLOAD;
    return LZO_OK;
}

/* ================================================================== FSE encoder */

typedef struct {
    int16_t t_k, t_w;
} eentry;

/* src/fse/encoder.rs:219-240 */
static void build_e_table(const uint16_t *weights, int n_sym, uint32_t n_states, eentry *table) {
    int n_clz = clz32(n_states);
    uint32_t total = 0;
    for (int i = 0; i < n_sym; i++) {
        uint32_t w = weights[i];
        eentry e;
        if (w == 0) {
            e.t_k = (int16_t)(-(int32_t)n_states);
            e.t_w = 0;
        } else {
            int k = clz32(w) - n_clz;
            e.t_k = (int16_t)(1024 * k - (int32_t)(w << k));
            e.t_w = (int16_t)((int32_t)n_states + (int32_t)total - (int32_t)w);
        }
        table[i] = e;
        total += w;
    }
}

/* src/bits/bit_writer.rs:8-58 over src/bits/bit_dst.rs:44-60 */
typedef struct {
    uint64_t accum;
    int accum_bits;
    vec_t *out;
} bitw;

static inline void bw_push(bitw *w, uint64_t bits, int n) {
    w->accum |= bits << w->accum_bits;
    w->accum_bits += n;
}
static inline void bw_flush(bitw *w) {
    int nbytes = w->accum_bits / 8;
    memcpy(w->out->p + w->out->len, &w->accum, 8); /* caller reserved >= 8 spare */
    w->out->len += (size_t)nbytes;
    w->accum = nbytes == 8 ? 0 : w->accum >> (nbytes * 8);
    w->accum_bits -= nbytes * 8;
}
static inline int bw_finalize(bitw *w) {
    int nbytes = (w->accum_bits + 7) / 8;
    memcpy(w->out->p + w->out->len, &w->accum, 8);
    w->out->len += (size_t)nbytes;
    return nbytes * 8 - w->accum_bits;
}

/* src/fse/encoder.rs:191-199 (EEntry::encode) */
static inline void e_encode(const eentry e, bitw *w, uint32_t *state) {
    uint32_t s = *state;
    uint32_t nb = (uint32_t)((int32_t)e.t_k + (int32_t)s) >> 10;
    *state = (uint32_t)((int32_t)e.t_w + ((int32_t)s >> nb));
    bw_push(w, s & ((1u << nb) - 1), (int)nb);
}

typedef struct {
    uint16_t l, m;
    uint32_t d; /* zeroed when equal to previous (src/fse/buffer.rs:107-117) */
} lmdpack;

/* src/fse/buffer.rs:16-23 + literals.rs / lmds.rs buffers */
typedef struct {
    uint8_t literals[LITERALS_PER_BLOCK + MAX_L_VALUE + 32];
    uint32_t n_literals;
    lmdpack lmds[LMDS_PER_BLOCK];
    uint32_t n_lmds;
    uint32_t n_match_bytes;
    uint32_t match_distance;
    const lzo_trace *trace;
} fse_buffer;

/* src/fse/buffer.rs:119-125 */
static void buffer_reset(fse_buffer *b) {
    b->n_literals = 0;
    b->n_lmds = 0;
    b->n_match_bytes = 0;
    b->match_distance = 0;
}

/* src/fse/buffer.rs:99-104 */
static void buffer_push_l(fse_buffer *b, uint16_t l) {
    b->match_distance = 1;
    lmdpack p = {l, 0, 1};
    b->lmds[b->n_lmds++] = p;
    if (b->trace && b->trace->pack) b->trace->pack(b->trace->ctx, l, 0, 1);
}

/* src/fse/buffer.rs:106-117 */
static void buffer_push_lmd(fse_buffer *b, uint16_t l, uint16_t m, uint32_t d) {
    if (b->match_distance == d) {
        d = 0;
    } else {
        b->match_distance = d;
    }
    lmdpack p = {l, m, d};
    b->lmds[b->n_lmds++] = p;
    b->n_match_bytes += m;
    if (b->trace && b->trace->pack) b->trace->pack(b->trace->ctx, l, m, d);
}

/* src/fse/buffer.rs:45-97. `lit`/`n_lit`/`match_len` are consumed in place. */
static int buffer_push(fse_buffer *b, const uint8_t **lit, uint32_t *n_lit, uint32_t *match_len,
                       uint32_t match_distance) {
    while (*n_lit > MAX_L_VALUE) {
        if (b->n_lmds == LMDS_PER_BLOCK) return 0;
        uint32_t limit = LITERALS_PER_BLOCK - b->n_literals;
        if (MAX_L_VALUE <= limit) {
            memcpy(b->literals + b->n_literals, *lit, MAX_L_VALUE);
            b->n_literals += MAX_L_VALUE;
            *lit += MAX_L_VALUE;
            *n_lit -= MAX_L_VALUE;
            buffer_push_l(b, (uint16_t)MAX_L_VALUE);
        } else if (limit != 0) {
            memcpy(b->literals + b->n_literals, *lit, limit);
            b->n_literals += limit;
            *lit += limit;
            *n_lit -= limit;
            buffer_push_l(b, (uint16_t)limit);
            return 0;
        } else {
            return 0;
        }
    }
    if (b->n_lmds == LMDS_PER_BLOCK) return 0;
    uint32_t literal_len = *n_lit;
    uint32_t limit = LITERALS_PER_BLOCK - b->n_literals;
    if (literal_len <= limit) {
        memcpy(b->literals + b->n_literals, *lit, literal_len);
        b->n_literals += literal_len;
        *lit += literal_len;
        *n_lit = 0;
    } else if (limit != 0) {
        memcpy(b->literals + b->n_literals, *lit, limit);
        b->n_literals += limit;
        *lit += limit;
        *n_lit -= limit;
        buffer_push_l(b, (uint16_t)limit);
        return 0;
    } else {
        return 0;
    }
    while (*match_len > MAX_M_VALUE) {
        buffer_push_lmd(b, (uint16_t)literal_len, (uint16_t)MAX_M_VALUE, match_distance);
        *match_len -= MAX_M_VALUE;
        literal_len = 0;
        if (b->n_lmds == LMDS_PER_BLOCK) return 0;
    }
    buffer_push_lmd(b, (uint16_t)literal_len, (uint16_t)*match_len, match_distance);
    *match_len = 0;
    return 1;
}

/* src/fse/backend.rs:39-54 (emit_block_v2) with
 * literals.rs:93-145, lmds.rs:62-93, weights.rs:25-64, block.rs:168-196 */
static int emit_block_v2(fse_buffer *b, vec_t *dst) {
    init_tables();
    if (b->trace && b->trace->block)
        b->trace->block(b->trace->ctx, b->n_lmds, b->n_literals, b->n_literals + b->n_match_bytes);
    size_t mark = dst->len;
    /* worst case: 32 + 630 + 50000 + 67508 (+ slack for 8-byte accumulator stores) */
    if (!vec_reserve(dst, 32 + 630 + 50008 + 67516 + 64)) return LZO_IO;
    memset(dst->p + dst->len, 0, V2_HEADER_SIZE);
    dst->len += V2_HEADER_SIZE;
    /* literals.rs:136-145 pad with literals[0] */
    memset(b->literals + b->n_literals, b->literals[0], 4);
    /* weights.rs:25-64 */
    uint16_t w[N_WEIGHTS];
    memset(w, 0, sizeof w);
    if (b->n_lmds) {
        for (uint32_t i = 0; i < b->n_lmds; i++) {
            w[L_SYM_OF[b->lmds[i].l]]++;
            w[20 + M_SYM_OF[b->lmds[i].m]]++;
            w[40 + d_sym_of(b->lmds[i].d)]++;
        }
        lzo_normalize_m1(w, 20, b->n_lmds, L_STATES);
        lzo_normalize_m1(w + 20, 20, b->n_lmds, M_STATES);
        lzo_normalize_m1(w + 40, 64, b->n_lmds, D_STATES);
    }
    if (b->n_literals) {
        for (uint32_t i = 0; i < b->n_literals; i++) w[104 + b->literals[i]]++;
        lzo_normalize_m1(w + 104, 256, b->n_literals, U_STATES);
    }
    uint32_t n_weight_bytes = lzo_weights_store_v2(w, dst->p + dst->len);
    dst->len += n_weight_bytes;
    /* encoder.rs:21-26 */
    eentry el[L_SYMBOLS], em[M_SYMBOLS], ed[D_SYMBOLS], eu[U_SYMBOLS];
    build_e_table(w, L_SYMBOLS, L_STATES, el);
    build_e_table(w + 20, M_SYMBOLS, M_STATES, em);
    build_e_table(w + 40, D_SYMBOLS, D_STATES, ed);
    build_e_table(w + 104, U_SYMBOLS, U_STATES, eu);
    /* literals.rs:93-133 */
    uint32_t n4 = (b->n_literals + 3) / 4 * 4;
    size_t lit_mark = dst->len;
    bitw bw = {0, 0, dst};
    uint32_t s0 = U_STATES, s1 = U_STATES, s2 = U_STATES, s3 = U_STATES;
    for (uint32_t i = n4; i != 0; i -= 4) {
        e_encode(eu[b->literals[i - 1]], &bw, &s3);
        e_encode(eu[b->literals[i - 2]], &bw, &s2);
        e_encode(eu[b->literals[i - 3]], &bw, &s1);
        e_encode(eu[b->literals[i - 4]], &bw, &s0);
        bw_flush(&bw);
    }
    uint32_t lit_bits = (uint32_t)bw_finalize(&bw);
    uint32_t lit_payload = (uint32_t)(dst->len - lit_mark);
    /* lmds.rs:62-93 */
    size_t lmd_mark = dst->len;
    memset(dst->p + dst->len, 0, 8);
    dst->len += 8;
    bitw bl = {0, 0, dst};
    uint32_t sl = L_STATES, sm = M_STATES, sd = D_STATES;
    for (uint32_t i = b->n_lmds; i != 0; i--) {
        lmdpack p = b->lmds[i - 1];
        int sym = d_sym_of(p.d);
        bw_push(&bl, p.d - D_BASE_VALUE[sym], D_EXTRA_BITS[sym]);
        e_encode(ed[sym], &bl, &sd);
        sym = M_SYM_OF[p.m];
        bw_push(&bl, p.m - M_BASE_VALUE[sym], M_EXTRA_BITS[sym]);
        e_encode(em[sym], &bl, &sm);
        sym = L_SYM_OF[p.l];
        bw_push(&bl, p.l - L_BASE_VALUE[sym], L_EXTRA_BITS[sym]);
        e_encode(el[sym], &bl, &sl);
        bw_flush(&bl);
    }
    uint32_t lmd_bits = (uint32_t)bw_finalize(&bl);
    uint32_t lmd_payload = (uint32_t)(dst->len - lmd_mark);
    /* block.rs:168-196 */
    uint8_t *h = dst->p + mark;
    st32(h, MAGIC_VX2);
    st32(h + 4, b->n_literals + b->n_match_bytes);
    uint64_t p = 0;
    p |= (uint64_t)n4;
    p |= (uint64_t)lit_payload << 20;
    p |= (uint64_t)b->n_lmds << 40;
    p |= (uint64_t)(7 - lit_bits) << 60;
    st64(h + 8, p);
    p = 0;
    p |= (uint64_t)(s0 - U_STATES);
    p |= (uint64_t)(s1 - U_STATES) << 10;
    p |= (uint64_t)(s2 - U_STATES) << 20;
    p |= (uint64_t)(s3 - U_STATES) << 30;
    p |= (uint64_t)lmd_payload << 40;
    p |= (uint64_t)(7 - lmd_bits) << 60;
    st64(h + 16, p);
    p = 0;
    p |= (uint64_t)(V2_HEADER_SIZE + n_weight_bytes);
    p |= (uint64_t)(sl - L_STATES) << 32;
    p |= (uint64_t)(sm - M_STATES) << 42;
    p |= (uint64_t)(sd - D_STATES) << 52;
    st64(h + 24, p);
    buffer_reset(b);
    return LZO_OK;
}

/* ================================================================== VN backend */

/* src/vn/backend.rs:26-31 */
typedef struct {
    size_t mark;
    uint32_t match_distance, n_literals, n_match_bytes;
} vn_backend;

static int vn_l(vec_t *dst, const uint8_t **lit, uint32_t *n_lit, uint32_t len, uint32_t opu,
                uint32_t op_len) {
    /* backend.rs:146-163 */
    if (!vec_reserve(dst, op_len + len + 8)) return 0;
    memcpy(dst->p + dst->len, &opu, op_len);
    memcpy(dst->p + dst->len + op_len, *lit, len);
    dst->len += op_len + len;
    *lit += len;
    *n_lit -= len;
    return 1;
}

/* backend.rs:216-236: opcode bytes then up to 3 literal bytes */
static int vn_lmd(vec_t *dst, const uint8_t **lit, uint32_t *n_lit, uint32_t literal_len,
                  uint32_t opu, uint32_t op_len) {
    if (!vec_reserve(dst, op_len + literal_len + 8)) return 0;
    memcpy(dst->p + dst->len, &opu, op_len);
    memcpy(dst->p + dst->len + op_len, *lit, literal_len);
    dst->len += op_len + literal_len;
    *lit += literal_len;
    *n_lit -= literal_len;
    return 1;
}

static int vn_m(vec_t *dst, uint32_t opu, uint32_t op_len) {
    if (!vec_reserve(dst, op_len + 8)) return 0;
    memcpy(dst->p + dst->len, &opu, op_len);
    dst->len += op_len;
    return 1;
}

/* src/vn/opc.rs encoders */
static uint32_t opc_sml_l(uint32_t l) { return 0xE0u | l; }                   /* opc.rs:7-14 */
static uint32_t opc_lrg_l(uint32_t l) { return 0xE0u | ((l - 0x10) << 8); }   /* :31-38 */
static uint32_t opc_sml_m(uint32_t m) { return 0xF0u | m; }                   /* :55-62 */
static uint32_t opc_lrg_m(uint32_t m) { return 0xF0u | ((m - 0x10) << 8); }   /* :79-86 */
static uint32_t opc_pre_d(uint32_t l, uint32_t m) {                           /* :104-114 */
    return 0x6u | ((m - 3) << 3) | (l << 6);
}
static uint32_t opc_sml_d(uint32_t l, uint32_t m, uint32_t d) { /* :136-148 */
    return ((d >> 8) & 7) | ((m - 3) << 3) | (l << 6) | ((d & 0xFF) << 8);
}
static uint32_t opc_med_d(uint32_t l, uint32_t m, uint32_t d) { /* :172-186 */
    m -= 3;
    return ((m >> 2) & 7) | (l << 3) | (0x5u << 5) | ((m & 3) << 8) | (d << 10);
}
static uint32_t opc_lrg_d(uint32_t l, uint32_t m, uint32_t d) { /* :213-225 */
    return 0x7u | ((m - 3) << 3) | (l << 6) | (d << 8);
}

/* src/vn/backend.rs:57-74 */
static int vn_push_literals(vn_backend *b, vec_t *dst, const uint8_t *lit, uint32_t n_lit) {
    b->n_literals += n_lit;
    while (n_lit >= 0x10) {
        uint32_t len = n_lit < 0x10F ? n_lit : 0x10F;
        if (!vn_l(dst, &lit, &n_lit, len, opc_lrg_l(len), 2)) return 0;
    }
    if (n_lit > 0) {
        uint32_t len = n_lit;
        if (!vn_l(dst, &lit, &n_lit, len, opc_sml_l(len), 1)) return 0;
    }
    return 1;
}

/* src/vn/backend.rs:76-125 */
static int vn_push_match(vn_backend *b, vec_t *dst, const uint8_t *lit, uint32_t n_lit,
                         uint32_t match_len, uint32_t match_distance) {
    b->n_literals += n_lit;
    b->n_match_bytes += match_len;
    while (n_lit >= 0x10) {
        uint32_t len = n_lit < 0x10F ? n_lit : 0x10F;
        if (!vn_l(dst, &lit, &n_lit, len, opc_lrg_l(len), 2)) return 0;
    }
    if (n_lit >= 4) {
        uint32_t len = n_lit;
        if (!vn_l(dst, &lit, &n_lit, len, opc_sml_l(len), 1)) return 0;
    }
    uint32_t literal_len = n_lit;
    uint32_t n = 0x0A - 2 * literal_len; /* opc.rs:229-232 match_len_x */
    if (n > match_len) n = match_len;
    match_len -= n;
    int ok;
    if (match_distance == b->match_distance) {
        if (literal_len == 0)
            ok = vn_m(dst, opc_sml_m(n), 1);
        else
            ok = vn_lmd(dst, &lit, &n_lit, literal_len, opc_pre_d(literal_len, n), 1);
    } else if (match_distance < 0x600) {
        ok = vn_lmd(dst, &lit, &n_lit, literal_len, opc_sml_d(literal_len, n, match_distance), 2);
    } else if (match_distance >= 0x4000 || match_len == 0 || n + match_len > 0x22) {
        ok = vn_lmd(dst, &lit, &n_lit, literal_len, opc_lrg_d(literal_len, n, match_distance), 3);
    } else {
        /* quirk kept: MedD carries n, not n + rest (backend.rs:110-114) */
        ok = vn_lmd(dst, &lit, &n_lit, literal_len, opc_med_d(literal_len, n, match_distance), 3);
    }
    if (!ok) return 0;
    b->match_distance = match_distance;
    while (match_len > 0x0F) {
        uint32_t limit = match_len < 0x10F ? match_len : 0x10F;
        if (!vn_m(dst, opc_lrg_m(limit), 2)) return 0;
        match_len -= limit;
    }
    if (match_len > 0)
        if (!vn_m(dst, opc_sml_m(match_len), 1)) return 0;
    return 1;
}

/* ================================================================== frontend */

typedef struct {
    uint32_t val, idx;
} item_t;
typedef struct {
    item_t q[HASH_WIDTH];
} history_t;

typedef struct {
    uint32_t idx, match_idx, match_len;
} match_t;

typedef struct {
    int vn;             /* backend type: 0 = Fse (fse/object.rs:23-44), 1 = Vn (vn/object.rs:23-60) */
    uint32_t max_dist;
    const uint8_t *src;
    size_t n;
    history_t *table;
    match_t pending;
    uint32_t literal_index;
    fse_buffer *fse;
    vn_backend vnb;
    vec_t *dst;
    const lzo_trace *trace;
    int err;
} frontend;

/* history.rs:221-224 with MatchUnit::hash_u (fse/object.rs:38-43, vn/object.rs:31-46) */
static inline uint32_t bucket_of(int vn, uint32_t val) {
    if (vn) val &= 0x00FFFFFFu;
    return (val * 0x9E3779B1u) >> (32 - HASH_BITS);
}

/* history.rs:24-31,110-118 */
static inline history_t table_push(history_t *table, int vn, uint32_t val, uint32_t idx) {
    history_t *q = &table[bucket_of(vn, val)];
    history_t copy = *q;
    q->q[3] = q->q[2];
    q->q[2] = q->q[1];
    q->q[1] = q->q[0];
    q->q[0].val = val;
    q->q[0].idx = idx;
    return copy;
}

/* match_kit/match_fast.rs:22-49 */
static uint32_t match_inc(const uint8_t *b, size_t index, size_t match_index, size_t len,
                          size_t max) {
    while (len + 8 <= max) {
        uint64_t x = ld64(b + index + len) ^ ld64(b + match_index + len);
        if (x) return (uint32_t)(len + (size_t)(__builtin_ctzll(x) >> 3));
        len += 8;
    }
    while (len < max) {
        if (b[index + len] != b[match_index + len]) return (uint32_t)len;
        len++;
    }
    return (uint32_t)max;
}

/* match_kit/match_fast.rs:61-89 */
static uint32_t match_dec(const uint8_t *b, size_t index, size_t match_index, size_t max) {
    size_t len = 0;
    while (len != max) {
        if (b[index - len - 1] != b[match_index - len - 1]) break;
        len++;
    }
    return (uint32_t)len;
}

/* frontend_bytes.rs:214-244 (forward part, shared with lzo_candidates) */
static match_t find_match_fwd(const frontend *f, const history_t *queue, uint32_t val,
                              uint32_t idx) {
    match_t m = {0, 0, 0};
    for (int c = 0; c < HASH_WIDTH; c++) {
        uint32_t distance = idx - queue->q[c].idx;
        if (distance > f->max_dist) break;
        uint32_t len;
        uint32_t x = val ^ queue->q[c].val;
        if (x == 0) {
            len = match_inc(f->src, idx, queue->q[c].idx, 4, f->n - idx);
        } else if (f->vn && (x & 0x00FFFFFFu) == 0) {
            len = 3;
        } else {
            len = 0;
        }
        if (len > m.match_len) {
            m.match_len = len;
            m.match_idx = queue->q[c].idx;
        }
    }
    m.idx = idx;
    return m;
}

/* frontend_bytes.rs:232-243,259-268 */
static match_t find_match(const frontend *f, const history_t *queue, uint32_t val, uint32_t idx) {
    match_t m = find_match_fwd(f, queue, val, idx);
    if (m.match_len == 0) return m;
    size_t literal_len = (size_t)idx - f->literal_index;
    size_t max = literal_len < m.match_idx ? literal_len : m.match_idx;
    uint32_t dec = match_dec(f->src, idx, m.match_idx, max);
    m.idx -= dec;
    m.match_idx -= dec;
    m.match_len += dec;
    return m;
}

/* match_object.rs:12-33; returns 1 and *out when a match is selected */
static int match_select(match_t *self, match_t incoming, match_t *out) {
    if (incoming.match_len == 0) return 0;
    if (incoming.match_len >= GOOD_MATCH_LEN) {
        *out = incoming;
        self->match_len = 0;
        return 1;
    }
    if (self->match_len == 0) {
        *self = incoming;
        return 0;
    }
    if (self->idx + self->match_len <= incoming.idx) {
        *out = *self;
        *self = incoming;
        return 1;
    }
    if (incoming.match_len > self->match_len) {
        *out = incoming;
        self->match_len = 0;
        return 1;
    }
    *out = *self;
    self->match_len = 0;
    return 1;
}

/* fse/backend.rs:76-90 and vn/backend.rs:76 */
static void backend_push_match(frontend *f, const uint8_t *lit, uint32_t n_lit, uint32_t match_len,
                               uint32_t dist) {
    if (f->err) return;
    if (f->vn) {
        if (!vn_push_match(&f->vnb, f->dst, lit, n_lit, match_len, dist)) f->err = LZO_IO;
        return;
    }
    for (;;) {
        if (buffer_push(f->fse, &lit, &n_lit, &match_len, dist)) break;
        int e = emit_block_v2(f->fse, f->dst);
        if (e) {
            f->err = e;
            return;
        }
    }
}

/* frontend_bytes.rs:287-302 */
static void push_match(frontend *f, match_t m) {
    uint32_t dist = m.idx - m.match_idx;
    if (f->trace && f->trace->match)
        f->trace->match(f->trace->ctx, f->literal_index, m.idx, m.match_len, dist);
    const uint8_t *lit = f->src + f->literal_index;
    uint32_t n_lit = m.idx - f->literal_index;
    f->literal_index = m.idx + m.match_len;
    backend_push_match(f, lit, n_lit, m.match_len, dist);
}

/* frontend_bytes.rs:121-211,271-375: match_blocks (match_any per block, reposition between blocks), flush_pending,
 * flush_literals. guide / slack: BLOCK_GUIDE = 0x7FFF_FFFF and SLACK = 0x1000_0000 (:19-23) -- parameters here so that the
 * tests can make the front end reposition on inputs of a few MiB (lzo_encode_guide); a slice of up to guide + 3 bytes is one
 * block and never repositions. Positions are relative to f->src, which moves up by `delta` at every reposition, as
 * self.src does (:372). */
#define BLOCK_GUIDE 0x7FFFFFFFu
#define BLOCK_SLACK 0x10000000u
static void frontend_finalize_guide(frontend *f, uint32_t guide, uint32_t slack) {
    size_t total = f->n;          /* bytes from f->src to the end of the input */
    uint32_t index = 0;           /* self.index between blocks: 0, then MAX_MATCH_DISTANCE (is_any :384-388) */
    for (;;) {
        const uint8_t *src = f->src;
        /* match_any :160-211 */
        int is_short = total <= (size_t)guide + 3;
        size_t block_len = is_short ? total : (size_t)guide;
        uint32_t end = is_short ? (uint32_t)block_len - 3 : (uint32_t)block_len - slack - 3;
        f->n = block_len;         /* matches end with the block (:253: max = self.block.len() - index) */
        for (;;) {
            uint32_t val = ld32(src + index);
            history_t queue = table_push(f->table, f->vn, val, index);
            match_t incoming = find_match(f, &queue, val, index);
            match_t sel;
            if (match_select(&f->pending, incoming, &sel)) {
                push_match(f, sel);
                if (f->err) return;
                if (f->literal_index >= end) break;
                index += 1;
                while (index < f->literal_index) { /* sync_history :336-344 */
                    table_push(f->table, f->vn, ld32(src + index), index);
                    index++;
                }
                if (index >= end) break;
            } else {
                index += 1;
                if (index == end) break;
            }
        }
        if (is_short) break;
        /* reposition :348-375. self.index is the block's limit here, whatever the loop's own `index` got to (match_any keeps a
         * local copy, :168,180): positions between the one the last match was found at and the limit are never pushed when
         * that match ran past the limit. */
        index = end;
        while (index < f->literal_index) { /* sync_history over self.src */
            table_push(f->table, f->vn, ld32(src + index), index);
            index++;
        }
        uint32_t delta = index - f->max_dist;
        if (f->literal_index < delta) {
            /* literals that have passed the buffer head go as they are, the pending match is dropped */
            f->pending.match_len = 0;
            uint32_t len = delta - f->literal_index;
            if (f->trace && f->trace->match)
                f->trace->match(f->trace->ctx, f->literal_index, f->literal_index + len, 0, 1);
            const uint8_t *lit = src + f->literal_index;
            f->literal_index += len;
            if (f->vn) {
                if (!vn_push_literals(&f->vnb, f->dst, lit, len)) f->err = LZO_IO;
            } else {
                backend_push_match(f, lit, len, 0, 1);
            }
            if (f->err) return;
        }
        /* history.rs:62-66,121-130 clamp_rebias: idx values are wrapping u32 */
        for (size_t b = 0; b < ((size_t)1 << HASH_BITS); b++)
            for (int k = 0; k < HASH_WIDTH; k++) {
                item_t *it = &f->table[b].q[k];
                if ((uint32_t)(index - it->idx) > 0x40000000u) it->idx = index - 0x40000000u - delta;
                else it->idx -= delta;
            }
        /* match_object.rs:35-38 */
        f->pending.match_idx -= delta;
        f->pending.idx -= delta;
        f->src += delta;
        total -= delta;
        f->literal_index -= delta;
        index -= delta;
    }
    const uint8_t *src = f->src;
    /* flush_pending :271-285 */
    if (f->pending.match_len != 0) {
        push_match(f, f->pending);
        f->pending.match_len = 0;
    }
    /* flush_literals :304-317 -> push_literals (fse/backend.rs:67-73 => M=0, D=1) */
    uint32_t len = (uint32_t)f->n - f->literal_index;
    if (len != 0 && !f->err) {
        if (f->trace && f->trace->match)
            f->trace->match(f->trace->ctx, f->literal_index, (uint32_t)f->n, 0, 1);
        const uint8_t *lit = src + f->literal_index;
        f->literal_index += len;
        if (f->vn) {
            if (!vn_push_literals(&f->vnb, f->dst, lit, len)) f->err = LZO_IO;
        } else {
            backend_push_match(f, lit, len, 0, 1);
        }
    }
}
static void frontend_finalize(frontend *f) { frontend_finalize_guide(f, BLOCK_GUIDE, BLOCK_SLACK); }

/* history.rs:72-84: every entry (val 0, idx Q0 - Q1 = 0xC000_0000) */
static void table_reset(history_t *t) {
    for (size_t i = 0; i < ((size_t)1 << HASH_BITS); i++)
        for (int k = 0; k < HASH_WIDTH; k++) {
            t[i].q[k].val = 0;
            t[i].q[k].idx = 0xC0000000u;
        }
}

/* raw/ops.rs:19-29 */
static int raw_compress(vec_t *dst, const uint8_t *src, size_t n) {
    if (!vec_put32(dst, MAGIC_RAW)) return 0;
    if (!vec_put32(dst, (uint32_t)n)) return 0;
    return vec_put(dst, src, n);
}

size_t lzo_encode_bound(size_t n) { return n + n / 2 + n / 4 + 4096; }

/* encoder.rs:49-53 -> frontend_bytes.rs:41-111 */
static int encode_guide(const uint8_t *src, size_t n, uint8_t *out, size_t cap, size_t *out_len, const lzo_trace *trace,
                        uint32_t guide, uint32_t slack);
int lzo_encode(const uint8_t *src, size_t n, uint8_t *out, size_t cap, size_t *out_len,
               const lzo_trace *trace) {
    return encode_guide(src, n, out, cap, out_len, trace, BLOCK_GUIDE, BLOCK_SLACK);
}
/* The same front end with another block guide and slack (test hook: repositions on small inputs). The reference's own
 * assertions (:166-168,359): 256 <= slack, 2 * slack <= guide, and the limit guide - slack - 3 >= MAX_MATCH_DISTANCE. */
int lzo_encode_guide(const uint8_t *src, size_t n, uint8_t *out, size_t cap, size_t *out_len, uint32_t guide, uint32_t slack) {
    if (slack < 256 || (uint64_t)slack * 2 > guide || guide > BLOCK_GUIDE || guide - slack - 3 < MAX_D_VALUE) return LZO_UNSUPPORTED;
    return encode_guide(src, n, out, cap, out_len, NULL, guide, slack);
}
static int encode_guide(const uint8_t *src, size_t n, uint8_t *out, size_t cap, size_t *out_len, const lzo_trace *trace,
                        uint32_t guide, uint32_t slack) {
    init_tables();
    vec_t dst = {0, 0, 0};
    int status = LZO_OK;
    history_t *table = NULL;
    fse_buffer *fb = NULL;
    if (n > RAW_CUTOFF) {
        table = (history_t *)malloc(sizeof(history_t) << HASH_BITS);
        if (!table) return LZO_IO;
        table_reset(table);
    }
    if (n > VN_CUTOFF) {
        fb = (fse_buffer *)malloc(sizeof(fse_buffer));
        if (!fb) {
            free(table);
            return LZO_IO;
        }
        buffer_reset(fb);
        fb->trace = trace;
        fb->literals[0] = 0;
        frontend f;
        memset(&f, 0, sizeof f);
        f.vn = 0;
        f.max_dist = MAX_D_VALUE;
        f.src = src;
        f.n = n;
        f.table = table;
        f.fse = fb;
        f.dst = &dst;
        f.trace = trace;
        frontend_finalize_guide(&f, guide, slack);
        if (!f.err) f.err = emit_block_v2(fb, &dst); /* fse/backend.rs:92-95 */
        status = f.err;
    } else if (n > RAW_CUTOFF) {
        frontend f;
        memset(&f, 0, sizeof f);
        f.vn = 1;
        f.max_dist = VN_MAX_D;
        f.src = src;
        f.n = n;
        f.table = table;
        f.dst = &dst;
        f.trace = trace;
        size_t mark = dst.len;
        /* vn/backend.rs:42-55 */
        uint8_t zero[VN_HEADER_SIZE] = {0};
        if (!vec_put(&dst, zero, VN_HEADER_SIZE)) f.err = LZO_IO;
        if (!f.err) frontend_finalize(&f);
        if (!f.err) {
            /* vn/backend.rs:127-135 */
            uint64_t eos = 0x06;
            if (!vec_put(&dst, &eos, 8)) f.err = LZO_IO;
        }
        if (!f.err) {
            uint32_t n_payload = (uint32_t)(dst.len - mark) - VN_HEADER_SIZE;
            st32(dst.p + mark, MAGIC_VXN);
            st32(dst.p + mark + 4, f.vnb.n_literals + f.vnb.n_match_bytes);
            st32(dst.p + mark + 8, n_payload);
            /* frontend_bytes.rs:92-99: raw fallback when not smaller */
            size_t dst_len = dst.len - mark;
            if (n < RAW_LIMIT && n + 8 <= dst_len) {
                dst.len = mark;
                if (!raw_compress(&dst, src, n)) f.err = LZO_IO;
            }
        }
        status = f.err;
    } else {
        if (!raw_compress(&dst, src, n)) status = LZO_IO;
    }
    if (status == LZO_OK && !vec_put32(&dst, MAGIC_EOS)) status = LZO_IO;
    if (status == LZO_OK) {
        if (dst.len > cap) {
            status = LZO_BUFFER_OVERFLOW;
        } else {
            memcpy(out, dst.p, dst.len);
            *out_len = dst.len;
        }
    }
    free(dst.p);
    free(table);
    free(fb);
    return status;
}

/* Stage-1 dump for GPU parity: forward-only find_match at every position. */
int lzo_candidates(const uint8_t *src, size_t n, uint32_t *match_idx, uint32_t *fwd_len) {
    if (n <= VN_CUTOFF || n > (size_t)0x7FFFFFFFu) return LZO_UNSUPPORTED;
    history_t *table = (history_t *)malloc(sizeof(history_t) << HASH_BITS);
    if (!table) return LZO_IO;
    table_reset(table);
    frontend f;
    memset(&f, 0, sizeof f);
    f.max_dist = MAX_D_VALUE;
    f.src = src;
    f.n = n;
    for (uint32_t i = 0; i + 4 <= n; i++) {
        uint32_t val = ld32(src + i);
        history_t queue = table_push(table, 0, val, i);
        match_t m = find_match_fwd(&f, &queue, val, i);
        match_idx[i] = m.match_len ? m.match_idx : 0xFFFFFFFFu;
        fwd_len[i] = m.match_len;
    }
    free(table);
    return LZO_OK;
}

/* Stage-0 dump for GPU parity: the candidate queue (history.rs:110-118 push returns the pre-insert row) of every
 * position as if all positions were inserted in order, which is what the encoder does (frontend_bytes.rs:187,336-344).
 * rows[4 i + c] = idx of candidate c (newest first), 0xFFFFFFFF for the reset sentinel (history.rs:72-84). */
int lzo_table_rows(const uint8_t *src, size_t n, uint32_t *rows) {
    if (n <= VN_CUTOFF || n > (size_t)0x7FFFFFFFu) return LZO_UNSUPPORTED;
    history_t *table = (history_t *)malloc(sizeof(history_t) << HASH_BITS);
    if (!table) return LZO_IO;
    table_reset(table);
    for (uint32_t i = 0; i + 4 <= n; i++) {
        uint32_t val = ld32(src + i);
        history_t queue = table_push(table, 0, val, i);
        for (int c = 0; c < HASH_WIDTH; c++)
            rows[4 * (size_t)i + c] = queue.q[c].idx >= 0x80000000u ? 0xFFFFFFFFu : queue.q[c].idx;
    }
    free(table);
    return LZO_OK;
}

/* ================================================================== ring frontend */
/* Restatement of the ring/stream encoder: LzfseRingEncoder::encode (encode/ring_encoder.rs:55-67),
 * LzfseWriter / LzfseWriterBytes (encode/writer.rs:39-75, writer_bytes.rs:44-78) over FrontendRing
 * (encode/frontend_ring.rs:30-687). Its parse is NOT the slice parse: a 512 KiB ring fed in 16 KiB blocks
 * (encode/constants.rs:23-33), matched in rounds (match_long) with a capped, "coarse" forward length, a backward
 * length that stops at the ring head, literals that pass the head pushed as they are (pending discarded), and a final
 * match_short over what is left. The ring itself is simulated byte for byte, shadow zones included, so the coarse
 * compares read exactly what the reference reads (also beyond `tail`, where the ring holds older data or, for a
 * fresh encoder, zeros: a RingBox starts zeroed, ring/ring_box.rs:9-17; a REUSED reference encoder would see the
 * previous stream's bytes there -- this oracle is a fresh LzfseRingEncoder per stream).
 * Parity pin: the KATs of frontend_ring.rs:768-993 (tests/test_oracle_ring.py). */

#define OVERMATCH_LEN 40u                     /* ring/object.rs:15 (5 * size_of::<usize>(), 64-bit target) */
#define OVERMATCH_SLACK (4u + OVERMATCH_LEN)  /* frontend_ring.rs:21 */
#define RING_WIDE 32u                         /* kit/wide.rs:2 */
#define Q1 0x40000000u

enum { COMMIT_NONE = 0, COMMIT_FSE = 1, COMMIT_VN = 2 };

struct lzo_ring {
    uint32_t ring_size, ring_blk, ring_limit;
    uint8_t *box;  /* ring/ring_type.rs:12: RING_SIZE + 2 * RING_LIMIT + WIDE, zeroed (ring_box.rs:9-17) */
    uint8_t *ring; /* box + RING_LIMIT (ring/object.rs:241-250) */
    history_t *table;
    int commit;
    match_t pending;
    uint32_t head, literal_idx, idx, tail, mark, clamp; /* wrapping Idx values (types/idx.rs) */
    uint64_t n_raw_bytes;
    frontend be; /* backend plumbing shared with the slice path: fse buffer / vn backend / dst / err / vn type */
    fse_buffer *fb;
    vec_t dst;
    uint8_t *lit_tmp;
    int dummy; /* KAT mode: encode/dummy.rs backend (trace only) */
};

/* types/idx.rs:75-80: ordering by wrapping signed difference */
static inline int idx_lt(uint32_t a, uint32_t b) { return (int32_t)(a - b) < 0; }
static inline int idx_ge(uint32_t a, uint32_t b) { return (int32_t)(a - b) >= 0; }

/* ring/object.rs:213-216 */
static inline uint32_t rf_get_u32(const lzo_ring *f, uint32_t idx) { return ld32(f->ring + (idx % f->ring_size)); }

/* ring/object.rs:147-155,181-189 via zone_copy_1 / zone_copy_2 (:252-261) */
static void rf_head_copy_out(lzo_ring *f) { memcpy(f->ring + f->ring_size, f->ring, f->ring_limit); }
static void rf_tail_copy_out(lzo_ring *f) {
    memcpy(f->ring - f->ring_limit, f->ring - f->ring_limit + f->ring_size, f->ring_limit);
}

/* frontend_ring.rs:577-583 */
static void rf_manage_ring_zones(lzo_ring *f) {
    if (f->mark % f->ring_size == f->ring_blk)
        rf_head_copy_out(f);
    else if (f->mark % f->ring_size == 0)
        rf_tail_copy_out(f);
}

/* ring/object.rs:39-84 with LEN = 4 */
static size_t rf_match_inc_coarse4(const lzo_ring *f, uint32_t a, uint32_t b, size_t max) {
    size_t i0 = a % f->ring_size, i1 = b % f->ring_size;
    uint64_t x = ld64(f->ring + i0 + 4) ^ ld64(f->ring + i1 + 4);
    if (x) return 4 + (size_t)(__builtin_ctzll(x) >> 3);
    size_t len = 4 + 8;
    for (;;) {
        for (size_t i = 0; i < 4; i++) {
            size_t off = 4 + 8 + i * 8;
            x = ld64(f->ring + i0 + off) ^ ld64(f->ring + i1 + off);
            if (x) return len + i * 8 + (size_t)(__builtin_ctzll(x) >> 3);
        }
        if (len >= max) break;
        len += 32;
        i0 = (i0 + 32) % f->ring_size;
        i1 = (i1 + 32) % f->ring_size;
    }
    return max;
}

/* ring/object.rs:88-135 with LEN = 0 */
static size_t rf_match_dec_coarse0(const lzo_ring *f, uint32_t a, uint32_t b, size_t max) {
    size_t i0 = (uint32_t)(a - OVERMATCH_LEN) % f->ring_size, i1 = (uint32_t)(b - OVERMATCH_LEN) % f->ring_size;
    uint64_t x = ld64(f->ring + i0 + 32) ^ ld64(f->ring + i1 + 32);
    if (x) return (size_t)(__builtin_clzll(x) >> 3);
    size_t len = 8;
    for (;;) {
        for (size_t i = 0; i < 4; i++) {
            size_t off = (3 - i) * 8;
            x = ld64(f->ring + i0 + off) ^ ld64(f->ring + i1 + off);
            if (x) return len + i * 8 + (size_t)(__builtin_clzll(x) >> 3);
        }
        if (len >= max) break;
        len += 32;
        i0 = (i0 + f->ring_size - 32) % f->ring_size;
        i1 = (i1 + f->ring_size - 32) % f->ring_size;
    }
    return max;
}

/* frontend_ring.rs:453-507; f_short = the const generic F */
static match_t rf_find_match(const lzo_ring *f, const history_t *queue, uint32_t val, uint32_t idx, uint32_t max,
                             int f_short) {
    match_t m = {0, 0, 0};
    for (int c = 0; c < HASH_WIDTH; c++) {
        uint32_t distance = idx - queue->q[c].idx;
        if (distance > f->be.max_dist) break;
        uint32_t x = val ^ queue->q[c].val, len;
        if (x == 0)
            len = (uint32_t)rf_match_inc_coarse4(f, idx, queue->q[c].idx, max); /* :493-501 */
        else if (f->be.vn && (x & 0x00FFFFFFu) == 0)
            len = 3;
        else
            len = 0;
        if (len > m.match_len) {
            m.match_len = len;
            m.match_idx = queue->q[c].idx;
        }
    }
    if (m.match_len == 0) return m;
    uint32_t literal_len = idx - f->literal_idx;
    m.idx = idx;
    if (f_short && m.match_len > max) m.match_len = max;
    uint32_t room = m.match_idx - f->head;
    uint32_t bmax = room < literal_len ? room : literal_len;
    uint32_t dec = (uint32_t)rf_match_dec_coarse0(f, m.idx, m.match_idx, bmax);
    if (dec > bmax) dec = bmax;
    m.idx -= dec;
    m.match_idx -= dec;
    m.match_len += dec;
    return m;
}

/* ring/ring_view.rs: the bytes [from, from + len) of the ring, wrap handled */
static const uint8_t *rf_view(lzo_ring *f, uint32_t from, uint32_t len) {
    size_t i = from % f->ring_size;
    if (i + len <= f->ring_size) return f->ring + i;
    size_t first = f->ring_size - i;
    memcpy(f->lit_tmp, f->ring + i, first);
    memcpy(f->lit_tmp + first, f->ring, len - first);
    return f->lit_tmp;
}

static void rf_backend_push(lzo_ring *f, uint32_t lit_from, uint32_t n_lit, uint32_t match_len, uint32_t dist) {
    if (f->be.trace && f->be.trace->match)
        f->be.trace->match(f->be.trace->ctx, lit_from, lit_from + n_lit, match_len, dist);
    if (f->dummy) return;
    const uint8_t *lit = rf_view(f, lit_from, n_lit);
    if (f->be.vn && match_len == 0) { /* vn/backend.rs:57-74 */
        if (!f->be.err && !vn_push_literals(&f->be.vnb, f->be.dst, lit, n_lit)) f->be.err = LZO_IO;
        return;
    }
    backend_push_match(&f->be, lit, n_lit, match_len, dist);
}

/* frontend_ring.rs:523-535 */
static void rf_push_match(lzo_ring *f, match_t m) {
    uint32_t from = f->literal_idx;
    f->literal_idx = m.idx + m.match_len;
    rf_backend_push(f, from, m.idx - from, m.match_len, m.idx - m.match_idx);
}

/* frontend_ring.rs:550-562: Backend::push_literals = push_match(literals, 0, 1) for Fse (fse/backend.rs:67-73) */
static void rf_push_literals(lzo_ring *f, uint32_t len) {
    uint32_t from = f->literal_idx;
    f->literal_idx += len;
    rf_backend_push(f, from, len, 0, 1);
}

/* frontend_ring.rs:359-397 */
static void rf_match_long(lzo_ring *f) {
    const uint32_t long_match_len = f->ring_size / 2 - f->ring_blk - OVERMATCH_SLACK; /* :110 */
    uint32_t idx = f->idx;
    f->idx = f->head + f->ring_size / 2 + f->ring_blk;
    for (;;) {
        uint32_t u = rf_get_u32(f, idx);
        history_t queue = table_push(f->table, f->be.vn, u, idx);
        match_t incoming = rf_find_match(f, &queue, u, idx, long_match_len, 0), sel;
        if (match_select(&f->pending, incoming, &sel)) {
            rf_push_match(f, sel);
            if (f->be.err) return;
            idx += 1;
            for (int32_t k = (int32_t)(f->literal_idx - idx); k > 0; k--) {
                table_push(f->table, f->be.vn, rf_get_u32(f, idx), idx);
                idx += 1;
            }
            if (idx_ge(idx, f->idx)) {
                f->idx = idx;
                break;
            }
        } else {
            idx += 1;
            if (idx == f->idx) break;
        }
    }
}

/* frontend_ring.rs:401-450 */
static void rf_match_short(lzo_ring *f) {
    uint32_t len = f->tail - f->idx;
    if (len < 4) return;
    const uint32_t unit = f->be.vn ? 3 : 4; /* MATCH_UNIT: fse/object.rs:24, vn/object.rs:24, encode/dummy.rs:27 */
    uint32_t idx = f->idx;
    f->idx = f->tail - unit + 1;
    for (;;) {
        uint32_t u = rf_get_u32(f, idx);
        history_t queue = table_push(f->table, f->be.vn, u, idx);
        uint32_t max = f->tail - idx;
        match_t incoming = rf_find_match(f, &queue, u, idx, max, 1), sel;
        if (match_select(&f->pending, incoming, &sel)) {
            rf_push_match(f, sel);
            if (f->be.err) return;
            if (idx_ge(f->literal_idx, f->idx)) {
                f->idx = f->literal_idx;
                break;
            }
            idx += 1;
            for (int32_t k = (int32_t)(f->literal_idx - idx); k > 0; k--) {
                table_push(f->table, f->be.vn, rf_get_u32(f, idx), idx);
                idx += 1;
            }
            if (idx_ge(idx, f->idx)) {
                f->idx = idx;
                break;
            }
        } else {
            idx += 1;
            if (idx == f->idx) break;
        }
    }
}

/* history.rs:45-50,121-130 with delta = 0 */
static void rf_table_clamp(history_t *t, uint32_t idx) {
    for (size_t i = 0; i < ((size_t)1 << HASH_BITS); i++)
        for (int k = 0; k < HASH_WIDTH; k++)
            if ((uint32_t)(idx - t[i].q[k].idx) > Q1) t[i].q[k].idx = idx - Q1;
}

/* history.rs:76-83 */
static void rf_table_reset_with_idx(history_t *t, uint32_t idx) {
    for (size_t i = 0; i < ((size_t)1 << HASH_BITS); i++)
        for (int k = 0; k < HASH_WIDTH; k++) {
            t[i].q[k].val = 0;
            t[i].q[k].idx = idx - Q1;
        }
}

/* frontend_ring.rs:209-227 (+ :250-272, :585-595) */
static void rf_match_block(lzo_ring *f) {
    rf_manage_ring_zones(f);
    if (f->mark != f->head + f->ring_size) {
        f->mark += f->ring_blk;
        return;
    }
    if (f->commit == COMMIT_NONE) { /* commit(Fse): fse/backend.rs:59-63 */
        f->commit = COMMIT_FSE;
        buffer_reset(f->fb);
    }
    rf_match_long(f);
    if (f->be.err) return;
    /* reposition_head :250-254 */
    uint32_t delta = f->idx - f->head;
    delta = (delta - f->ring_size / 2) / f->ring_blk * f->ring_blk;
    f->head += delta;
    /* push_literal_overflow :257-272 */
    if (idx_lt(f->literal_idx, f->head)) {
        f->pending.match_len = 0;
        rf_push_literals(f, f->head - f->literal_idx);
    }
    /* clamp :585-595 */
    if ((int32_t)(f->idx - f->clamp) >= 0) {
        rf_table_clamp(f->table, f->idx);
        f->clamp += Q1;
    }
    f->mark = f->tail + f->ring_blk;
}

/* frontend_ring.rs:564-575 */
static void rf_init(lzo_ring *f) {
    table_reset(f->table);
    f->commit = COMMIT_NONE;
    memset(&f->pending, 0, sizeof f->pending);
    f->head = f->literal_idx = f->idx = f->tail = 0;
    f->mark = f->ring_blk;
    f->clamp = Q1;
    f->n_raw_bytes = 0;
}

lzo_ring *lzo_ring_new(uint32_t ring_size, uint32_t ring_blk, uint32_t ring_limit, const lzo_trace *trace) {
    init_tables();
    if (!ring_size) { /* encode/constants.rs:23-33 (Input) */
        ring_size = 0x80000u;
        ring_blk = 0x4000u;
        ring_limit = 0x140u;
    }
    lzo_ring *f = (lzo_ring *)calloc(1, sizeof *f);
    if (!f) return NULL;
    f->ring_size = ring_size;
    f->ring_blk = ring_blk;
    f->ring_limit = ring_limit;
    f->box = (uint8_t *)calloc(1, (size_t)ring_size + 2 * (size_t)ring_limit + RING_WIDE + 64);
    f->table = (history_t *)malloc(sizeof(history_t) << HASH_BITS);
    f->fb = (fse_buffer *)malloc(sizeof(fse_buffer));
    f->lit_tmp = (uint8_t *)malloc(ring_size);
    if (!f->box || !f->table || !f->fb || !f->lit_tmp) {
        lzo_ring_free(f);
        return NULL;
    }
    f->ring = f->box + ring_limit;
    buffer_reset(f->fb);
    f->fb->trace = trace;
    f->fb->literals[0] = 0;
    f->be.vn = 0;
    f->be.max_dist = MAX_D_VALUE;
    f->be.fse = f->fb;
    f->be.dst = &f->dst;
    f->be.trace = trace;
    rf_init(f);
    return f;
}

void lzo_ring_free(lzo_ring *f) {
    if (!f) return;
    free(f->box);
    free(f->table);
    free(f->fb);
    free(f->lit_tmp);
    free(f->dst.p);
    free(f);
}

/* frontend_ring.rs:167-206 (`write`); `copy` (:138-164) feeds the same blocks from a reader */
int lzo_ring_write(lzo_ring *f, const uint8_t *src, size_t len) {
    f->n_raw_bytes += len;
    for (;;) {
        size_t index = f->tail % f->ring_size;
        size_t limit = (uint32_t)(f->mark - f->tail);
        if (len < limit) {
            memcpy(f->ring + index, src, len);
            f->tail += (uint32_t)len;
            return f->be.err;
        }
        memcpy(f->ring + index, src, limit);
        f->tail += (uint32_t)limit;
        src += limit;
        len -= limit;
        rf_match_block(f);
        if (f->be.err) return f->be.err;
    }
}

/* frontend_ring.rs:344-355 */
static void rf_finalize(lzo_ring *f) {
    rf_match_short(f);
    if (f->be.err) return;
    if (f->pending.match_len != 0) { /* flush_pending :510-520 */
        rf_push_match(f, f->pending);
        f->pending.match_len = 0;
    }
    uint32_t len = f->tail - f->literal_idx; /* flush_literals :538-548 */
    if (len != 0) rf_push_literals(f, len);
}

/* frontend_ring.rs:275-342 (flush, flush_select, flush_backend, flush_raw); the stream is then dst[0..*out_len) */
int lzo_ring_finish(lzo_ring *f, const uint8_t **out, size_t *out_len) {
    vec_t *dst = &f->dst;
    rf_manage_ring_zones(f);
    if (f->commit == COMMIT_FSE) {
        rf_finalize(f);
        if (!f->be.err) f->be.err = emit_block_v2(f->fb, dst);
    } else {
        uint32_t len = f->tail - f->idx;
        if (len > VN_CUTOFF) {
            f->commit = COMMIT_FSE;
            buffer_reset(f->fb);
            rf_finalize(f);
            if (!f->be.err) f->be.err = emit_block_v2(f->fb, dst);
        } else if (len > RAW_CUTOFF) {
            f->commit = COMMIT_VN;
            f->be.vn = 1;
            f->be.max_dist = VN_MAX_D;
            size_t mark = dst->len;
            memset(&f->be.vnb, 0, sizeof f->be.vnb); /* vn/backend.rs:42-55 */
            uint8_t zero[VN_HEADER_SIZE] = {0};
            if (!vec_put(dst, zero, VN_HEADER_SIZE)) f->be.err = LZO_IO;
            if (!f->be.err) rf_finalize(f);
            if (!f->be.err) { /* vn/backend.rs:127-135 */
                uint64_t eos = 0x06;
                if (!vec_put(dst, &eos, 8)) f->be.err = LZO_IO;
            }
            if (!f->be.err) {
                st32(dst->p + mark, MAGIC_VXN);
                st32(dst->p + mark + 4, f->be.vnb.n_literals + f->be.vnb.n_match_bytes);
                st32(dst->p + mark + 8, (uint32_t)(dst->len - mark) - VN_HEADER_SIZE);
                if (len < RAW_LIMIT && (size_t)len + 8 <= dst->len - mark) { /* :323-330 */
                    dst->len = mark;
                    if (!raw_compress(dst, rf_view(f, f->head, f->tail - f->head), f->tail - f->head)) f->be.err = LZO_IO;
                    f->literal_idx = f->tail;
                }
            }
        } else {
            if (!raw_compress(dst, rf_view(f, f->head, f->tail - f->head), f->tail - f->head)) f->be.err = LZO_IO;
            f->literal_idx = f->tail;
        }
    }
    if (!f->be.err && !vec_put32(dst, MAGIC_EOS)) f->be.err = LZO_IO;
    *out = dst->p;
    *out_len = dst->len;
    return f->be.err;
}

/* LzfseRingEncoder::encode over a whole buffer, fed in `piece`-byte writes (0 = all at once) */
int lzo_ring_encode(const uint8_t *src, size_t n, size_t piece, uint8_t *out, size_t cap, size_t *out_len,
                    const lzo_trace *trace) {
    lzo_ring *f = lzo_ring_new(0, 0, 0, trace);
    if (!f) return LZO_IO;
    int status = LZO_OK;
    if (!piece) piece = n ? n : 1;
    for (size_t o = 0; o < n && !status; o += piece) status = lzo_ring_write(f, src + o, n - o < piece ? n - o : piece);
    const uint8_t *p = NULL;
    size_t len = 0;
    if (!status) status = lzo_ring_finish(f, &p, &len);
    if (!status) {
        if (len > cap)
            status = LZO_BUFFER_OVERFLOW;
        else {
            memcpy(out, p, len);
            *out_len = len;
        }
    }
    lzo_ring_free(f);
    return status;
}

/* The set-ups of the reference's in-file KATs (frontend_ring.rs:861-992), which drive match_short / match_long on a
 * hand-made state with the Dummy backend (encode/dummy.rs: 3-byte match unit, distance <= 0x3FFF_FFFF; every push is
 * reported through trace->match as (literal_idx, match idx, match len, distance)). `ring_data` fills the ring from
 * index 0. mode 0: match_short from idx = 0 with tail = n (:869-883,:899-911,:971-982); mode 1: match_long from
 * idx = literal_idx = idx0 after table.reset_with_idx(idx0), head = 0, tail = mark = RING_SIZE (:929-941). Either
 * way a pending match is pushed afterwards and, in mode 0, the remaining literals. */
int lzo_ring_kat(int mode, uint32_t ring_size, uint32_t ring_blk, uint32_t ring_limit, const uint8_t *ring_data,
                 size_t n_data, uint32_t idx0, uint32_t n, const lzo_trace *trace) {
    lzo_ring *f = lzo_ring_new(ring_size, ring_blk, ring_limit, trace);
    if (!f) return LZO_IO;
    f->dummy = 1;
    f->be.vn = 1;
    f->be.max_dist = 0x3FFFFFFFu;
    if (n_data > ring_size) n_data = ring_size;
    memcpy(f->ring, ring_data, n_data);
    rf_head_copy_out(f);
    rf_tail_copy_out(f);
    if (mode == 0) {
        f->tail = n;
        f->mark = (n + ring_blk - 1) / ring_blk * ring_blk;
        rf_match_short(f);
    } else {
        rf_table_reset_with_idx(f->table, idx0);
        f->literal_idx = f->idx = idx0;
        f->tail = f->mark = ring_size;
        rf_match_long(f);
    }
    if (f->pending.match_len != 0) rf_push_match(f, f->pending);
    if (mode == 0 && f->tail != f->literal_idx) rf_push_literals(f, f->tail - f->literal_idx);
    lzo_ring_free(f);
    return LZO_OK;
}

/* ================================================================== decoder */

/* fse/decoder.rs:205-238 */
typedef struct {
    uint8_t k, v_bits;
    int16_t delta;
    uint32_t v_base;
} ventry;
typedef struct {
    uint8_t k, symbol;
    int16_t delta;
} uentry;

/* fse/decoder.rs:244-294 (offset folded out: each table is indexed from 0 here) */
static void build_v_table(const uint16_t *weights, int n_sym, const uint8_t *bits,
                          const uint32_t *base, ventry *table, uint32_t n_states) {
    int n_clz = clz32(n_states);
    uint32_t total = 0;
    for (int i = 0; i < n_sym; i++) {
        uint32_t w = weights[i];
        if (w == 0) continue;
        int k = clz32(w) - n_clz;
        uint32_t x = ((n_states << 1) >> k) - w;
        ventry e;
        e.v_bits = bits[i];
        e.v_base = base[i];
        e.k = (uint8_t)k;
        for (uint32_t j = 0; j < x; j++) {
            e.delta = (int16_t)((int32_t)((w + j) << k) - (int32_t)n_states);
            table[total + j] = e;
        }
        e.k = (uint8_t)(k - 1);
        for (uint32_t j = x; j < w; j++) {
            e.delta = (int16_t)((j - x) << (k - 1));
            table[total + j] = e;
        }
        total += w;
    }
    for (uint32_t i = total; i < n_states; i++) {
        ventry e = {0, 0, (int16_t)i, 0};
        table[i] = e;
    }
}

/* fse/decoder.rs:300-335 */
static void build_u_table(const uint16_t *weights, uentry *table) {
    const uint32_t n_states = U_STATES;
    int n_clz = clz32(n_states);
    uint32_t total = 0;
    for (int i = 0; i < U_SYMBOLS; i++) {
        uint32_t w = weights[i];
        if (w == 0) continue;
        int k = clz32(w) - n_clz;
        uint32_t x = ((n_states << 1) >> k) - w;
        uentry e;
        e.symbol = (uint8_t)i;
        e.k = (uint8_t)k;
        for (uint32_t j = 0; j < x; j++) {
            e.delta = (int16_t)((int32_t)((w + j) << k) - (int32_t)n_states);
            table[total + j] = e;
        }
        e.k = (uint8_t)(k - 1);
        for (uint32_t j = x; j < w; j++) {
            e.delta = (int16_t)((j - x) << (k - 1));
            table[total + j] = e;
        }
        total += w;
    }
    for (uint32_t i = total; i < n_states; i++) {
        uentry e = {0, 0, (int16_t)i};
        table[i] = e;
    }
}

/* bits/bit_reader.rs:11-72 over bits/bit_src.rs:33-53. `base` points at the 8 pad bytes
 * that precede the payload; len counts pad + payload. */
typedef struct {
    const uint8_t *base;
    int64_t idx;
    uint64_t accum;
    int accum_bits;
} bitr;

static inline uint64_t br_read(const bitr *r, int64_t idx) { return idx >= 0 ? ld64(r->base + idx) : 0; }

static int br_init(bitr *r, const uint8_t *base, size_t len, uint32_t off) {
    r->base = base;
    r->idx = (int64_t)len - 8;
    r->accum = br_read(r, r->idx);
    r->accum_bits = 64 - (int)off;
    if (off != 0 && (r->accum >> r->accum_bits) != 0) return LZO_BAD_BIT_STREAM;
    return LZO_OK;
}
static inline void br_flush(bitr *r) {
    int nbytes = (64 - r->accum_bits) / 8;
    r->idx -= nbytes;
    r->accum = br_read(r, r->idx);
    r->accum_bits += nbytes * 8;
}
static inline uint32_t br_pull(bitr *r, int n) {
    r->accum_bits -= n;
    uint64_t s = r->accum >> (r->accum_bits & 63);
    return (uint32_t)(s & ((1ull << n) - 1));
}
static int br_finalize(bitr *r) {
    br_flush(r);
    if ((int64_t)r->accum_bits + r->idx * 8 < 64) return LZO_PAYLOAD_UNDERFLOW;
    return LZO_OK;
}

typedef struct {
    uint32_t n_raw_bytes;
    uint32_t lit_num, lit_payload, lit_bits;
    uint16_t lit_state[4];
    uint32_t lmd_num, lmd_payload, lmd_bits;
    uint16_t lmd_state[3];
} fse_block;

/* fse/block.rs:218-226,267-283,324-341 (order: lmd, literal, raw count) */
static int fse_block_validate(const fse_block *b) {
    uint32_t lmd_limit = 1024 + 8 + (b->lmd_num * 14 + b->lmd_num * 17 + b->lmd_num * 23 + 7) / 8;
    if (b->lmd_num > LMDS_PER_BLOCK || b->lmd_payload < 8 ||
        (b->lmd_num <= LMDS_PER_BLOCK && b->lmd_payload > lmd_limit))
        return LZO_FSE_BAD_LMD_COUNT;
    if (b->lmd_bits > 7) return LZO_FSE_BAD_LMD_BITS;
    if (b->lmd_state[0] >= L_STATES || b->lmd_state[1] >= M_STATES || b->lmd_state[2] >= D_STATES)
        return LZO_FSE_BAD_LMD_STATE;
    if (b->lit_num % 4 != 0 || b->lit_num > LITERALS_PER_BLOCK) return LZO_FSE_BAD_LITERAL_COUNT;
    if (b->lit_payload > 1024 + (b->lit_num * 10 + 7) / 8) return LZO_FSE_BAD_LITERAL_COUNT;
    if (b->lit_bits > 7) return LZO_FSE_BAD_LITERAL_BITS;
    for (int i = 0; i < 4; i++)
        if (b->lit_state[i] >= U_STATES) return LZO_FSE_BAD_LMD_PAYLOAD;
    if (b->n_raw_bytes > b->lit_num + b->lmd_num * MAX_M_VALUE) return LZO_FSE_BAD_RAW_BYTE_COUNT;
    return LZO_OK;
}

/* fse/block.rs:108-136 */
static int fse_block_load_v2(fse_block *b, const uint8_t *h, uint32_t *n_weight_bytes) {
    b->n_raw_bytes = ld32(h + 4);
    uint64_t p = ld64(h + 8);
    b->lit_num = (uint32_t)(p & 0xFFFFF);
    b->lit_payload = (uint32_t)((p >> 20) & 0xFFFFF);
    b->lmd_num = (uint32_t)((p >> 40) & 0xFFFFF);
    b->lit_bits = 7 - (uint32_t)((p >> 60) & 7);
    p = ld64(h + 16);
    b->lit_state[0] = (uint16_t)(p & 0x3FF);
    b->lit_state[1] = (uint16_t)((p >> 10) & 0x3FF);
    b->lit_state[2] = (uint16_t)((p >> 20) & 0x3FF);
    b->lit_state[3] = (uint16_t)((p >> 30) & 0x3FF);
    b->lmd_payload = (uint32_t)((p >> 40) & 0xFFFFF);
    b->lmd_bits = 7 - (uint32_t)((p >> 60) & 7);
    p = ld64(h + 24);
    uint32_t header_size = (uint32_t)p;
    b->lmd_state[0] = (uint16_t)((p >> 32) & 0x3FF);
    b->lmd_state[1] = (uint16_t)((p >> 42) & 0x3FF);
    b->lmd_state[2] = (uint16_t)((p >> 52) & 0x3FF);
    uint32_t nw = header_size - V2_HEADER_SIZE; /* wrapping_sub */
    if (nw > V2_WEIGHT_PAYLOAD_BYTES_MAX) return LZO_FSE_BAD_WEIGHT_PAYLOAD;
    *n_weight_bytes = nw;
    return fse_block_validate(b);
}

/* fse/block.rs:80-104 */
static int fse_block_load_v1(fse_block *b, const uint8_t *h) {
    b->n_raw_bytes = ld32(h + 4);
    uint32_t n_payload_bytes = ld32(h + 8);
    b->lit_num = ld32(h + 12);
    b->lmd_num = ld32(h + 16);
    b->lit_payload = ld32(h + 20);
    b->lmd_payload = ld32(h + 24);
    b->lit_bits = 0u - ld32(h + 28);
    for (int i = 0; i < 4; i++) b->lit_state[i] = ld16(h + 32 + 2 * i);
    b->lmd_bits = 0u - ld32(h + 40);
    for (int i = 0; i < 3; i++) b->lmd_state[i] = ld16(h + 44 + 2 * i);
    if (n_payload_bytes < b->lit_payload + b->lmd_payload) return LZO_FSE_BAD_PAYLOAD_COUNT;
    return fse_block_validate(b);
}

typedef struct {
    uint8_t *dst;
    size_t cap, len;
} lzout;

/* lz/writer.rs:144-180 byte-serial semantics; D is bounded by the bytes produced in this
 * call (the reference bounds by the whole Vec, SURVEY.md 8b "back-references"). */
static int lz_write_match(lzout *o, uint32_t len, uint32_t distance) {
    if (distance == 0 || distance > o->len) return LZO_BAD_D_VALUE;
    if (o->len + len > o->cap) return LZO_BUFFER_OVERFLOW;
    uint8_t *d = o->dst + o->len;
    const uint8_t *s = d - distance;
    for (uint32_t t = 0; t < len; t++) d[t] = s[t];
    o->len += len;
    return LZO_OK;
}
static int lz_write_bytes(lzout *o, const uint8_t *src, size_t n) {
    if (o->len + n > o->cap) return LZO_BUFFER_OVERFLOW;
    memcpy(o->dst + o->len, src, n);
    o->len += n;
    return LZO_OK;
}

/* decoder.rs:102-141 + fse_core.rs:36-141 + literals.rs:49-91 */
static int decode_fse(const uint8_t *src, size_t n, size_t *pos, lzout *out, int v1,
                      const lzo_trace *trace) {
    init_tables();
    size_t avail = n - *pos;
    const uint8_t *p = src + *pos;
    fse_block blk;
    uint16_t weights[N_WEIGHTS];
    uint32_t hdr, nw;
    int e;
    if (v1) {
        if (avail < V1_HEADER_SIZE) return LZO_PAYLOAD_UNDERFLOW;
        if ((e = fse_block_load_v1(&blk, p))) return e;
        hdr = V1_HEADER_SIZE;
        nw = V1_WEIGHT_PAYLOAD_BYTES;
        if (avail - hdr < nw) return LZO_PAYLOAD_UNDERFLOW;
        for (int i = 0; i < N_WEIGHTS; i++) weights[i] = ld16(p + hdr + 2 * i); /* weights.rs:66-80 */
        if (!weights_check_totals(weights)) return LZO_FSE_BAD_WEIGHT_PAYLOAD;
    } else {
        if (avail < V2_HEADER_SIZE) return LZO_PAYLOAD_UNDERFLOW;
        if ((e = fse_block_load_v2(&blk, p, &nw))) return e;
        hdr = V2_HEADER_SIZE;
        if (avail - hdr < nw) return LZO_PAYLOAD_UNDERFLOW;
        if ((e = lzo_weights_load_v2(p + hdr, nw, weights))) return e;
    }
    static __thread ventry vl[L_STATES], vm[M_STATES], vd[D_STATES];
    static __thread uentry vu[U_STATES];
    static __thread uint8_t literals[LITERALS_PER_BLOCK + 64];
    build_v_table(weights, L_SYMBOLS, L_EXTRA_BITS, L_BASE_VALUE, vl, L_STATES);
    build_v_table(weights + 20, M_SYMBOLS, M_EXTRA_BITS, M_BASE_VALUE, vm, M_STATES);
    build_v_table(weights + 40, D_SYMBOLS, D_EXTRA_BITS, D_BASE_VALUE, vd, D_STATES);
    build_u_table(weights + 104, vu);
    size_t off = (size_t)hdr + nw - 8; /* fse_core.rs:46,59: 8 bytes lent as reader pad */
    /* literals: take(lit_payload + 8) fse_core.rs:62-69,201-203 */
    if (avail - off < (size_t)blk.lit_payload + 8) return LZO_PAYLOAD_UNDERFLOW;
    {
        bitr r;
        if ((e = br_init(&r, p + off, (size_t)blk.lit_payload + 8, blk.lit_bits))) return e;
        uint32_t s0 = blk.lit_state[0], s1 = blk.lit_state[1], s2 = blk.lit_state[2],
                 s3 = blk.lit_state[3];
        for (uint32_t i = 0; i != blk.lit_num; i += 4) {
            uentry u;
            u = vu[s0]; literals[i + 0] = u.symbol; s0 = (uint32_t)((int32_t)br_pull(&r, u.k) + u.delta);
            u = vu[s1]; literals[i + 1] = u.symbol; s1 = (uint32_t)((int32_t)br_pull(&r, u.k) + u.delta);
            u = vu[s2]; literals[i + 2] = u.symbol; s2 = (uint32_t)((int32_t)br_pull(&r, u.k) + u.delta);
            u = vu[s3]; literals[i + 3] = u.symbol; s3 = (uint32_t)((int32_t)br_pull(&r, u.k) + u.delta);
            br_flush(&r);
        }
        if ((e = br_finalize(&r))) return e;
        if (s0 | s1 | s2 | s3) return LZO_FSE_BAD_LMD_PAYLOAD;
    }
    off += (size_t)blk.lit_payload + 8;
    /* lmds: take(lmd_payload) fse_core.rs:80-141 */
    if (avail - off < blk.lmd_payload) return LZO_PAYLOAD_UNDERFLOW;
    {
        bitr r;
        if ((e = br_init(&r, p + off, blk.lmd_payload, blk.lmd_bits))) return e;
        uint32_t sl = blk.lmd_state[0], sm = blk.lmd_state[1], sd = blk.lmd_state[2];
        uint32_t literal_index = 0, n_match_bytes = 0, match_distance = 0;
        for (uint32_t k = blk.lmd_num; k != 0; k--) {
            ventry v;
            v = vl[sl]; sl = (uint32_t)((int32_t)br_pull(&r, v.k) + v.delta);
            uint32_t l = v.v_base + br_pull(&r, v.v_bits);
            v = vm[sm]; sm = (uint32_t)((int32_t)br_pull(&r, v.k) + v.delta);
            uint32_t m = v.v_base + br_pull(&r, v.v_bits);
            v = vd[sd]; sd = (uint32_t)((int32_t)br_pull(&r, v.k) + v.delta);
            uint32_t d = v.v_base + br_pull(&r, v.v_bits);
            br_flush(&r);
            if (d != 0) match_distance = d; /* lmd_type.rs:153-160 */
            if (trace && trace->lmd) trace->lmd(trace->ctx, l, m, match_distance);
            const uint8_t *lp = literals + literal_index;
            literal_index += l;
            if (literal_index > LITERALS_PER_BLOCK) return LZO_FSE_BAD_LMD_PAYLOAD;
            if ((e = lz_write_bytes(out, lp, l))) return e;
            if (m != 0) {
                n_match_bytes += m;
                if ((e = lz_write_match(out, m, match_distance))) return e;
            }
        }
        if ((e = br_finalize(&r))) return e;
        if (!(literal_index <= blk.lit_num && n_match_bytes + literal_index == blk.n_raw_bytes &&
              sl == 0 && sm == 0 && sd == 0))
            return LZO_FSE_BAD_LMD_PAYLOAD;
    }
    off += blk.lmd_payload;
    *pos += off;
    return LZO_OK;
}

/* vn/constants.rs:25-72 OP_TABLE expressed as ranges */
enum { OP_SML_L, OP_LRG_L, OP_SML_M, OP_LRG_M, OP_PRE_D, OP_SML_D, OP_MED_D, OP_LRG_D, OP_EOS, OP_UDEF, OP_NOP };

static int vn_op_of(uint32_t b) {
    uint32_t hi = b >> 4, lo = b & 15;
    if (hi == 0xE) return lo == 0 ? OP_LRG_L : OP_SML_L;
    if (hi == 0xF) return lo == 0 ? OP_LRG_M : OP_SML_M;
    if (hi == 0x7 || hi == 0xD) return OP_UDEF;
    if (hi == 0xA || hi == 0xB) return OP_MED_D;
    uint32_t low3 = b & 7;
    if (low3 == 7) return OP_LRG_D;
    if (low3 == 6) {
        if (b == 0x06) return OP_EOS;
        if (b == 0x0E || b == 0x16) return OP_NOP;
        if (b < 0x40) return OP_UDEF; /* 0x1E,0x26,0x2E,0x36,0x3E */
        return OP_PRE_D;
    }
    return OP_SML_D;
}

/* decoder.rs:144-157 + vn/vn_core.rs:41-287. `remaining stream` plays the role of the
 * (cycled) view: every op needs 8 spare bytes after what it consumes. */
static int decode_vn(const uint8_t *src, size_t n, size_t *pos, lzout *out) {
    size_t avail = n - *pos;
    const uint8_t *p = src + *pos;
    if (avail < VN_HEADER_SIZE) return LZO_PAYLOAD_UNDERFLOW;
    uint32_t n_raw_bytes = ld32(p + 4), n_payload_bytes = ld32(p + 8);
    size_t q = VN_HEADER_SIZE;
    uint32_t match_distance = 0;
    size_t out_mark = out->len;
    int e;
    if (avail - q < 8) return LZO_PAYLOAD_UNDERFLOW;
    for (;;) {
        size_t rem = avail - q; /* >= 8 here */
        const uint8_t *s = p + q;
        uint32_t opu = ld32(s);
        uint32_t l = 0, m = 0, op_len = 0;
        int has_d = 0;
        switch (vn_op_of(opu & 0xFF)) {
        case OP_SML_L: l = opu & 0xF; op_len = 1; goto literal_op;
        case OP_LRG_L: l = ((opu >> 8) & 0xFF) + 0x10; op_len = 2; goto literal_op;
        literal_op:
            if (rem - op_len < (size_t)l + 8) return LZO_PAYLOAD_UNDERFLOW;
            if ((e = lz_write_bytes(out, s + op_len, l))) return e;
            q += op_len + l;
            continue;
        case OP_SML_M: m = opu & 0xF; op_len = 1; goto match_op;
        case OP_LRG_M: m = ((opu >> 8) & 0xFF) + 0x10; op_len = 2; goto match_op;
        match_op:
            if (rem - op_len < 8) return LZO_PAYLOAD_UNDERFLOW;
            if ((e = lz_write_match(out, m, match_distance))) return e;
            q += op_len;
            continue;
        case OP_PRE_D:
            m = ((opu >> 3) & 7) + 3; l = (opu >> 6) & 3; op_len = 1;
            break;
        case OP_SML_D:
            m = ((opu >> 3) & 7) + 3; l = (opu >> 6) & 3; op_len = 2; has_d = 1;
            match_distance = ((opu & 7) << 8) | ((opu >> 8) & 0xFF);
            break;
        case OP_MED_D:
            m = (((opu & 7) << 2) | ((opu >> 8) & 3)) + 3; l = (opu >> 3) & 3; op_len = 3; has_d = 1;
            match_distance = (opu >> 10) & 0x3FFF;
            break;
        case OP_LRG_D:
            m = ((opu >> 3) & 7) + 3; l = (opu >> 6) & 3; op_len = 3; has_d = 1;
            match_distance = (opu >> 8) & 0xFFFF;
            break;
        case OP_NOP:
            if (rem - 1 < 8) return LZO_PAYLOAD_UNDERFLOW;
            q += 1;
            continue;
        case OP_EOS: {
            static const uint8_t eos[8] = {0x06, 0, 0, 0, 0, 0, 0, 0};
            if (memcmp(s, eos, 8) != 0) return LZO_VN_BAD_PAYLOAD;
            q += 8;
            size_t consumed = q - VN_HEADER_SIZE, produced = out->len - out_mark;
            if (consumed > n_payload_bytes) return LZO_PAYLOAD_UNDERFLOW;
            if (produced > n_raw_bytes) return LZO_VN_BAD_PAYLOAD;
            if (consumed != n_payload_bytes) return LZO_PAYLOAD_OVERFLOW;
            if (produced != n_raw_bytes) return LZO_VN_BAD_PAYLOAD;
            *pos += q;
            return LZO_OK;
        }
        default: return LZO_VN_BAD_OPCODE;
        }
        (void)has_d;
        /* pre_d / typ_d: vn_core.rs:222-283 (write_quad then write_match) */
        if (rem - op_len < (size_t)l + 8) return LZO_PAYLOAD_UNDERFLOW;
        if ((e = lz_write_bytes(out, s + op_len, l))) return e;
        if ((e = lz_write_match(out, m, match_distance))) return e;
        q += op_len + l;
    }
}

/* decoder.rs:160-173 + raw/block.rs:21-93 */
static int decode_raw(const uint8_t *src, size_t n, size_t *pos, lzout *out) {
    size_t avail = n - *pos;
    const uint8_t *p = src + *pos;
    if (avail < 8) return LZO_PAYLOAD_UNDERFLOW;
    uint32_t n_raw = ld32(p + 4);
    if (avail - 8 < n_raw) return LZO_PAYLOAD_UNDERFLOW;
    int e = lz_write_bytes(out, p + 8, n_raw);
    if (e) return e;
    *pos += 8 + (size_t)n_raw;
    return LZO_OK;
}

/* decoder.rs:61-99 */
int lzo_decode(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_len,
               const lzo_trace *trace) {
    lzout out = {dst, cap, 0};
    size_t pos = 0;
    int e;
    for (;;) {
        if (n - pos < 4) return LZO_PAYLOAD_UNDERFLOW;
        uint32_t magic = ld32(src + pos);
        if (magic == MAGIC_EOS) break;
        if (magic == MAGIC_VX2)
            e = decode_fse(src, n, &pos, &out, 0, trace);
        else if (magic == MAGIC_VX1)
            e = decode_fse(src, n, &pos, &out, 1, trace);
        else if (magic == MAGIC_VXN)
            e = decode_vn(src, n, &pos, &out);
        else if (magic == MAGIC_RAW)
            e = decode_raw(src, n, &pos, &out);
        else
            return LZO_BAD_BLOCK;
        if (e) return e;
    }
    if (n - pos != 4) return LZO_PAYLOAD_OVERFLOW;
    *out_len = out.len;
    return LZO_OK;
}

/* decode/probe.rs:11-35 (bvxn skip uses the payload length) */
int lzo_decode_size(const uint8_t *src, size_t n, uint64_t *raw_len) {
    size_t pos = 0;
    uint64_t total = 0;
    for (;;) {
        if (n - pos < 4) return LZO_PAYLOAD_UNDERFLOW;
        uint32_t magic = ld32(src + pos);
        size_t avail = n - pos, skip;
        uint32_t n_raw;
        if (magic == MAGIC_EOS) break;
        if (magic == MAGIC_VX2) {
            if (avail < V2_HEADER_SIZE) return LZO_PAYLOAD_UNDERFLOW;
            fse_block b;
            uint32_t nw;
            int e = fse_block_load_v2(&b, src + pos, &nw);
            if (e) return e;
            skip = (size_t)V2_HEADER_SIZE + nw + b.lit_payload + b.lmd_payload;
            n_raw = b.n_raw_bytes;
        } else if (magic == MAGIC_VX1) {
            if (avail < V1_HEADER_SIZE) return LZO_PAYLOAD_UNDERFLOW;
            fse_block b;
            int e = fse_block_load_v1(&b, src + pos);
            if (e) return e;
            skip = (size_t)V1_HEADER_SIZE + V1_WEIGHT_PAYLOAD_BYTES + b.lit_payload + b.lmd_payload;
            n_raw = b.n_raw_bytes;
        } else if (magic == MAGIC_VXN) {
            if (avail < VN_HEADER_SIZE) return LZO_PAYLOAD_UNDERFLOW;
            n_raw = ld32(src + pos + 4);
            skip = (size_t)VN_HEADER_SIZE + ld32(src + pos + 8);
        } else if (magic == MAGIC_RAW) {
            if (avail < 8) return LZO_PAYLOAD_UNDERFLOW;
            n_raw = ld32(src + pos + 4);
            skip = (size_t)8 + n_raw;
        } else {
            return LZO_BAD_BLOCK;
        }
        if (skip >= avail) return LZO_PAYLOAD_UNDERFLOW;
        pos += skip;
        total += n_raw;
    }
    if (n - pos != 4) return LZO_PAYLOAD_OVERFLOW;
    *raw_len = total;
    return LZO_OK;
}
