/* A C caller of include/lzfse_mi.h (plain gcc, no C++ / Python / torch anywhere): proves the header is C and that the
 * boundary works as the reference's lzfse_sys-style binding would use it (lzfse_sys/src/lib.rs:29-56 is the template:
 * caller-owned buffers, sizes in, size out), the streaming entry points included. Exit codes: 0 = round trip ok, 3 = no HIP device (LZFSE_MI_NO_DEVICE from
 * lzfse_mi_create: the product never falls back to a CPU codec), anything else = failure.
 *
 *   gcc -std=c11 -Wall -Wextra -Werror -I include tests/abi_driver.c -L lzfse_rust_amd -llzfse_mi -o abi_driver
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lzfse_mi.h"

static unsigned lcg(unsigned *s) { *s = *s * 1103515245u + 12345u; return *s >> 8; }

/* what a binding's Write / Vec<u8> sink is to the streaming entry points */
struct sink { uint8_t *p; size_t len, cap; };
static int sink_write(void *user, const uint8_t *bytes, size_t n) {
    struct sink *s = user;
    if (s->len + n > s->cap) return 1;
    memcpy(s->p + s->len, bytes, n);
    s->len += n;
    return 0;
}

int main(void) {
    printf("%s\n", lzfse_mi_version());
    if (strcmp(lzfse_mi_status_string(LZFSE_MI_FSE_BAD_LMD_PAYLOAD), "FSE: bad LMD payload") != 0) return 10;

    /* host-only entry points work without a device */
    const uint8_t tiny[4] = {'t', 'e', 's', 't'};
    uint8_t small[64];
    size_t n_small = 0;
    if (lzfse_mi_encode_small(tiny, 4, small, sizeof small, &n_small) != LZFSE_MI_OK || n_small != 16) return 11;
    uint64_t raw_len = 0;
    if (lzfse_mi_decode_size(small, n_small, &raw_len) != LZFSE_MI_OK || raw_len != 4) return 12;
    if (lzfse_mi_encode_bound(100000) < 100000) return 13;

    lzfse_mi_ctx *ctx = NULL;
    int st = lzfse_mi_create(0, &ctx);
    if (st == LZFSE_MI_NO_DEVICE) { printf("no HIP device: create refused (no CPU fallback)\n"); return 3; }
    if (st != LZFSE_MI_OK) return 20;

    /* a compressible 300 000-byte input: words from a small vocabulary */
    const size_t n = 300000;
    uint8_t *src = malloc(n), *enc, *dec;
    unsigned seed = 7;
    for (size_t i = 0; i < n;) {
        unsigned w = lcg(&seed) % 97, len = 2 + w % 7;
        for (unsigned k = 0; k < len && i < n; k++) src[i++] = (uint8_t)('a' + (w * 7 + k * 3) % 26);
        if (i < n) src[i++] = ' ';
    }
    size_t cap = lzfse_mi_encode_bound(n), enc_len = 0, dec_len = 0;
    enc = malloc(cap);
    dec = malloc(n);
    if (lzfse_mi_encode(ctx, src, n, enc, cap, &enc_len) != LZFSE_MI_OK) return 21;
    if (enc_len >= n || memcmp(enc, "bvx2", 4) != 0 || memcmp(enc + enc_len - 4, "bvx$", 4) != 0) return 22;
    if (lzfse_mi_decode_size(enc, enc_len, &raw_len) != LZFSE_MI_OK || raw_len != n) return 23;
    if (lzfse_mi_decode(ctx, enc, enc_len, dec, n, &dec_len) != LZFSE_MI_OK || dec_len != n) return 24;
    if (memcmp(src, dec, n) != 0) return 25;

    /* a damaged stream: status code and its payload */
    enc[0] = 'q';
    uint32_t detail = 0;
    st = lzfse_mi_decode(ctx, enc, enc_len, dec, n, &dec_len);
    if (st != LZFSE_MI_BAD_BLOCK) return 26;
    if (lzfse_mi_last_error_detail(ctx, 0, &detail) != LZFSE_MI_OK || detail != 0x32787671u /* "qvx2" */) return 27;

    /* batch entry point, two streams, one too small a destination */
    const uint8_t *srcs[2] = {src, src + 1000};
    size_t lens[2] = {50000, 60000}, caps[2] = {cap, 16}, outs[2] = {0, 0};
    uint8_t *dsts[2] = {enc, dec};
    int sts[2] = {-1, -1};
    if (lzfse_mi_encode_batch(ctx, 2, srcs, lens, dsts, caps, outs, sts) != LZFSE_MI_OK) return 28;
    if (sts[0] != LZFSE_MI_OK || sts[1] != LZFSE_MI_BUFFER_OVERFLOW || outs[0] == 0 || outs[1] != 0) return 29;
    if (lzfse_mi_set_option(ctx, LZFSE_MI_OPT_ENCODE_LANES, 1) != LZFSE_MI_OK) return 30;
    if (lzfse_mi_set_option(ctx, LZFSE_MI_OPT_DIAG_LZ_PATH, 1) != LZFSE_MI_UNSUPPORTED) return 31;

    /* the streaming surface from C: 3 MiB through the stream encoder in 1 MiB windows and odd pieces -- the bytes of the
     * ring encoder's one-call form -- and back through the stream decoder */
    {
        const size_t big = 3u << 20;
        uint8_t *raw = malloc(big), *one = malloc(lzfse_mi_encode_bound(big));
        for (size_t i = 0; i < big; i++) raw[i] = src[(i * 7 + (i >> 12)) % n];
        size_t one_len = 0;
        if (lzfse_mi_encode_ring(ctx, raw, big, one, lzfse_mi_encode_bound(big), &one_len) != LZFSE_MI_OK) return 40;
        struct sink es = {malloc(lzfse_mi_encode_bound(big)), 0, lzfse_mi_encode_bound(big)}, ds = {malloc(big), 0, big};
        lzfse_mi_estream *e = NULL;
        if (lzfse_mi_estream_create(ctx, 1u << 20, &e) != LZFSE_MI_OK) return 41;
        for (size_t o = 0; o < big;) {
            size_t k = 100000 + o % 7777; if (k > big - o) k = big - o;
            if (lzfse_mi_estream_feed(e, raw + o, k, sink_write, &es) != LZFSE_MI_OK) return 42;
            o += k;
        }
        const size_t early = es.len;
        uint64_t u = 0, v = 0;
        if (lzfse_mi_estream_finish(e, sink_write, &es, &u, &v) != LZFSE_MI_OK || u != big || v != es.len) return 43;
        lzfse_mi_estream_destroy(e);
        if (es.len != one_len || memcmp(es.p, one, one_len) != 0 || early == 0) return 44;
        lzfse_mi_dstream *d = NULL;
        if (lzfse_mi_dstream_create(ctx, 1u << 20, &d) != LZFSE_MI_OK) return 45;
        for (size_t o = 0; o < es.len;) {
            size_t k = 65536; if (k > es.len - o) k = es.len - o;
            if (lzfse_mi_dstream_feed(d, es.p + o, k, o + k == es.len, sink_write, &ds) != LZFSE_MI_OK) return 46;
            o += k;
        }
        lzfse_mi_dstream_destroy(d);
        if (ds.len != big || memcmp(ds.p, raw, big) != 0) return 47;
        free(raw); free(one); free(es.p); free(ds.p);
    }

    lzfse_mi_destroy(ctx);
    free(src); free(enc); free(dec);
    printf("abi driver ok: %zu -> %zu bytes and back\n", n, enc_len);
    return 0;
}
