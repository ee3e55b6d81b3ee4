/* A C caller of include/lzfse_mi.h (plain gcc, no C++ / Python / torch anywhere): proves the header is C and that the
 * boundary works as the reference's lzfse_sys-style binding would use it (lzfse_sys/src/lib.rs:29-56 is the template:
 * caller-owned buffers, sizes in, size out). Exit codes: 0 = round trip ok, 3 = no HIP device (LZFSE_MI_NO_DEVICE from
 * lzfse_mi_create: the product never falls back to a CPU codec), anything else = failure.
 *
 *   gcc -std=c11 -Wall -Wextra -Werror -I include tests/abi_driver.c -L lzfse_rust_amd -llzfse_mi -o abi_driver
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lzfse_mi.h"

static unsigned lcg(unsigned *s) { *s = *s * 1103515245u + 12345u; return *s >> 8; }

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

    lzfse_mi_destroy(ctx);
    free(src); free(enc); free(dec);
    printf("abi driver ok: %zu -> %zu bytes and back\n", n, enc_len);
    return 0;
}
