/* Test infrastructure (oracle side), NOT product code: a CPU study asked for before any speculative sub-block decoding is built
 * (SURVEY section 7, H1 "self-synchronisation"). For the bvx2 blocks of the reference's Snappy fixtures: start the LMD decoder
 * (three interleaved tANS states L, M, D over ONE shared bit cursor, fse_core.rs:91-141) and the literal decoder (four states,
 * literals.rs:49-91) somewhere inside a block with something WRONG -- the states, or the cursor by a few bits -- and count the
 * steps until (cursor, states) fall onto the true decoder's sequence. A decoder that never does within 4 096 steps counts as "never".
 *
 *   make -C oracle fse_sync_study.bin && oracle/fse_sync_study.bin tests/golden/snappy/ *.lzfse
 */
#include "lzfse_oracle.c"

#include <stdio.h>

static inline uint32_t bits_at(const uint8_t *base, int64_t len, int64_t P, int n) { /* bits [P - n, P), 0 where the payload ends */
    if (n == 0) return 0;
    int64_t lo = P - n;
    if (lo < 0) return 0;
    int64_t byte = lo >> 3;
    uint64_t v = 0;
    for (int k = 0; k < 8 && byte + k < len; k++) v |= (uint64_t)base[byte + k] << (8 * k);
    return (uint32_t)((v >> (lo & 7)) & ((1ull << n) - 1));
}

typedef struct { int64_t P; uint16_t s[4]; } rec_t;

enum { MAX_STEPS = 4096, NB = 8 };
static const int EDGE[NB] = {4, 16, 64, 256, 1024, 4096, 1 << 30, 0};
static const char *EDGE_NAME[NB] = {"<=4", "<=16", "<=64", "<=256", "<=1024", "<=4096", "never", ""};
static uint64_t hist[6][NB];   /* trial kinds: 0 lmd wrong states (zeros) 1 lmd wrong states (true ^ 1 in L only) 2 lmd cursor + 1 bit
                                  3 lit wrong states (zeros) 4 lit one state wrong 5 lit cursor + 1 bit */
static void tally(int kind, int steps) {
    for (int b = 0; b < NB - 1; b++) if (steps <= EDGE[b]) { hist[kind][b]++; return; }
}

/* one LMD step from (P, sl, sm, sd) */
static inline void lmd_step(const ventry *vl, const ventry *vm, const ventry *vd, const uint8_t *base, int64_t len, int64_t *P, uint16_t *s) {
    ventry v;
    v = vl[s[0] & (L_STATES - 1)]; s[0] = (uint16_t)((int32_t)bits_at(base, len, *P, v.k) + v.delta); *P -= v.k; *P -= v.v_bits;
    v = vm[s[1] & (M_STATES - 1)]; s[1] = (uint16_t)((int32_t)bits_at(base, len, *P, v.k) + v.delta); *P -= v.k; *P -= v.v_bits;
    v = vd[s[2] & (D_STATES - 1)]; s[2] = (uint16_t)((int32_t)bits_at(base, len, *P, v.k) + v.delta); *P -= v.k; *P -= v.v_bits;
}
static inline void lit_step(const uentry *vu, const uint8_t *base, int64_t len, int64_t *P, uint16_t *s) {
    for (int q = 0; q < 4; q++) { uentry u = vu[s[q] & (U_STATES - 1)]; s[q] = (uint16_t)((int32_t)bits_at(base, len, *P, u.k) + u.delta); *P -= u.k; }
}

/* steps until a decoder started at (P, s) meets the true sequence `t` (n records, P strictly decreasing... non-increasing) */
static int meet(const rec_t *t, uint32_t n, uint32_t from, int64_t P, uint16_t *s, int ns, int is_lit, const void *ta, const void *tb, const void *tc,
                const uint8_t *base, int64_t len) {
    uint32_t j = from;   /* t[j].P >= P is kept */
    for (int step = 0; step <= MAX_STEPS; step++) {
        while (j + 1 < n && t[j + 1].P >= P) j++;
        /* the true decoder passes cursor P with these states? (records are taken at step boundaries only) */
        for (uint32_t q = j; q < n && t[q].P == P; q++) {
            int same = 1;
            for (int k = 0; k < ns; k++) same &= t[q].s[k] == s[k];
            if (same) return step;
            if (q == n - 1) break;
        }
        for (uint32_t q = j; q > from && t[q].P == P; q--) {
            int same = 1;
            for (int k = 0; k < ns; k++) same &= t[q].s[k] == s[k];
            if (same) return step;
        }
        if (P <= 0) break;
        if (is_lit) lit_step((const uentry *)ta, base, len, &P, s);
        else lmd_step((const ventry *)ta, (const ventry *)tb, (const ventry *)tc, base, len, &P, s);
    }
    return 1 << 30;
}

int main(int argc, char **argv) {
    init_tables();
    uint64_t n_blocks = 0, n_trials = 0;
    for (int a = 1; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) continue;
        fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
        uint8_t *src = malloc((size_t)n + 64); memset(src + n, 0, 64);
        if (fread(src, 1, (size_t)n, f) != (size_t)n) return 1;
        fclose(f);
        size_t pos = 0;
        while (pos + 4 <= (size_t)n && ld32(src + pos) == MAGIC_VX2) {
            const uint8_t *p = src + pos;
            fse_block blk; uint16_t weights[N_WEIGHTS]; uint32_t nw;
            if (fse_block_load_v2(&blk, p, &nw) || lzo_weights_load_v2(p + V2_HEADER_SIZE, nw, weights)) break;
            static ventry vl[L_STATES], vm[M_STATES], vd[D_STATES];
            static uentry vu[U_STATES];
            build_v_table(weights, L_SYMBOLS, L_EXTRA_BITS, L_BASE_VALUE, vl, L_STATES);
            build_v_table(weights + 20, M_SYMBOLS, M_EXTRA_BITS, M_BASE_VALUE, vm, M_STATES);
            build_v_table(weights + 40, D_SYMBOLS, D_EXTRA_BITS, D_BASE_VALUE, vd, D_STATES);
            build_u_table(weights + 104, vu);
            size_t off = (size_t)V2_HEADER_SIZE + nw - 8;
            /* literal stream: payload of lit_payload + 8 bytes, cursor starts at len * 8 - lit_bits */
            const uint8_t *lb = p + off; int64_t llen = (int64_t)blk.lit_payload + 8;
            uint32_t ng = blk.lit_num / 4;
            rec_t *lt = malloc(sizeof(rec_t) * (ng + 1));
            { int64_t P = llen * 8 - blk.lit_bits; uint16_t s[4] = {blk.lit_state[0], blk.lit_state[1], blk.lit_state[2], blk.lit_state[3]};
              for (uint32_t g = 0; g <= ng; g++) { lt[g].P = P; memcpy(lt[g].s, s, 8); if (g < ng) lit_step(vu, lb, llen, &P, s); } }
            off += (size_t)blk.lit_payload + 8;
            const uint8_t *mb = p + off; int64_t mlen = blk.lmd_payload;
            uint32_t nl = blk.lmd_num;
            rec_t *mt = malloc(sizeof(rec_t) * (nl + 1));
            { int64_t P = mlen * 8 - blk.lmd_bits; uint16_t s[4] = {blk.lmd_state[0], blk.lmd_state[1], blk.lmd_state[2], 0};
              for (uint32_t g = 0; g <= nl; g++) { mt[g].P = P; memcpy(mt[g].s, s, 8); if (g < nl) lmd_step(vl, vm, vd, mb, mlen, &P, s); } }
            for (uint32_t g = 256; g + 64 < nl; g += 256) {
                uint16_t s[4];
                memset(s, 0, 8);                                    tally(0, meet(mt, nl + 1, g, mt[g].P, s, 3, 0, vl, vm, vd, mb, mlen));
                memcpy(s, mt[g].s, 8); s[0] ^= 1;                   tally(1, meet(mt, nl + 1, g, mt[g].P, s, 3, 0, vl, vm, vd, mb, mlen));
                memcpy(s, mt[g].s, 8);                              tally(2, meet(mt, nl + 1, g, mt[g].P + 1, s, 3, 0, vl, vm, vd, mb, mlen));
                n_trials++;
            }
            for (uint32_t g = 256; g + 64 < ng; g += 256) {
                uint16_t s[4];
                memset(s, 0, 8);                                    tally(3, meet(lt, ng + 1, g, lt[g].P, s, 4, 1, vu, 0, 0, lb, llen));
                memcpy(s, lt[g].s, 8); s[0] ^= 1;                   tally(4, meet(lt, ng + 1, g, lt[g].P, s, 4, 1, vu, 0, 0, lb, llen));
                memcpy(s, lt[g].s, 8);                              tally(5, meet(lt, ng + 1, g, lt[g].P + 1, s, 4, 1, vu, 0, 0, lb, llen));
            }
            free(lt); free(mt);
            off += blk.lmd_payload;
            pos += off;
            n_blocks++;
        }
        free(src);
    }
    static const char *KIND[6] = {"LMD stream, true cursor, all three states wrong (0, 0, 0)", "LMD stream, true cursor, only the L state wrong (bit 0 flipped)",
                                  "LMD stream, true states, cursor one bit too far", "literal stream, true cursor, all four states wrong (0)",
                                  "literal stream, true cursor, one state wrong (bit 0 flipped)", "literal stream, true states, cursor one bit too far"};
    printf("# oracle/fse_sync_study.c over %llu bvx2 blocks of the Snappy fixtures, one start point every 256 steps: steps until a decoder that\n"
           "# starts with the named defect is on the true decoder's (cursor, states) sequence again\n", (unsigned long long)n_blocks);
    for (int k = 0; k < 6; k++) {
        uint64_t tot = 0;
        for (int b = 0; b < NB - 1; b++) tot += hist[k][b];
        printf("%-66s n=%-6llu", KIND[k], (unsigned long long)tot);
        for (int b = 0; b < NB - 1; b++) printf("  %s: %5.1f %%", EDGE_NAME[b], tot ? 100.0 * (double)hist[k][b] / (double)tot : 0.0);
        printf("\n");
    }
    (void)n_trials;
    return 0;
}
