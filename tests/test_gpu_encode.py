"""GPU parity tests (encode): every stage and the final bytes vs the oracle, bit-exact."""
import ctypes as C
import hashlib

import numpy as np
import pytest

from oracle_py import rng_gen_vec, seq_masked

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import lzfse_rust_amd as m
    return m.Context(0)


def gpu_candidates(ctx, raw):
    """(chain links, candidate records) of one stream from the diagnostic build's stage hook."""
    from lzfse_rust_amd import _native
    L = _native.lib(diag=True)
    f = L.lzfse_mi_debug_candidates
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    a = np.frombuffer(raw, dtype=np.uint8)
    link = np.zeros(a.size - 3, dtype=np.uint32)  # link records: distance (18 bits) | check bits
    rec = np.zeros((a.size - 3, 2), dtype=np.uint32)
    st = f(ctx._h, a.ctypes.data, a.size, link.ctypes.data, rec.ctypes.data)
    assert st == 0
    dist = (link & 0x3FFFF).astype(np.int64)
    prev = np.where(dist != 0, np.arange(a.size - 3, dtype=np.int64) - dist, -1)
    return prev, rec


def synth_cases():
    rng = np.random.default_rng(3)
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(2000)]
    text = b" ".join(words[int(i)] for i in rng.integers(0, 2000, size=120000))
    return {
        "text": text,
        "zeros_4097": bytes(4097),
        "zeros_300k": bytes(300000),
        "abc": b"abc" * 70000,
        "period_70k": (rng.integers(0, 256, size=70000, dtype=np.uint8).tobytes()) * 4,
        "noise_mask": seq_masked(1, 0x01010101, 1 << 19),
        "noise": rng_gen_vec(9, 200000),
        "ramp": bytes(range(256)) * 1500,
        "tail3": text[:4100],
        # heavily skewed symbol statistics: one FSE symbol takes most of a table, so states lose almost no bits per
        # step (the speculative state chains of enc_block_kernel must fall back to their exact fix-up path)
        "skew2": rng.choice(np.array([0x41, 0x42], dtype=np.uint8), p=[0.93, 0.07], size=300000).tobytes(),
        "skew_geo": np.minimum(rng.geometric(0.4, size=400000) - 1, 255).astype(np.uint8).tobytes(),
        "skew_sparse": (rng.integers(1, 256, size=500000, dtype=np.uint8) * (rng.random(500000) < 0.05)).astype(np.uint8).tobytes(),
    }


def test_candidate_stage_matches_oracle(diag_ctx, oracle, snappy_raw):
    """enc_chain / enc_link / enc_cand vs the oracle: the bucket chain of every position equals the newest entry of the
    history row it sees (history.rs push), and the as-if-visited find_match at every position. The 3 MiB case spans
    48 candidate tiles (links across chain tiles of every length)."""
    from oracle_py import seq_masked
    cases = dict(synth_cases())
    for k in ("html", "alice29.txt", "kppkn.gtb", "urls.10K"):
        cases[k] = snappy_raw[k]
    cases["tiles"] = (snappy_raw["lcet10.txt"] + seq_masked(3, 0x03030303, 200000)) * 5
    for name, raw in cases.items():
        mi, fl = oracle.candidates(raw)
        prev, rec = gpu_candidates(diag_ctx, raw)
        if name in ("tiles", "text", "noise_mask"):
            # the same links whatever the length of a chain tile (1, 2 or 4 candidate tiles; the call picks one)
            for force in (0x10, 0x20, 0x40):
                diag_ctx.set_option("diag_chain", force)
                try:
                    prev_f, rec_f = gpu_candidates(diag_ctx, raw)
                finally:
                    diag_ctx.set_option("diag_chain", 0)
                assert (prev_f == prev).all() and (rec_f == rec).all(), (name, force)
        want = oracle.table_rows(raw)
        n = len(raw) - 3
        # prev[i] = newest entry of the row, as long as it lies within the match window (links across chain tiles stop there)
        near = (want[:, 0] != 0xFFFFFFFF) & (np.arange(n, dtype=np.int64) - want[:, 0].astype(np.int64) <= 262139)
        assert (prev[near] == want[near, 0]).all(), name
        assert (prev[~near] == -1).all(), name   # nothing, or nothing within the window
        idx = np.arange(n, dtype=np.int64)
        fwd = rec[:, 1]
        dist = rec[:, 0] & 0x3FFFF
        capped = (rec[:, 0] >> 31) != 0
        exact = ~capped
        assert (fwd[exact] == fl[exact]).all(), name
        has = exact & (fl > 0)
        assert ((idx[has] - dist[has]) == mi[has]).all(), name
        assert (fwd[capped] == 1023).all() and (fl[capped] >= 1023).all(), name   # FCAP: "at least 1023"


def test_encode_bit_exact_snappy(ctx, oracle, snappy_raw):
    names = list(snappy_raw)
    outs, st = ctx.encode_batch([snappy_raw[n] for n in names])
    for n, o, e in zip(names, outs, st):
        assert e == 0, (n, e)
        exp = oracle.encode(snappy_raw[n])
        assert len(o) == len(exp), (n, len(o), len(exp))
        assert o.tobytes() == exp, n


def test_encode_bit_exact_synthetic(ctx, oracle):
    cases = synth_cases()
    names = list(cases)
    outs, st = ctx.encode_batch([cases[n] for n in names])
    for n, o, e in zip(names, outs, st):
        assert e == 0, (n, e)
        assert o.tobytes() == oracle.encode(cases[n]), n


def test_chain_ballot_kernel_matches(diag_ctx, oracle, snappy_raw):
    """The chain links are made by one LDS exchange per position, which relies on the measured (not architectural)
    lane order of that instruction and checks itself; a tile that fails the check is redone by the ballot kernel. The
    diagnostic build sends every tile there: same streams, same stage results."""
    from oracle_py import seq_masked
    cases = dict(synth_cases())
    cases["tiles"] = (snappy_raw["lcet10.txt"] + seq_masked(3, 0x03030303, 200000)) * 5
    for k in ("html", "urls.10K", "kppkn.gtb"):
        cases[k] = snappy_raw[k]
    names = list(cases)
    diag_ctx.set_option("diag_chain", 1)
    try:
        outs, st = diag_ctx.encode_batch([cases[n] for n in names])
        slow = {n: gpu_candidates(diag_ctx, cases[n]) for n in ("tiles", "text", "abc")}
    finally:
        diag_ctx.set_option("diag_chain", 0)
    for n, o, e in zip(names, outs, st):
        assert e == 0, (n, e)
        assert o.tobytes() == oracle.encode(cases[n]), n
    for n, (prev_s, rec_s) in slow.items():
        prev_f, rec_f = gpu_candidates(diag_ctx, cases[n])
        assert (prev_s == prev_f).all() and (rec_s == rec_f).all(), n


def test_encode_zero_4097_kat(ctx):
    """The reference's only bvx2 byte-exact KAT (frontend_bytes.rs:513-531) through the GPU path."""
    from test_oracle import ZERO_4097
    outs, st = ctx.encode_batch([bytes(4097)])
    assert st[0] == 0 and outs[0].tobytes() == ZERO_4097


def test_encode_many_sizes(ctx, oracle):
    rng = np.random.default_rng(8)
    raws = []
    base = synth_cases()["text"]
    for n in list(range(4097, 4130)) + [8191, 8192, 8193, 39999, 40000, 40001, 65535, 65536, 65537, 131073]:
        raws.append(base[:n])
    for n in (5000, 70000):
        raws.append(rng.integers(0, 4, size=n, dtype=np.uint8).tobytes())
    outs, st = ctx.encode_batch(raws)
    for r, o, e in zip(raws, outs, st):
        assert e == 0
        assert o.tobytes() == oracle.encode(r), len(r)


def test_encode_decode_roundtrip_gpu(ctx, snappy_raw):
    raws = list(snappy_raw.values()) * 8
    encs, st = ctx.encode_batch(raws)
    assert all(e == 0 for e in st)
    outs, st = ctx.decode_batch([e.tobytes() for e in encs])
    assert all(e == 0 for e in st)
    for r, o in zip(raws, outs):
        assert o.tobytes() == r


def test_encode_api_appends(ctx, oracle, snappy_raw):
    import lzfse_rust_amd as m
    enc = m.LzfseEncoder(context=ctx)
    dst = bytearray(b"xy")
    n = enc.encode_bytes(snappy_raw["html"], dst)
    assert bytes(dst[2:]) == oracle.encode(snappy_raw["html"]) and n == len(dst) - 2


def test_encode_all_size_classes_through_api(ctx, oracle):
    """raw / LZVN / bvx2 size classes in one batch (frontend_bytes.rs:63-77), KATs of :455-531 included."""
    from test_oracle import ZERO_4097
    base = synth_cases()["text"]
    raws = [bytes(n) for n in (0, 1, 20, 21, 4096, 4097)] + [base[:n] for n in (5, 100, 3000, 4096, 4097, 9000)]
    outs, st = ctx.encode_batch(raws)
    for r, o, e in zip(raws, outs, st):
        assert e == 0
        assert o.tobytes() == oracle.encode(r), len(r)
    assert outs[5].tobytes() == ZERO_4097
    dec, st2 = ctx.decode_batch([o.tobytes() for o in outs])
    assert all(e == 0 for e in st2)
    for r, o in zip(raws, dec):
        assert o.tobytes() == r


def test_encode_random_structures_bit_exact(ctx, oracle):
    """Property-style sweep: 160 streams of random length and random structure (noise, low-entropy noise, repeats
    with mutations, periodic data, word soup, long runs) in one batch, every output equal to the oracle's."""
    rng = np.random.default_rng(1234)
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(1, 12)), dtype=np.uint8)) for _ in range(500)]

    def gen(kind, n):
        if kind == 0:
            return rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        if kind == 1:
            return (rng.integers(0, 4, size=n, dtype=np.uint8) * 37).astype(np.uint8).tobytes()
        if kind == 2:
            chunk = rng.integers(0, 256, size=int(rng.integers(5, 5000)), dtype=np.uint8)
            a = np.tile(chunk, n // chunk.size + 1)[:n].copy()
            idx = rng.integers(0, n, size=max(1, n // int(rng.integers(50, 5000))))
            a[idx] ^= rng.integers(1, 256, size=idx.size, dtype=np.uint8)
            return a.tobytes()
        if kind == 3:
            per = int(rng.integers(1, 70000))
            return (bytes(rng.integers(0, 256, size=per, dtype=np.uint8)) * (n // per + 1))[:n]
        if kind == 4:
            out = bytearray()
            while len(out) < n:
                out += words[int(rng.integers(0, 500))] + b" "
            return bytes(out[:n])
        runs = bytearray()
        while len(runs) < n:
            runs += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 3000))
        return bytes(runs[:n])

    raws = [gen(int(rng.integers(0, 6)), int(rng.integers(4097, 300000))) for _ in range(160)]
    outs, st = ctx.encode_batch(raws)
    assert all(e == 0 for e in st)
    for r, o in zip(raws, outs):
        assert o.tobytes() == oracle.encode(r), len(r)
    dec, st2 = ctx.decode_batch([o.tobytes() for o in outs])
    assert all(e == 0 for e in st2)
    for r, o in zip(raws, dec):
        assert o.tobytes() == r


def test_encode_few_streams_many_waves_per_stream(ctx, oracle, snappy_raw):
    """Calls of at most 8 streams run the stitcher with 8 waves per stream (enc_stitch_kernel<8>: 512 boundaries a step, the
    true walk by wave 0 while the others wait) and, up to 32 MiB, quarter segments: data whose segment logs do not meet
    (runs, periods, deserts of noise between text) in calls of 1, 3 and 8 streams, both parses, and one call beyond 32 MiB
    (2 048-position segments, still 8 waves)."""
    rng = np.random.default_rng(99)
    text = snappy_raw["lcet10.txt"] + snappy_raw["plrabn12.txt"]
    noise = lambda n: rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
    kinds = [bytes(3 << 20),                                            # one run
             (noise(1100) * 3000)[:3 << 20],                            # period 1 100
             text[:700000] + noise(400000) + text[200000:900000] + bytes(300000) + text[:500000],
             b"".join(bytes([int(v)]) * int(c) for v, c in zip(rng.integers(0, 256, 1500), rng.integers(1, 6000, 1500)))[:2500000],
             text * 2, noise(1 << 20), (text[:65000] + noise(300)) * 40, text[:4097 + 64 * 1024]]
    for ns in (1, 3, 8):
        raws = kinds[:ns] if ns > 1 else [kinds[2]]
        for ring in (False, True):
            outs, st = ctx.encode_batch(raws, ring=ring)
            assert all(e == 0 for e in st)
            for r, o in zip(raws, outs):
                assert o.tobytes() == (oracle.ring_encode(r) if ring else oracle.encode(r)), (ns, ring, len(r))
    big = [text * 24, kinds[2] * 3]        # 39 MB in two streams: the large-call segments under the many-wave stitcher
    outs, st = ctx.encode_batch(big)
    assert all(e == 0 for e in st)
    for r, o in zip(big, outs):
        assert o.tobytes() == oracle.encode(r), len(r)


def test_encode_tile_and_batch_edges_bit_exact(ctx, diag_ctx, oracle, snappy_raw):
    """Stream lengths around the edges of the match-finding kernels: the 65 472-position candidate tile and the chain tiles of
    1, 2 and 4 of them (last tile of 1, 2, 63 positions; exactly full), the 1 024-position batch of the chain kernel (tail batch
    of 1 / 1 023 positions; a tile of fewer batches than the workgroup has waves), the 256-position candidate workgroup and the
    64-position wave; text, a highly repetitive input and low-entropy noise, through the LDS-exchange chain kernel and through
    its ballot fallback, with the call's own choice of chain tile and with each length forced (LZFSE_MI_OPT_DIAG_CHAIN)."""
    from oracle_py import seq_masked
    tile = 65472
    text = (snappy_raw["alice29.txt"] + snappy_raw["lcet10.txt"]) * 2
    rep = (snappy_raw["html"][:7000] * 170)
    noise = seq_masked(5, 0x07070707, 1190000)
    lens = [tile + 3, tile + 4, tile + 5, tile + 3 + 63, tile + 3 + 64, tile + 3 + 65, 2 * tile + 2, 2 * tile + 3, 2 * tile + 4,
            tile + 3 + 2047, tile + 3 + 2048, tile + 3 + 2049, tile + 3 + 255, tile + 3 + 256, tile + 3 + 257, 3 * tile + 3 + 1,
            tile + 3 + 1023, tile + 3 + 1024, tile + 3 + 1025, 4 * tile + 2, 4 * tile + 3, 4 * tile + 4, 4 * tile + 3 + 64,
            8 * tile + 3, 8 * tile + 4, 12 * tile + 3 + 7 * 1024 + 5, 4100, 4097 + 1024, 8 * 1024 + 3, 8 * 1024 + 4]
    raws = [src[:n] for src in (text, rep, noise) for n in lens if n <= len(src)]
    want = [oracle.encode(r) for r in raws]
    for c, force in ((ctx, None), (diag_ctx, 1), (diag_ctx, 0x10), (diag_ctx, 0x20), (diag_ctx, 0x40), (diag_ctx, 0x41), (diag_ctx, 0x21)):
        if force is not None:
            c.set_option("diag_chain", force)
        try:
            outs, st = c.encode_batch(raws)
        finally:
            if force is not None:
                c.set_option("diag_chain", 0)
        assert all(e == 0 for e in st)
        for r, o, w in zip(raws, outs, want):
            assert o.tobytes() == w, (len(r), force)


def test_encode_block_boundaries_bit_exact(ctx, oracle):
    """Multi-block streams whose blocks close on the literal limit (noise), on the LMD limit (dense short matches),
    inside over-long literal runs (L > 315) and inside over-long matches (M > 2 359): fse/buffer.rs:45-97."""
    rng = np.random.default_rng(77)
    noise = rng.integers(0, 256, size=1_300_000, dtype=np.uint8).tobytes()
    dense = np.tile(rng.integers(0, 256, size=8, dtype=np.uint8), 200_000)
    dense[::9] ^= rng.integers(1, 256, size=dense[::9].size, dtype=np.uint8)  # a mismatch every 9 bytes: ~10^5 tiny matches
    long_runs = b"".join(bytes([int(rng.integers(0, 256))]) * int(rng.integers(2000, 9000)) for _ in range(300))
    mixed = noise[:50_000] + long_runs[:400_000] + dense.tobytes()[:300_000] + noise[50_000:120_000] + bytes(250_000)
    lit_runs = b"".join(rng.integers(0, 256, size=int(rng.integers(300, 700)), dtype=np.uint8).tobytes() + b"0123456789abcdef" * 3
                        for _ in range(2500))
    raws = [noise, dense.tobytes(), long_runs, mixed, lit_runs]
    outs, st = ctx.encode_batch(raws)
    assert all(e == 0 for e in st)
    for r, o in zip(raws, outs):
        assert o.tobytes() == oracle.encode(r), len(r)


def test_encode_blocks_cut_all_at_once_bit_exact(ctx, oracle, snappy_raw):
    """Large streams have all their bvx2 blocks cut at once when only the 10 000-LMD limit closes blocks
    (enc_segpar_kernel), and by the serial cut otherwise: text with long zero runs (matches of many LMDs that straddle block
    ends), with runs of noise of some hundred bytes (literal runs split into several LMDs), and with runs of noise long
    enough to close blocks by the 40 000-literal limit (the stream is left to the serial kernel) -- the oracle's bytes."""
    rng = np.random.default_rng(41)
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(3000)]

    def text(n):
        out = bytearray()
        while len(out) < n:
            out += words[int(rng.integers(0, len(words)))] + b" "
        return bytes(out[:n])

    def spliced(n, zero_runs, noise_runs, noise_len):
        a = bytearray(text(n))
        for _ in range(zero_runs):
            p, m = int(rng.integers(0, n - 40000)), int(rng.integers(2400, 30000))
            a[p:p + m] = bytes(m)
        for _ in range(noise_runs):
            p, m = int(rng.integers(0, n - 70000)), int(rng.integers(noise_len[0], noise_len[1]))
            a[p:p + m] = rng.integers(0, 256, size=m, dtype=np.uint8).tobytes()
        return bytes(a)

    raws = [spliced(6 << 20, 150, 150, (320, 3000)),          # LMD-bound throughout
            spliced(6 << 20, 40, 12, (30000, 65000)),           # some blocks closed by the literal limit
            text(5 << 20) + bytes(3 << 20) + text(1 << 20),     # one match of 1 300 LMDs across a block end
            (snappy_raw["lcet10.txt"] * 16)[: 6 << 20]]
    encs, st = ctx.encode_batch(raws)
    assert all(e == 0 for e in st)
    for r, e in zip(raws, encs):
        assert hashlib.sha256(e.tobytes()).digest() == hashlib.sha256(oracle.encode(r)).digest(), len(r)
