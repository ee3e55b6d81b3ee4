"""GPU parity tests (decode): HIP path through the C ABI vs the oracle and the reference's fixtures."""
import glob
import hashlib
import os

import numpy as np
import pytest

from oracle_py import rng_gen_vec, seq_masked

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import lzfse_rust_amd as m
    return m.Context(0)


def _fixture_files(golden):
    fs = []
    for sub in ("snappy", "special", "mutate"):
        fs += sorted(glob.glob(os.path.join(golden, sub, "*.lzfse")))
    return [f for f in fs if not f.endswith("null.vx2.lzfse")]


def test_decode_fixtures_batch(ctx, oracle, golden_dir):
    """test/src/data.rs:33-100: every fixture, SHA-256 of decoded bytes, all block types."""
    fs = _fixture_files(golden_dir)
    srcs = [open(f, "rb").read() for f in fs]
    outs, st = ctx.decode_batch(srcs)
    for f, s, o, e in zip(fs, srcs, outs, st):
        assert e == 0, (f, e)
        assert hashlib.sha256(o.tobytes()).digest() == open(f[:-6] + ".hash", "rb").read(), f
        assert o.tobytes() == oracle.decode(s), f


def gpu_last_lmds(dctx):
    """LMD records dec_fse_kernel left for the LZ stage in the context's last decode pass (diagnostic build's stage hook)."""
    import ctypes as C
    from lzfse_rust_amd import _native
    L = _native.lib(diag=True)
    f = L.lzfse_mi_debug_last_lmds
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    n = C.c_size_t(0)
    assert f(dctx._h, None, 0, C.byref(n)) in (0, 6)   # (6: BUFFER_OVERFLOW = "tell me the size")
    rec = np.zeros((n.value, 2), dtype=np.uint32)
    assert f(dctx._h, rec.ctypes.data, n.value, C.byref(n)) == 0
    return rec


def _parse_lmd_text(text):
    """src/ring/ring_lz_writer.rs:88-155: one 'L:n' line per LMD, followed by 'M:n' and 'D:n' when M != 0."""
    toks = text.split()
    kinds = np.array([ord(t[0]) for t in toks], dtype=np.uint8)
    vals = np.array([int(t[2:]) for t in toks], dtype=np.int64)
    out, i = [], 0
    while i < len(toks):
        assert kinds[i] == ord("L")
        if i + 2 < len(toks) and kinds[i + 1] == ord("M"):
            assert kinds[i + 2] == ord("D")
            out.append((vals[i], vals[i + 1], vals[i + 2]))
            i += 3
        else:
            out.append((vals[i], 0, -1))
            i += 1
    return np.array(out, dtype=np.int64).reshape(-1, 3)


def test_fse_stage_matches_reference_lmd_streams(diag_ctx, golden_dir):
    """dec_fse_kernel's LMD array against the reference's own decoded-LMD streams
    (data/snappy/lmdy_output/*.lmd, committed as tests/golden/lmd/*.lmd.gz): L always, M and D when M != 0, D = 0 already
    replaced by the previous distance -- the same comparison tests/test_oracle.py makes for the oracle, at the stage."""
    import gzip
    fs = sorted(glob.glob(os.path.join(golden_dir, "lmd", "*.lmd.gz")))
    assert len(fs) == 12
    for f in fs:
        name = os.path.basename(f)[:-7]
        exp = _parse_lmd_text(gzip.open(f, "rt").read())
        src = open(os.path.join(golden_dir, "snappy", name + ".lzfse"), "rb").read()
        outs, st = diag_ctx.decode_batch([src])
        assert st[0] == 0, name
        rec = gpu_last_lmds(diag_ctx).astype(np.int64)
        assert rec.shape[0] == exp.shape[0], (name, rec.shape, exp.shape)
        l, m, d = rec[:, 0] & 0xFFFF, rec[:, 0] >> 16, rec[:, 1]
        assert np.array_equal(l, exp[:, 0]), name
        assert np.array_equal(m, exp[:, 1]), name
        has_m = exp[:, 1] != 0
        assert np.array_equal(d[has_m], exp[has_m, 2]), name


def test_decode_single_api(ctx, golden_dir):
    import lzfse_rust_amd as m
    dec = m.LzfseDecoder(context=ctx)
    dst = bytearray(b"keep")
    n = dec.decode_bytes(open(os.path.join(golden_dir, "snappy", "alice29.txt.lzfse"), "rb").read(), dst)
    assert n == 152089 and dst[:4] == b"keep" and len(dst) == 4 + n


def test_decode_bytes_writes_the_vec_where_it_lies(ctx, oracle, snappy_raw):
    """The mirror decodes into the bytearray's own tail (codec._into_tail): what was there stays, what is appended is the
    stream's bytes, an error leaves the Vec as it was, and a Vec with a live view cannot grow (as for `+=`)."""
    import lzfse_rust_amd as m
    dec = m.LzfseDecoder(context=ctx)
    raw = snappy_raw["alice29.txt"] * 9      # (1.3 MB: a destination the library faults in while its kernels run)
    enc = oracle.encode(raw)
    dst = bytearray(b"head")
    for k in (1, 2):
        assert dec.decode_bytes(enc, dst) == len(raw)
        assert len(dst) == 4 + k * len(raw) and dst[:4] == b"head" and bytes(dst[4 + (k - 1) * len(raw):]) == raw
    bad = bytearray(enc)
    bad[0:4] = b"bvx9"
    before = bytes(dst)
    with pytest.raises(m.LzfseError):
        dec.decode_bytes(bytes(bad), dst)
    assert bytes(dst) == before
    view = memoryview(dst)
    with pytest.raises(BufferError):
        dec.decode_bytes(enc, dst)
    view.release()
    assert bytes(dst) == before
    out = bytearray()
    assert dec.decode_bytes(oracle.encode(b""), out) == 0 and out == bytearray()
    big = bytearray(b"x")
    assert m.LzfseEncoder(context=ctx).encode_bytes(raw, big) == len(enc) and bytes(big[1:]) == enc


def test_decode_oracle_streams(ctx, oracle, snappy_raw):
    """Streams produced by the oracle encoder (10 000-LMD blocks, unlike Apple's 9 992)."""
    raws = list(snappy_raw.values())
    raws.append(seq_masked(3, 0x01010101, 1 << 20))
    raws.append(rng_gen_vec(7, 300000))
    raws.append(bytes(500000))
    raws.append(b"abc" * 100000)
    raws.append(bytes(range(256)) * 2000)
    raws += [bytes(n) for n in (0, 1, 20, 21, 4096, 4097)]
    raws += [rng_gen_vec(n, n) for n in (5, 100, 4096, 4097, 40001)]
    encs = [oracle.encode(r) for r in raws]
    outs, st = ctx.decode_batch(encs)
    for r, o, e in zip(raws, outs, st):
        assert e == 0
        assert o.tobytes() == r


def test_decode_many_streams_both_lz_variants(diag_ctx, oracle, snappy_raw):
    ctx = diag_ctx
    encs = [oracle.encode(r) for r in snappy_raw.values()] * 50
    raws = list(snappy_raw.values()) * 50
    for variant in (0, 1):
        ctx.set_option("diag_lz_tile", variant)
        try:
            outs, st = ctx.decode_batch(encs)
        finally:
            ctx.set_option("diag_lz_tile", -1)
        assert all(e == 0 for e in st)
        for r, o in zip(raws, outs):
            assert o.tobytes() == r


def test_decode_errors_match_oracle(ctx, oracle, golden_dir, snappy_raw):
    """Every malformed input must give Err on both sides (never crash / hang): mutate_0.rs."""
    base = bytearray(open(os.path.join(golden_dir, "mutate", "vx2.lzfse"), "rb").read())
    cases = []
    rng = np.random.default_rng(11)
    for i in rng.choice(len(base), size=300, replace=False):
        m = bytearray(base)
        m[i] ^= 1 << int(rng.integers(0, 8))
        cases.append(bytes(m))
    enc = oracle.encode(snappy_raw["html"])
    cases += [enc[:-1], enc[:-4], enc[:-5], enc + b"\0", b"abcd" + enc, enc[:100], b"", b"bvx"]
    e2 = bytearray(enc)
    n = int.from_bytes(e2[4:8], "little")
    e2[4:8] = (n + 1).to_bytes(4, "little")
    cases.append(bytes(e2))
    outs, st = ctx.decode_batch(cases, caps=[1 << 20] * len(cases))
    for c, o, e in zip(cases, outs, st):
        es = oracle.decode_status(c, 1 << 20)
        assert e == es, (e, es)
        if e == 0:
            assert o.tobytes() == oracle.decode(c, cap=1 << 20)
    assert st[-1] == 22  # BadLmdPayload for n_raw_bytes + 1 (fse/test.rs:434,458)


def test_decode_bytes_has_vec_semantics(ctx, oracle, snappy_raw):
    """decode_bytes appends to a Vec (decode/decoder.rs:52-57): a block that produces MORE than its n_raw_bytes fails with
    BadLmdPayload at its end (fse/fse_core.rs:132-140), not with the fixed-capacity status of the C entry points."""
    import lzfse_rust_amd as m
    enc = bytearray(oracle.encode(snappy_raw["html"]))
    n = int.from_bytes(enc[4:8], "little")
    enc[4:8] = (n - 1).to_bytes(4, "little")
    assert oracle.decode_status(bytes(enc), 1 << 26) == 22
    outs, st = ctx.decode_batch([bytes(enc)])
    assert st[0] == 6
    with pytest.raises(m.LzfseError) as e:
        m.LzfseDecoder(context=ctx).decode_bytes(bytes(enc), bytearray())
    assert e.value.status == 22


def test_decode_capacity_too_small(ctx, oracle, snappy_raw):
    enc = oracle.encode(snappy_raw["html"])
    outs, st = ctx.decode_batch([enc], caps=[1000])
    assert st[0] == 6


def test_parallel_header_walk_matches_serial(diag_ctx, oracle, golden_dir, snappy_raw):
    """Large streams have their block headers found by a magic scan + chain ranking instead of the serial walk; anything
    that is not a clean run of bvx2 blocks from position 0 to the end-of-stream magic falls back to the serial walk. The
    diagnostic build sends EVERY stream through the parallel walk first (1) or through the serial walk only (2): fixtures
    of all block kinds, multi-block streams, payloads that contain the bvx2 magic, damaged and cut streams -- same bytes,
    same status codes, both equal to the oracle's."""
    fs = _fixture_files(golden_dir)
    srcs = [open(f, "rb").read() for f in fs]
    for k in ("raw", "vx1", "vx2", "vxn"):
        srcs.append(open(os.path.join(golden_dir, "mutate", k + ".lzfse"), "rb").read())
    srcs.append(open(os.path.join(golden_dir, "special", "compound.lzfse"), "rb").read())
    rng = np.random.default_rng(17)
    magic = b"bvx2" + bytes(rng.integers(0, 256, size=40, dtype=np.uint8))
    raws = [snappy_raw["lcet10.txt"] * 3, bytes(rng.integers(0, 256, size=600000, dtype=np.uint8)),
            (magic * 3000) + snappy_raw["html"], b"".join(bytes(rng.integers(0, 256, size=50, dtype=np.uint8)) + b"bvx2bvx$bvxnbvx-" for _ in range(20000))]
    encs = [oracle.encode(r) for r in raws]
    srcs += encs
    big = encs[0]
    for _ in range(60):      # damaged copies of a multi-block stream
        m = bytearray(big)
        if rng.random() < 0.3:
            m = m[: int(rng.integers(4, len(m)))]
        else:
            m[int(rng.integers(0, len(m)))] ^= 1 << int(rng.integers(0, 8))
        srcs.append(bytes(m))
    srcs.append(big + b"\0")        # bytes behind the end-of-stream magic
    srcs.append(big[:-4])            # no end-of-stream magic
    caps = [max(oracle_cap(oracle, s_), 64) for s_ in srcs]
    want = [(oracle.decode_status(s_, c_), s_, c_) for s_, c_ in zip(srcs, caps)]
    got = {}
    for mode in (1, 2):
        diag_ctx.set_option("diag_walk", mode)
        try:
            got[mode] = diag_ctx.decode_batch(srcs, caps=caps)
        finally:
            diag_ctx.set_option("diag_walk", 0)
    for i, (ws, s_, c_) in enumerate(want):
        for mode in (1, 2):
            outs, st = got[mode]
            assert st[i] == ws, (i, mode, st[i], ws)
            if ws == 0:
                assert outs[i].tobytes() == oracle.decode(s_, cap=c_), (i, mode)


def oracle_cap(oracle, stream):
    """Capacity for a possibly damaged stream: what its headers promise, bounded."""
    try:
        return min(int(oracle.decode_size(stream)) + 64, 8 << 20)
    except Exception:
        return 4 << 20


def test_decode_pointer_jumping_path(diag_ctx, oracle, golden_dir, snappy_raw):
    """The LZ stage for large streams (origin pointer jumping) forced on for every stream."""
    fs = _fixture_files(golden_dir)
    srcs = [open(f, "rb").read() for f in fs]
    raws = [bytes(700000), b"abc" * 300000, seq_masked(4, 0x01010101, 3 << 20), bytes(range(256)) * 9000,
            snappy_raw["html_x_4"] * 6]
    encs = [oracle.encode(r) for r in raws]
    ctx = diag_ctx
    ctx.set_option("diag_lz_path", 1)
    try:
        outs, st = ctx.decode_batch(srcs + encs)
        # malformed input through the same path
        base = bytearray(open(os.path.join(golden_dir, "mutate", "vx2.lzfse"), "rb").read())
        cases = []
        rng = np.random.default_rng(5)
        for i in rng.choice(len(base), size=200, replace=False):
            m = bytearray(base)
            m[i] ^= 1 << int(rng.integers(0, 8))
            cases.append(bytes(m))
        outs2, st2 = ctx.decode_batch(cases, caps=[1 << 20] * len(cases))
    finally:
        ctx.set_option("diag_lz_path", -1)
    for f, s, o, e in zip(fs, srcs, outs, st):
        assert e == 0, f
        assert hashlib.sha256(o.tobytes()).digest() == open(f[:-6] + ".hash", "rb").read(), f
    for r, o, e in zip(raws, outs[len(fs):], st[len(fs):]):
        assert e == 0 and o.tobytes() == r
    for c, o, e in zip(cases, outs2, st2):
        es = oracle.decode_status(c, 1 << 20)
        assert e == es, (e, es)
        if e == 0:
            assert o.tobytes() == oracle.decode(c, cap=1 << 20)


def test_decode_many_tiny_blocks_rewalk(ctx, oracle, snappy_raw):
    """Streams with far more blocks than their share of the header-walk cache (one per 2 KiB of input) take the
    serial re-walk when the block descriptors are placed; mixed with ordinary streams in one batch."""
    tiny = b"".join(b"bvx-" + (1).to_bytes(4, "little") + bytes([i & 255]) for i in range(3000)) + b"bvx$"
    tiny2 = b"".join(b"bvx-" + (3).to_bytes(4, "little") + bytes([i & 255, 7, 9]) for i in range(500)) + b"bvx$"
    ordinary = oracle.encode(snappy_raw["html"])
    srcs = [ordinary, tiny, ordinary, tiny2]
    outs, st = ctx.decode_batch(srcs)
    assert all(e == 0 for e in st), st
    for s, o in zip(srcs, outs):
        assert o.tobytes() == oracle.decode(s)
    assert len(outs[1]) == 3000 and len(outs[3]) == 1500


def test_entropy_stage_damaged_and_cut_blocks(diag_ctx, oracle, golden_dir, snappy_raw):
    """The entropy stage on what its fast path must hand to the careful one: every fixture, multi-block streams whose blocks
    differ in length, blocks cut short, damaged payloads (the run-out rule of the bit reader, literals past the block limit,
    bad final states) -- the oracle's bytes and status codes."""
    import glob
    import os
    rng = np.random.default_rng(77)
    srcs = []
    for sub in ("snappy", "special", "mutate"):
        srcs += [open(f, "rb").read() for f in sorted(glob.glob(os.path.join(golden_dir, sub, "*.lzfse")))]
    big = oracle.encode(snappy_raw["lcet10.txt"] + snappy_raw["html"] * 2 + snappy_raw["kppkn.gtb"][:70000])
    srcs += [big, oracle.encode(snappy_raw["urls.10K"]), oracle.encode(bytes(300000)), oracle.encode(snappy_raw["fireworks.jpeg"])]
    for base in (open(os.path.join(golden_dir, "mutate", "vx2.lzfse"), "rb").read(), open(os.path.join(golden_dir, "mutate", "vx1.lzfse"), "rb").read()):
        for bit in range(0, 8 * len(base), 7):
            m = bytearray(base)
            m[bit >> 3] ^= 1 << (bit & 7)
            srcs.append(bytes(m))
    for _ in range(150):
        m = bytearray(big)
        if rng.random() < 0.3:
            m = m[: int(rng.integers(40, len(m)))]
        else:
            for _k in range(int(rng.integers(1, 3))):
                m[int(rng.integers(0, len(m)))] ^= 1 << int(rng.integers(0, 8))
        srcs.append(bytes(m))
    cap = 1 << 21
    want = []
    for s in srcs:
        st = oracle.decode_status(s, cap)
        want.append((st, oracle.decode(s, cap=cap) if st == 0 else None))
    outs, st = diag_ctx.decode_batch(srcs, caps=[cap] * len(srcs))
    for i, (o, e) in enumerate(zip(outs, st)):
        assert e == want[i][0], (i, e, want[i][0])
        if e == 0:
            assert o.tobytes() == want[i][1], i
