"""GPU error-path parity: damaged streams of every block type through the device decoders; the status CODE (not just
ok / err) must equal the oracle's, which follows the reference's order of checks (decoder.rs:76-173,
fse_core.rs:49-141, vn_core.rs:41-287). Mirrors test/src/mutate_*.rs (all four data/mutate fixtures) and adds
special/compound (bvx- + bvx1 + bvx2 + bvxn in one stream)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import lzfse_rust_amd as m
    return m.Context(0)


def _mutations(base, rng, n_flips, n_bytes, n_trunc):
    out = []
    for i in rng.choice(len(base), size=min(n_flips, len(base)), replace=False):
        m = bytearray(base)
        m[i] ^= 1 << int(rng.integers(0, 8))
        out.append(bytes(m))
    for i in rng.choice(len(base), size=min(n_bytes, len(base)), replace=False):
        m = bytearray(base)
        m[i] = int(rng.integers(0, 256))
        out.append(bytes(m))
    for i in rng.choice(len(base), size=min(n_trunc, len(base)), replace=False):
        out.append(bytes(base[:i]))
    return out


def _check_cases(ctx, oracle, cases, cap):
    outs, st = ctx.decode_batch(cases, caps=[cap] * len(cases))
    bad = []
    for k, (c, o, e) in enumerate(zip(cases, outs, st)):
        es = oracle.decode_status(c, cap)
        if e != es:
            bad.append((k, e, es))
        elif e == 0:
            assert o.tobytes() == oracle.decode(c, cap=cap)
    assert not bad, f"{len(bad)} of {len(cases)} status codes differ (case, gpu, oracle): {bad[:10]}"


@pytest.mark.parametrize("name", ["mutate/raw", "mutate/vx1", "mutate/vx2", "mutate/vxn", "special/compound"])
def test_damaged_fixture_status_codes_equal_oracle(ctx, oracle, golden_dir, name):
    base = open(os.path.join(golden_dir, name + ".lzfse"), "rb").read()
    rng = np.random.default_rng(len(base))
    cases = _mutations(base, rng, 400, 150, 60)
    # the header region decides most error kinds: every bit of the first 64 bytes as well (mutate_1.rs-style sweep)
    for i in range(min(64, len(base))):
        for b in range(8):
            m = bytearray(base)
            m[i] ^= 1 << b
            cases.append(bytes(m))
    _check_cases(ctx, oracle, cases, 1 << 20)


def test_damaged_multi_block_stream_status_codes(ctx, oracle, snappy_raw):
    """Several bvx2 blocks per stream: an error inside an early block must win over a header error in a later one
    (the reference decodes in order); bad distances must come before the block's end-of-stream checks."""
    raw = snappy_raw["urls.10K"]
    enc = oracle.encode(raw)
    rng = np.random.default_rng(77)
    cases = _mutations(enc, rng, 250, 100, 40)
    # two independent damages: one early, one late
    for _ in range(60):
        m = bytearray(enc)
        a, b = sorted(int(x) for x in rng.integers(0, len(enc), size=2))
        m[a] ^= 1 << int(rng.integers(0, 8))
        m[b] ^= 1 << int(rng.integers(0, 8))
        cases.append(bytes(m))
    _check_cases(ctx, oracle, cases, len(raw) + 4096)


def test_damaged_streams_pointer_jumping_path(diag_ctx, oracle, snappy_raw, golden_dir):
    raw = snappy_raw["urls.10K"]
    enc = oracle.encode(raw)
    base = open(os.path.join(golden_dir, "mutate", "vx2.lzfse"), "rb").read()
    rng = np.random.default_rng(5)
    cases = _mutations(enc, rng, 120, 40, 20) + _mutations(base, rng, 120, 40, 20)
    diag_ctx.set_option("diag_lz_path", 1)
    try:
        _check_cases(diag_ctx, oracle, cases, len(raw) + 4096)
    finally:
        diag_ctx.set_option("diag_lz_path", -1)


def test_jump_path_fresh_scratch_odd_sizes(oracle):
    """origin entries that no kernel writes (padding between streams of odd size, failed blocks) must be
    defined. A fresh context (nothing recycled) decodes streams of (3 << 20) + 1 and 2 MiB + 3 bytes on the default
    (cost-chosen) jump path, then a >= 2 MiB stream with one damaged middle block."""
    import lzfse_rust_amd as m
    from oracle_py import seq_masked
    c = m.Context(0)
    raws = [seq_masked(9, 0x03030303, (3 << 20) + 1), seq_masked(10, 0x01010101, (2 << 20) + 3)]
    encs = [oracle.encode(r) for r in raws]
    outs, st = c.decode_batch(encs)
    assert list(st) == [0, 0]
    for r, o in zip(raws, outs):
        assert o.tobytes() == r
    bad = bytearray(encs[0])
    bad[len(bad) // 2] ^= 0x10
    cap = len(raws[0]) + 4096
    outs, st = c.decode_batch([bytes(bad), encs[1]], caps=[cap, cap])
    assert st[0] == oracle.decode_status(bytes(bad), cap)
    assert st[1] == 0 and outs[1].tobytes() == raws[1]
    c.close()


def test_error_detail_payloads(ctx, oracle, snappy_raw):
    """Error::BadBlock(magic), FseErrorKind::BadLmdCount(n), BadLiteralCount(n) carry a u32 (error/mod.rs:47,
    fse/error_kind.rs:12-21): lzfse_mi_last_error_detail returns it per stream of the last call."""
    enc = bytearray(oracle.encode(snappy_raw["html"]))
    good = bytes(enc)
    bad_magic = bytearray(enc)
    bad_magic[0:4] = b"bvxq"
    # n_lmds field: bits [40,60) of the u64 at +8 (fse/block.rs:108-136) -> 0xFFFFF
    bad_lmd = bytearray(enc)
    q = int.from_bytes(bad_lmd[8:16], "little") | (0xFFFFF << 40)
    bad_lmd[8:16] = q.to_bytes(8, "little")
    # n_literals: bits [0,20) -> 40 004 (a multiple of 4 above LITERALS_PER_BLOCK)
    bad_lit = bytearray(enc)
    q = (int.from_bytes(bad_lit[8:16], "little") & ~0xFFFFF) | 40004
    bad_lit[8:16] = q.to_bytes(8, "little")
    cases = [bytes(bad_magic), good, bytes(bad_lmd), bytes(bad_lit)]
    outs, st = ctx.decode_batch(cases, caps=[1 << 20] * 4)
    assert list(st) == [2, 0, 21, 17]
    assert [oracle.decode_status(c, 1 << 20) for c in cases] == [2, 0, 21, 17]
    assert ctx.error_detail(0) == int.from_bytes(b"bvxq", "little")
    assert ctx.error_detail(1) == 0
    assert ctx.error_detail(2) == 0xFFFFF
    assert ctx.error_detail(3) == 40004
    import lzfse_rust_amd as m
    with pytest.raises(m.LzfseError) as ei:
        m.LzfseDecoder(context=ctx).decode_bytes(bytes(bad_magic), bytearray())
    assert ei.value.status == 2 and ei.value.detail == int.from_bytes(b"bvxq", "little")
