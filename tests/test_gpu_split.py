"""GPU parity tests for large batches, which the library cuts into sub-batches that run side by side on two HIP
streams (split batches): results, statuses and ordering must be those of one call."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import lzfse_rust_amd as m
    return m.Context(0)


def _mixed(snappy_raw):
    rng = np.random.default_rng(21)
    big = [snappy_raw[k] for k in ("urls.10K", "plrabn12.txt", "html_x_4", "lcet10.txt", "kppkn.gtb", "fireworks.jpeg")] * 3
    tiny = [bytes(n) for n in (0, 1, 20, 21, 4096, 4097)] + [rng.integers(0, 256, size=n, dtype=np.uint8).tobytes() for n in (7, 300, 4096, 5000)]
    raws = []
    for i, b in enumerate(big):  # interleave so that both lanes see every size class
        raws.append(b)
        if i < len(tiny):
            raws.append(tiny[i])
    return raws


def test_split_encode_matches_oracle(ctx, oracle, snappy_raw):
    raws = _mixed(snappy_raw)
    assert len(raws) >= 8 and sum(map(len, raws)) >= 4 << 20  # large enough to be split
    # (round 5: a call below 448 MiB runs as one pass unless told otherwise -- the lanes are asked for here)
    ctx.set_option("encode_lanes", 2)
    ctx.enable_timing(True)
    try:
        outs, st = ctx.encode_batch(raws)
        t = ctx.timings()
    finally:
        ctx.enable_timing(False)
        ctx.set_option("encode_lanes", 0)
    assert t["enc_cand"][1] >= 2, t  # sub-batches ran side by side
    for r, o, e in zip(raws, outs, st):
        assert e == 0
        assert o.tobytes() == oracle.encode(r), len(r)


def test_split_decode_statuses_per_stream(ctx, oracle, snappy_raw):
    raws = _mixed(snappy_raw)
    encs = [oracle.encode(r) for r in raws]
    caps = [max(len(r), 16) for r in raws]
    # damage some streams in both lanes, and give two streams too little room
    bad = {3: "flip", 8: "trunc", 13: "cap", 20: "flip", 21: "cap"}
    for i, kind in bad.items():
        if kind == "flip":
            e = bytearray(encs[i]); e[len(e) // 2] ^= 0x40; encs[i] = bytes(e)
        elif kind == "trunc":
            encs[i] = encs[i][:-5]
        else:
            caps[i] = max(len(raws[i]) - 1, 0)
    outs, st = ctx.decode_batch(encs, caps=caps)
    for i, (r, e, o, s) in enumerate(zip(raws, encs, outs, st)):
        es = oracle.decode_status(e, caps[i])
        assert (s == 0) == (es == 0), (i, s, es)
        if s == 0:
            assert o.tobytes() == oracle.decode(e, cap=max(caps[i], 1))
        if i not in bad:
            assert s == 0 and o.tobytes() == r


def test_split_lane_count_option(snappy_raw, oracle):
    """LZFSE_MI_OPT_ENCODE_LANES / _DECODE_LANES: every lane count leaves the bytes of an unsplit call."""
    import lzfse_rust_amd as m
    c = m.Context(0)
    r = snappy_raw["html"]
    want = oracle.encode(r)
    for lanes in (1, 2, 3, 4, 0):
        c.set_option("encode_lanes", lanes)
        c.set_option("decode_lanes", lanes)
        many, st = c.encode_batch([r] * 96)        # 9.8 MB: enough for four lanes
        assert all(e == 0 for e in st)
        assert all(o.tobytes() == want for o in many)
        back, st = c.decode_batch([want] * 96)
        assert all(e == 0 for e in st) and all(o.tobytes() == r for o in back)
    for stagger in (1, 0):   # lanes one after the other (LaneGate) / together (the default)
        c.set_option("stagger", stagger)
        many, st = c.encode_batch([r] * 96)
        assert all(e == 0 for e in st) and all(o.tobytes() == want for o in many)
    with pytest.raises(m.LzfseError):
        c.set_option("diag_lz_path", 1)   # the product library has no diagnostic options


def test_many_mid_size_streams_take_the_per_stream_lz_path(ctx, snappy_raw):
    """100 streams of 2.2 MiB (50 per lane): each is large enough for the pointer-jumping LZ path, but with this
    many the cost model keeps them on one workgroup per stream (api.hip, jump_mode); one 6 MiB stream alone jumps."""
    base = (snappy_raw["alice29.txt"] + snappy_raw["html"] + snappy_raw["kppkn.gtb"]) * 6
    raws = []
    for i in range(100):
        a = np.frombuffer(base[: 2200 * 1024 + 997 * i], dtype=np.uint8).copy()
        a[i % 251::251] ^= np.uint8(1 + i)
        raws.append(a.tobytes())
    outs, st = ctx.encode_batch(raws)
    assert all(e == 0 for e in st)
    ctx.enable_timing(True)
    dec, st2 = ctx.decode_batch([o.tobytes() for o in outs])
    t = ctx.timings()
    ctx.enable_timing(False)
    assert all(e == 0 for e in st2)
    for r, o in zip(raws, dec):
        assert o.tobytes() == r
    assert "dec_jump_rounds" not in t and "dec_lz" in t, t
    big = base[: 6 << 20]
    o1, s1 = ctx.encode_batch([big])
    ctx.enable_timing(True)
    d1, s2 = ctx.decode_batch([o1[0].tobytes()])
    t1 = ctx.timings()
    ctx.enable_timing(False)
    assert s1[0] == 0 and s2[0] == 0 and d1[0].tobytes() == big
    assert "dec_jump_rounds" in t1, t1
