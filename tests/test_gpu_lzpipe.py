"""GPU parity tests (decode, LZ stage with several workgroups per stream): tickets of one LMD group each, the independent
part of a tile done ahead of its turn, errors raised in stream order -- the bytes and the status codes of the one-workgroup
kernel and of the oracle, for every number of workgroups per stream and both tile sizes."""
import glob
import os

import numpy as np
import pytest

from oracle_py import rng_gen_vec

pytestmark = pytest.mark.gpu

PIPES = (2, 5, 16, 0x103, 0x108)      # K | variant << 8


@pytest.fixture(scope="module")
def ctx():
    import lzfse_rust_amd as m
    c = m.Context(0)
    yield c
    c.set_option("decode_pipe", 0)


def _all(ctx, srcs, caps=None):
    res = {}
    for pipe in (1,) + PIPES:
        ctx.set_option("decode_pipe", pipe)
        outs, st = ctx.decode_batch(srcs, caps=caps)
        res[pipe] = ([o.tobytes() for o in outs], list(st))
    ctx.set_option("decode_pipe", 0)
    return res


def test_pipe_fixtures_and_oracle_streams(ctx, oracle, golden_dir, snappy_raw):
    fs = []
    for sub in ("snappy", "special", "mutate"):
        fs += sorted(glob.glob(os.path.join(golden_dir, sub, "*.lzfse")))
    srcs = [open(f, "rb").read() for f in fs]
    page = rng_gen_vec(9, 250000)
    raws = [snappy_raw["lcet10.txt"] * 4, page * 6, bytes(1 << 20), b"abc" * 300000, bytes(range(256)) * 3000,
            rng_gen_vec(3, 700000), snappy_raw["html"] + page + snappy_raw["html"], b"", b"x", bytes(5000)]
    srcs += [oracle.encode(r) for r in raws]
    want = [(oracle.decode(s, cap=1 << 24) if oracle.decode_status(s, 1 << 24) == 0 else None, oracle.decode_status(s, 1 << 24))
            for s in srcs]
    res = _all(ctx, srcs, caps=[1 << 24] * len(srcs))
    for pipe, (outs, st) in res.items():
        for i, (o, e) in enumerate(zip(outs, st)):
            assert e == want[i][1], (pipe, i, e, want[i][1])
            if e == 0:
                assert o == want[i][0], (pipe, i)


def test_pipe_errors_in_stream_order(ctx, oracle, golden_dir, snappy_raw):
    """Damaged and cut multi-block streams, destinations that are too small: the first error in stream order, whichever
    workgroup meets it."""
    rng = np.random.default_rng(31)
    big = oracle.encode(snappy_raw["lcet10.txt"] * 3)
    cases, caps = [], []
    for _ in range(120):
        m = bytearray(big)
        r = rng.random()
        if r < 0.3:
            m = m[: int(rng.integers(4, len(m)))]
        else:
            for _ in range(int(rng.integers(1, 4))):
                m[int(rng.integers(0, len(m)))] ^= 1 << int(rng.integers(0, 8))
        cases.append(bytes(m))
        caps.append(1 << 22 if rng.random() < 0.8 else int(rng.integers(1, 1300000)))
    for k in ("raw", "vx1", "vx2", "vxn"):
        base = open(os.path.join(golden_dir, "mutate", k + ".lzfse"), "rb").read()
        for i in rng.choice(len(base), size=min(40, len(base)), replace=False):
            m = bytearray(base)
            m[i] ^= 1 << int(rng.integers(0, 8))
            cases.append(bytes(m))
            caps.append(1 << 20)
    want = [oracle.decode_status(c_, cap) for c_, cap in zip(cases, caps)]
    assert sum(1 for w in want if w) > 100
    res = _all(ctx, cases, caps=caps)
    for pipe, (outs, st) in res.items():
        for i, e in enumerate(st):
            assert e == want[i], (pipe, i, e, want[i])
            if e == 0:
                assert outs[i] == oracle.decode(cases[i], cap=caps[i]), (pipe, i)


def test_pipe_is_chosen_for_few_mid_size_streams(ctx, oracle, snappy_raw):
    """64 streams of 1 MiB: the default takes the pipelined kernel (and says so in the stage timings)."""
    raw = (snappy_raw["lcet10.txt"] * 3)[: 1 << 20]
    enc = oracle.encode(raw)
    ctx.set_option("decode_pipe", 0)
    outs, st = ctx.decode_batch([enc] * 64)
    assert all(e == 0 for e in st) and all(o.tobytes() == raw for o in outs)
    # the hand-over's start-up self-test (dec_lzp_selftest_kernel) passed on this device, and no launch has refused since
    assert ctx.pipe_refusals() == 0


def test_pipe_refuses_streams_spread_over_xcds(oracle, snappy_raw):
    """The hand-over goes through ONE XCD's L2, so every workgroup that works on a stream checks the XCC id it runs on against
    the stream's. The diagnostic build can make them disagree: nothing may be taken from that launch -- the streams are
    decoded again by the one-workgroup kernel (stage dec_lz_again), and the context stops using the pipelined kernel."""
    import lzfse_rust_amd as m
    diag_ctx = m.Context(0, diag=True)      # its own context: the test leaves it without the pipelined kernel
    raws = [snappy_raw["lcet10.txt"] * 2, snappy_raw["urls.10K"], snappy_raw["html"] * 3, snappy_raw["alice29.txt"]]
    encs = [oracle.encode(r) for r in raws]
    bad = bytearray(encs[1]); bad[len(bad) // 2] ^= 0x40
    encs.append(bytes(bad))
    want = [oracle.decode_status(e, 1 << 22) for e in encs]
    diag_ctx.enable_timing(True)
    try:
        diag_ctx.set_option("decode_pipe", 0x104)
        outs, st = diag_ctx.decode_batch(encs, caps=[1 << 22] * len(encs))
        assert "dec_lz_again" not in diag_ctx.timings()
        assert list(st) == want and all(o.tobytes() == r for o, r in zip(outs, raws))
        assert diag_ctx.pipe_refusals() == 0
        diag_ctx.set_option("diag_pipe_scatter", 1)
        outs, st = diag_ctx.decode_batch(encs, caps=[1 << 22] * len(encs))
        assert diag_ctx.timings()["dec_lz_again"][1] == 1
        assert diag_ctx.pipe_refusals() == 1      # ... and the caller can see that it was given up
        assert list(st) == want and all(o.tobytes() == r for o, r in zip(outs, raws))
        outs, st = diag_ctx.decode_batch(encs, caps=[1 << 22] * len(encs))      # given up: the plain kernel from the start
        assert "dec_lz_again" not in diag_ctx.timings()
        assert list(st) == want and all(o.tobytes() == r for o, r in zip(outs, raws))
    finally:
        diag_ctx.set_option("diag_pipe_scatter", 0)
        diag_ctx.set_option("decode_pipe", 0)
        diag_ctx.enable_timing(False)


def test_hand_over_is_checked_in_the_diagnostic_build(oracle, snappy_raw):
    """Diagnostic build: every ticket leaves a checksum of the bytes it wrote beside `done`, and the next ticket reads
    those bytes back the way it reads all earlier output and compares. On sound hardware nothing is ever refused; a
    deliberately wrong sum (diag_pipe_scatter = 2) must be caught: the launch is thrown away, the streams are decoded again
    by the one-workgroup kernel and the context gives the pipelined kernel up."""
    import lzfse_rust_amd as m
    ctx = m.Context(0, diag=True)
    raws = [snappy_raw["lcet10.txt"] * 3, snappy_raw["plrabn12.txt"] * 2, snappy_raw["urls.10K"] * 2, snappy_raw["kppkn.gtb"]]
    encs = [oracle.encode(r) for r in raws]
    ctx.enable_timing(True)
    for pipe in (0x104, 0x102, 0x004):      # K = 4 and 2 with 1 024-thread tickets, K = 4 with 256-thread tickets
        ctx.set_option("decode_pipe", pipe)
        outs, st = ctx.decode_batch(encs)
        assert list(st) == [0] * len(encs) and all(o.tobytes() == r for o, r in zip(outs, raws))
        assert "dec_lz_again" not in ctx.timings() and ctx.pipe_refusals() == 0
    ctx.set_option("diag_pipe_scatter", 2)
    outs, st = ctx.decode_batch(encs)
    assert ctx.timings()["dec_lz_again"][1] == 1 and ctx.pipe_refusals() == 1
    assert list(st) == [0] * len(encs) and all(o.tobytes() == r for o, r in zip(outs, raws))
