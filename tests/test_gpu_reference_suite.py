"""The reference's integration suite (test/src/*.rs), restated against the C ABI (no Rust toolchain
exists here, so the crate's own tests cannot be compiled against the shim; these are the same inputs through the same
entry points). Where the reference only asserts a round trip (test/src/buddy.rs:50-66), every encoder output here must
in addition EQUAL the oracle's bytes, and every damaged stream's status code the oracle's.

    len.rs            every length 0 .. 0x4000 of Seq::masked(Rng(n), 0x0303)
    pattern_1 .. _6   zeros of growing size, shrinking non-overlapping matches, head pad + zeros, Useq (no 4-byte match
                      at all), i mod v, random short repeats
    patchwork_0 / _1  random self-copies, long and short files
    random_0 .. _2    low-entropy noise, masks 0x01010101 / 0x02020202 / 0x03030303
    mutate_0 .. _7    all four data/mutate fixtures: every bit, every byte value, 0x0000 / 0xFFFF words and double words,
                      compound random bit / byte damage, every truncation, every extension with a second copy
The loops the reference marks "expensive" are kept whole except where a scalar Python generator or the single-threaded
CPU check would dominate the suite's run time (stated at each test)."""
import hashlib
import os

import numpy as np
import pytest

import test_kit as tk

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import lzfse_rust_amd as m
    return m.Context(0)


def _batches(items, max_bytes=192 << 20, max_count=8192):
    cur, size = [], 0
    for it in items:
        if cur and (size + len(it) > max_bytes or len(cur) >= max_count):
            yield cur
            cur, size = [], 0
        cur.append(it)
        size += len(it)
    if cur:
        yield cur


def encode_decode(ctx, oracle, inputs, label):
    """Buddy::encode_decode for a list of inputs, batched: GPU stream == oracle stream, GPU decode of it == input."""
    k = 0
    for batch in _batches(inputs):
        encs, st = ctx.encode_batch(batch)
        assert all(s == 0 for s in st), (label, [i for i, s in enumerate(st) if s][:5])
        for j, (raw, e) in enumerate(zip(batch, encs)):
            want = oracle.encode(raw)
            assert e.tobytes() == want, f"{label}: input {k + j} ({len(raw)} bytes) encodes differently from the oracle"
        decs, st = ctx.decode_batch([e.tobytes() for e in encs], caps=[len(r) for r in batch])
        assert all(s == 0 for s in st), label
        for j, (raw, d) in enumerate(zip(batch, decs)):
            assert d.tobytes() == raw, f"{label}: input {k + j} does not round-trip"
        k += len(batch)


# ---------------------------------------------------------------------------------------------- len.rs

def test_len_every_length_below_0x4000(ctx, oracle):
    """test/src/len.rs:15-27."""
    inputs = [tk.seq(n, seed=n, mask=0x0000_0303) for n in range(0x4000)]
    encode_decode(ctx, oracle, inputs, "len")


# ---------------------------------------------------------------------------------------------- pattern_*.rs

def test_pattern_1_zeros_growing(ctx, oracle):
    """pattern_1.rs:13-46: zeros of every length < 0x8000; 0 .. 0x80200 in steps of 0x100; every length 0x7FE00 .. 0x80200."""
    z = bytes(0x80200)
    encode_decode(ctx, oracle, [z[:n] for n in range(0x8000)], "pattern_1/0")
    encode_decode(ctx, oracle, [z[:n] for n in range(0, 0x80200, 0x100)], "pattern_1/1")
    encode_decode(ctx, oracle, [z[:n] for n in range(0x7FE00, 0x80200)], "pattern_1/2")


def test_pattern_2_shrinking_nonoverlapping_matches(ctx, oracle):
    """pattern_2.rs:13-26: 0x400 random bytes, then for u = 0x3FF .. 1 append a copy of the last u bytes; every stage."""
    v = bytearray(tk.seq(0x400))
    inputs = []
    for u in range(0x3FF, 0, -1):
        v += v[len(v) - u:]
        inputs.append(bytes(v))
    encode_decode(ctx, oracle, inputs, "pattern_2")


def test_pattern_3_head_pad_then_zeros(ctx, oracle):
    """pattern_3.rs:12-47: u random bytes + v zeros; u 1..8 x v 0..0x1000, and the 0x8000 grid to 0x80000."""
    pad = tk.seq(0x80000)
    inputs = [pad[:u] + bytes(v) for u in range(1, 9) for v in range(0x1000)]
    encode_decode(ctx, oracle, inputs, "pattern_3/0")
    grid = range(0, 0x80000 + 1, 0x8000)
    encode_decode(ctx, oracle, [pad[:u] + bytes(v) for u in grid for v in grid], "pattern_3/1")


def test_pattern_4_no_matching_4_byte_sequence(ctx, oracle):
    """pattern_4.rs:11-16: Useq, 1 MiB."""
    encode_decode(ctx, oracle, [tk.useq(0x100000)], "pattern_4")


def test_pattern_5_repeating_sequences(ctx, oracle):
    """pattern_5.rs:9-124: i mod v for v in 2..16, 32, 64; 1 MiB each."""
    idx = np.arange(0x100000, dtype=np.uint32)
    inputs = [(idx % v).astype(np.uint8).tobytes() for v in list(range(2, 17)) + [32, 64]]
    encode_decode(ctx, oracle, inputs, "pattern_5")


def test_pattern_6_random_short_repeats(ctx, oracle):
    """pattern_6.rs:12-35: literal runs of 1..32 followed by 0..255 repeats at distance l; 16 seeds, every 3rd stage."""
    literals = tk.seq(0x4000)
    inputs = []
    for seed in range(0x10):
        rng = tk.Rng(seed)
        data = bytearray()
        pos, stage = 0, 0
        while pos < len(literals):
            l = min(rng.gen() % 0x20 + 1, len(literals) - pos)
            data += literals[pos:pos + l]
            pos += l
            m = rng.gen() % 0x100
            for _ in range(m):
                data.append(data[len(data) - l])
            stage += 1
            if stage % 3 == 0 or pos >= len(literals):
                inputs.append(bytes(data))
    encode_decode(ctx, oracle, inputs, "pattern_6")


# ---------------------------------------------------------------------------------------------- patchwork_*.rs

def test_patchwork_0_long_files(ctx, oracle):
    """patchwork_0.rs:13-87: (rounds, shift) = (0x10000, 28), (0x10000, 26), (0x1000, 22), (0x100, 18); all 0x100 seeds (round 4:
    every 8th until then), one batch per (rounds, shift)."""
    for rounds, shift in ((0x10000, 28), (0x10000, 26), (0x1000, 22), (0x100, 18)):
        encode_decode(ctx, oracle, [tk.patchwork(seed, rounds, shift) for seed in range(0x100)], f"patchwork_0/{rounds:x}/{shift}")


def test_patchwork_1_short_files(ctx, oracle):
    """patchwork_1.rs:13-66: (0x100, 28), (0x80, 26), (0x10, 22); all 0x1000 seeds."""
    inputs = []
    for rounds, shift in ((0x100, 28), (0x80, 26), (0x10, 22)):
        inputs += [tk.patchwork(seed, rounds, shift) for seed in range(0x1000)]
    encode_decode(ctx, oracle, inputs, "patchwork_1")


# ---------------------------------------------------------------------------------------------- random_*.rs

@pytest.mark.parametrize("mask", [0x01010101, 0x02020202, 0x03030303])
def test_random_low_entropy(ctx, oracle, mask):
    """random_0/1/2.rs:12-33: 1 MiB of masked noise, seeds 0 .. 0x80; 4 KiB, seeds 0 .. 0x800."""
    encode_decode(ctx, oracle, [tk.seq(0x100000, seed, mask) for seed in range(0x80)], f"random/{mask:08x}/0")
    encode_decode(ctx, oracle, [tk.seq(0x1000, seed, mask) for seed in range(0x800)], f"random/{mask:08x}/1")


# ---------------------------------------------------------------------------------------------- mutate_*.rs

FIXTURES = ["raw", "vxn", "vx1", "vx2"]


def _fixture(golden_dir, name):
    d = open(os.path.join(golden_dir, "mutate", name + ".lzfse"), "rb").read()
    h = open(os.path.join(golden_dir, "mutate", name + ".hash"), "rb").read()
    return d, h


def blind_decode(ctx, oracle, cases, cap, label):
    """Buddy::blind_decode for a list of damaged streams: must return (never hang or fault); the status code equals the
    oracle's and whatever still decodes is byte-identical."""
    bad = []
    k = 0
    for batch in _batches(cases, max_count=16384):
        outs, st = ctx.decode_batch(batch, caps=[cap] * len(batch))
        for j, (c, o, e) in enumerate(zip(batch, outs, st)):
            es = oracle.decode_status(c, cap)
            if e != es:
                bad.append((k + j, e, es))
            elif e == 0:
                assert o.tobytes() == oracle.decode(c, cap=cap), (label, k + j)
        k += len(batch)
    assert not bad, f"{label}: {len(bad)} of {len(cases)} status codes differ (case, gpu, oracle): {bad[:10]}"


def _intact(ctx, data, digest):
    outs, st = ctx.decode_batch([data])
    assert st[0] == 0 and hashlib.sha256(outs[0].tobytes()).digest() == digest


@pytest.mark.parametrize("name", FIXTURES)
def test_mutate_0_every_bit(ctx, oracle, golden_dir, name):
    """mutate_0.rs:25-38."""
    data, digest = _fixture(golden_dir, name)
    cases = []
    for i in range(len(data)):
        for b in range(8):
            m = bytearray(data)
            m[i] ^= 1 << b
            cases.append(bytes(m))
    blind_decode(ctx, oracle, cases, 1 << 20, f"mutate_0/{name}")
    _intact(ctx, data, digest)


@pytest.mark.parametrize("name", FIXTURES)
def test_mutate_1_every_byte_value(ctx, oracle, golden_dir, name):
    """mutate_1.rs:24-36: data[index] ^= byte for every byte value at EVERY position (round 4: every 16th behind the header until
    then), in batches of 64 positions."""
    data, digest = _fixture(golden_dir, name)
    for p0 in range(0, len(data), 64):
        cases = []
        for i in range(p0, min(p0 + 64, len(data))):
            for b in range(1, 256):
                m = bytearray(data)
                m[i] ^= b
                cases.append(bytes(m))
        blind_decode(ctx, oracle, cases, 1 << 20, f"mutate_1/{name}/{p0}")
    _intact(ctx, data, digest)


@pytest.mark.parametrize("name", FIXTURES)
def test_mutate_2_3_min_max_words(ctx, oracle, golden_dir, name):
    """mutate_2.rs:24-40 (16-bit words 0x0000 / 0xFFFF at every position), mutate_3.rs:24-48 (32-bit)."""
    data, digest = _fixture(golden_dir, name)
    cases = []
    for width in (2, 4):
        for i in range(len(data) - width + 1):
            for fill in (0x00, 0xFF):
                m = bytearray(data)
                m[i:i + width] = bytes([fill]) * width
                cases.append(bytes(m))
    blind_decode(ctx, oracle, cases, 1 << 20, f"mutate_2_3/{name}")
    _intact(ctx, data, digest)


@pytest.mark.parametrize("name", FIXTURES)
def test_mutate_4_5_compound_random_damage(ctx, oracle, golden_dir, name):
    """mutate_4.rs:25-41 (bits), mutate_5.rs:25-41 (bytes; the reference's own index = n / 8 is kept): damage accumulates
    over 0x100 steps per seed; all 0x100 seeds (round 4: 24 of them until then)."""
    data, digest = _fixture(golden_dir, name)
    cases = []
    for seed in range(0x100):
        rng = tk.Rng(seed)
        m = bytearray(data)
        for _ in range(0x100):
            n = rng.gen() % (len(data) * 8)
            m[n // 8] ^= 1 << (n % 8)
            cases.append(bytes(m))
        rng = tk.Rng(seed)
        m = bytearray(data)
        for _ in range(0x100):
            n = rng.gen() % len(data)
            m[n // 8] ^= rng.gen() & 0xFF
            cases.append(bytes(m))
    blind_decode(ctx, oracle, cases, 1 << 20, f"mutate_4_5/{name}")
    _intact(ctx, data, digest)


@pytest.mark.parametrize("name", FIXTURES)
def test_mutate_6_7_truncated_and_extended(ctx, oracle, golden_dir, name):
    """mutate_6.rs:24-31: every proper prefix is an error; mutate_7.rs:24-33: so is the stream followed by any proper,
    non-empty prefix of a second copy."""
    data, digest = _fixture(golden_dir, name)
    cases = [data[:i] for i in range(len(data) - 1)]
    twin = data + data
    cases += [twin[:i] for i in range(len(data) + 1, len(twin))]
    outs, st = ctx.decode_batch(cases, caps=[1 << 20] * len(cases))
    assert all(s != 0 for s in st), [i for i, s in enumerate(st) if s == 0][:5]
    blind_decode(ctx, oracle, cases, 1 << 20, f"mutate_6_7/{name}")
    _intact(ctx, data, digest)
