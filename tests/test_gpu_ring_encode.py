"""GPU parity tests (ring / stream encode): LzfseRingEncoder::encode, LzfseWriter, LzfseWriterBytes over the HIP path give,
byte for byte, the streams the reference's ring front end gives (encode/frontend_ring.rs, restated in the oracle's
"ring frontend" and pinned by the reference's in-file KATs: tests/test_oracle_ring.py) -- another parse than the slice
encoder's: rounds over a 512 KiB ring, capped and coarse forward lengths, the ring head as the backward limit, literals
that pass the head pushed in pieces."""
import io

import numpy as np
import pytest

import test_kit as tk
from oracle_py import rng_gen_vec
from test_oracle import EOS, ZERO_4097, raw_block

pytestmark = pytest.mark.gpu

RING, BLK = 0x80000, 0x4000


@pytest.fixture(scope="module")
def ctx():
    import lzfse_rust_amd as m
    return m.Context(0)


def _ring(ctx, data):
    outs, st = ctx.encode_batch([data], ring=True)
    assert st == [0]
    return outs[0].tobytes()


def _text(n, seed=7):
    rng = np.random.default_rng(seed)
    words = [bytes(rng.integers(97, 123, size=int(k), dtype=np.uint8)) for k in rng.integers(2, 9, size=500)]
    out = b" ".join(words[int(i)] for i in rng.integers(0, 500, size=n // 4 + 16))
    return out[:n]


# ---- the reference's byte-exact vectors through the device library: frontend_ring.rs:768-859 ----

def test_ring_kats(ctx):
    for n in (0, 1, 20):
        assert _ring(ctx, bytes(n)) == raw_block(bytes(n))
    assert _ring(ctx, bytes(21)) == bytes([0x62, 0x76, 0x78, 0x6E, 0x15, 0, 0, 0, 0x0C, 0, 0, 0, 0x68, 0x01, 0x00, 0xFC,
                                            0x06, 0, 0, 0, 0, 0, 0, 0]) + EOS
    assert _ring(ctx, bytes(4096)) == (bytes([0x62, 0x76, 0x78, 0x6E, 0x00, 0x10, 0, 0, 0x2B, 0, 0, 0, 0x68, 0x01, 0x00])
                                       + bytes([0xF0, 0xFF]) * 15 + bytes([0xF0, 0x06, 0x06, 0, 0, 0, 0, 0, 0, 0]) + EOS)
    assert _ring(ctx, bytes(4097)) == ZERO_4097
    assert _ring(ctx, rng_gen_vec(0, 4096))[:4] == b"bvx-"
    assert _ring(ctx, rng_gen_vec(0, 4097))[:4] == b"bvx2"


# ---- every Snappy file: bytes == ring oracle ----

def test_ring_snappy_files_match_the_oracle(ctx, oracle, snappy_raw):
    names = sorted(snappy_raw)
    outs, st = ctx.encode_batch([snappy_raw[k] for k in names], ring=True)
    assert st == [0] * len(names)
    for k, o in zip(names, outs):
        assert o.tobytes() == oracle.ring_encode(snappy_raw[k]), k


def _cases():
    text = _text(2_400_000)
    yield "text_2_4M", text
    yield "text_ring_exact", text[:RING]
    yield "text_ring_minus_1", text[:RING - 1]
    yield "text_ring_plus_1", text[:RING + 1]
    yield "text_ring_plus_blk", text[:RING + BLK]
    yield "text_ring_plus_blk_minus_1", text[:RING + BLK - 1]
    yield "noise_1_2M", tk.seq(1_200_000, seed=3)                       # a literal desert: push_literal_overflow every round
    yield "low_entropy_1M", tk.seq(1_000_000, seed=5, mask=0x01010101)   # many candidates that run to the end of the input
    yield "low_entropy_bits", tk.seq(700_000, seed=6, mask=0x00010001)
    yield "zeros_1_5M", bytes(1_500_000)                                 # LONG_MATCH_LEN caps every round's match
    yield "zeros_then_text", bytes(700_000) + text[:300_000] + bytes(400_000)
    yield "noise_sandwich", text[:40_000] + tk.seq(900_000, seed=9) + text[:40_000]
    per = bytes(np.random.default_rng(2).integers(0, 256, size=300_000, dtype=np.uint8))
    yield "period_300k", per * 4                                         # matches longer than LONG_MATCH_LEN at distance 300 000? no: beyond the window
    per2 = bytes(np.random.default_rng(3).integers(0, 256, size=250_000, dtype=np.uint8))
    yield "period_250k", per2 * 5                                        # distance 250 000 < 262 139: every round's match hits the cap
    yield "far_match_long_literals", _far_match_case()


def _far_match_case():
    """A match at nearly the maximum distance behind a long run of literals: its backward extension stops at the ring
    head (frontend_ring.rs:482), not at the start of the input."""
    rng = np.random.default_rng(17)
    a = bytes(rng.integers(0, 256, size=20_000, dtype=np.uint8))
    pad1 = tk.seq(560_000, seed=21)
    pad2 = tk.seq(262_000 - 20_000, seed=22)
    return pad1 + a + pad2 + a + tk.seq(300_000, seed=23)


@pytest.mark.parametrize("name,data", list(_cases()), ids=[n for n, _ in _cases()])
def test_ring_cases_match_the_oracle(ctx, oracle, name, data):
    got = _ring(ctx, data)
    exp = oracle.ring_encode(data)
    assert got == exp
    assert oracle.decode(got) == data


def test_ring_every_small_size_class(ctx, oracle):
    """The host-side size classes as the ring front end cuts them (flush_select, frontend_ring.rs:297-342): Vn's
    match_short visits one more position than the slice loop and compares past the end of the input."""
    rng = np.random.default_rng(23)
    datas = []
    for n in list(range(0, 300)) + list(range(300, 4097, 61)) + [4095, 4096]:
        k = int(rng.choice([2, 4, 16, 256]))
        datas.append(bytes(rng.integers(0, k, size=n, dtype=np.uint8)))
    outs, st = ctx.encode_batch(datas, ring=True)
    assert not any(st)
    n_other = 0
    for d, o in zip(datas, outs):
        exp = oracle.ring_encode(d)
        assert o.tobytes() == exp, len(d)
        n_other += exp != oracle.encode(d)
    assert n_other > 10


def test_ring_low_entropy_sweep(ctx, oracle):
    """Sizes around 4097 .. 300 000 over tiny alphabets: where the ring's candidate choice differs from the slice parse
    (candidates that run to the end of the input are measured past it)."""
    rng = np.random.default_rng(29)
    datas = []
    for _ in range(160):
        n = int(rng.integers(4097, 300_000))
        k = int(rng.choice([2, 3, 4, 16]))
        datas.append(bytes(rng.integers(0, k, size=n, dtype=np.uint8)))
    outs, st = ctx.encode_batch(datas, ring=True)
    assert not any(st)
    n_other = 0
    for d, o in zip(datas, outs):
        exp = oracle.ring_encode(d)
        assert o.tobytes() == exp, len(d)
        n_other += exp != oracle.encode(d)
    assert n_other >= 1


def test_ring_reference_patterns(ctx, oracle):
    """The generators of the reference's integration suite (test/src/pattern_*.rs, patchwork_*.rs, random_*.rs run
    `encode_writer_bytes` too: test/src/ops.rs:73-84), a sample of each, against the ring oracle."""
    datas = [tk.patchwork(seed, 0x40, 20) for seed in range(0, 24, 3)]
    datas += [tk.patchwork(seed, 0x200, 24) for seed in range(4)]
    datas += [tk.seq(n, seed=n) for n in (5000, 70_000, 600_000)]
    datas += [tk.seq(n, seed=n, mask=0x03030303) for n in (5000, 70_000, 600_000)]
    datas += [tk.cycle(n) for n in (4097, 100_000, RING + 5)]
    datas += [tk.useq(200_000)]
    datas += [tk.build_match_inc(9000, 5000, 4990, 3000), tk.build_match_dec(9000, 5000, 4990, 3000)]
    outs, st = ctx.encode_batch(datas, ring=True)
    assert not any(st)
    for i, (d, o) in enumerate(zip(datas, outs)):
        assert o.tobytes() == oracle.ring_encode(d), i


# ---- LzfseWriter / LzfseRingEncoder::encode: any piece sizes, test/src/fuzz_write.rs:8-33 ----

def test_fuzz_write(ctx, oracle):
    """fuzz_write.rs: 2 MiB of Seq written in random-length pieces ((gen % 0x20) * multiplier) must give the stream of
    one encode() call -- and here that stream is the ring oracle's. Multipliers 0x100 / 0x1000 / 0x10000 over a few seeds
    (1 and 0x10 are two million ctypes calls a seed; the piece size never reaches the device: feed only stores)."""
    import lzfse_rust_amd as m
    data = tk.seq(0x0020_0000)
    enc = m.LzfseRingEncoder(context=ctx)
    base = io.BytesIO()
    u, v = enc.encode(io.BytesIO(data), base)
    assert (u, v) == (len(data), len(base.getvalue()))
    assert base.getvalue() == oracle.ring_encode(data)
    for mult, seeds in ((0x100, 2), (0x1000, 3), (0x10000, 4)):
        for seed in range(seeds):
            out = bytearray()
            w = enc.writer_bytes(out)
            rng = tk.Rng(seed)
            pos = 0
            while pos < len(data):
                n = min((rng.gen() % 0x20) * mult, len(data) - pos)
                w.write(data[pos:pos + n])
                pos += n
            assert w.finalize() is out
            assert bytes(out) == base.getvalue(), (mult, seed)


def test_writer_small_pieces_and_sink_errors(ctx, oracle):
    import lzfse_rust_amd as m
    data = _text(70_000, seed=3)
    enc = m.LzfseRingEncoder(context=ctx)
    out = io.BytesIO()
    w = enc.writer(out)
    for i in range(0, len(data), 7):
        assert w.write(data[i:i + 7]) == len(data[i:i + 7])
    w.flush()
    assert w.finalize() is out
    assert out.getvalue() == oracle.ring_encode(data)
    # an empty stream
    out = bytearray(b"head")
    assert enc.writer_bytes(out).finalize() == bytearray(b"head") + raw_block(b"")
    # the sink's failure comes back as the sink's exception (io::Error of the inner writer, ring_short_writer.rs)

    class Bad:
        def write(self, b):
            raise OSError("disk full")

    w = enc.writer(Bad())
    w.write(data)
    with pytest.raises(OSError):
        w.finalize()
    # ... and the encoder object is still good
    dst = bytearray()
    assert enc.encode_bytes(data, dst) == len(dst) and bytes(dst) == oracle.encode(data)


class _Hip:
    """hipMalloc / hipMemcpy of the runtime the library itself is linked against (torch ships another copy of it)."""

    def __init__(self):
        import ctypes as C
        self.C = C
        self.rt = C.CDLL("libamdhip64.so")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipFree.argtypes = [C.c_void_p]

    def alloc(self, n):
        p = self.C.c_void_p()
        assert self.rt.hipMalloc(self.C.byref(p), n) == 0
        return p.value

    def upload(self, arr):
        p = self.alloc(arr.size)
        assert self.rt.hipMemcpy(p, arr.ctypes.data, arr.size, 1) == 0
        return p

    def download(self, p, n):
        out = np.empty(n, dtype=np.uint8)
        assert self.rt.hipMemcpy(out.ctypes.data, p, n, 2) == 0
        return out

    def free(self, p):
        self.rt.hipFree(p)


def test_ring_batch_device_resident(ctx, oracle, snappy_raw):
    """The device-resident form (what the bench times), ring and slice parses side by side on the same inputs."""
    hip = _Hip()
    names = sorted(snappy_raw)[:6]
    raws = [snappy_raw[k] for k in names] + [_text(1_300_000, seed=5)]
    offs, pos = [], 0
    for r in raws:
        offs.append(pos)
        pos += (len(r) + 255) & ~255
    src = np.zeros(pos, dtype=np.uint8)
    for o, r in zip(offs, raws):
        src[o:o + len(r)] = np.frombuffer(r, dtype=np.uint8)
    d_src = hip.upload(src)
    caps = [ctx._lib.lzfse_mi_encode_bound(len(r)) for r in raws]
    doffs = np.concatenate([[0], np.cumsum([(c + 255) & ~255 for c in caps])[:-1]]).astype(np.uint64)
    total = int(doffs[-1]) + caps[-1] + 256
    d_dst = hip.alloc(total)
    try:
        for ring in (True, False, True):
            ol, st = ctx.encode_batch_device(d_src, offs, [len(r) for r in raws], d_dst, doffs, caps, ring=ring)
            assert not st.any()
            host = hip.download(d_dst, total)
            for i, r in enumerate(raws):
                got = host[int(doffs[i]):int(doffs[i]) + int(ol[i])].tobytes()
                assert got == (oracle.ring_encode(r) if ring else oracle.encode(r)), (ring, i)
    finally:
        hip.free(d_src)
        hip.free(d_dst)


# ---- the stream encoder feeds the device a window at a time: same bytes as the whole input in one call ----

class _CountingSink:
    """collects the stream and remembers how much of it had arrived before finalize()"""

    def __init__(self):
        self.parts, self.calls = [], 0

    def write(self, b):
        self.parts.append(bytes(b))
        self.calls += 1

    def value(self):
        return b"".join(self.parts)


def _windowed(ctx, data, window, piece):
    import lzfse_rust_amd as m
    sink = _CountingSink()
    w = m.LzfseRingEncoder(context=ctx, window=window).writer(sink)
    for o in range(0, len(data), piece):
        w.write(data[o:o + piece])
    early = sum(map(len, sink.parts))
    w.finalize()
    return sink.value(), early


def _window_cases():
    text = _text(7_000_000, seed=11)
    yield "text_7M", text
    yield "noise_3M", tk.seq(3_000_000, seed=13)                          # literal deserts: the overflow pushes of the round ends
    yield "low_entropy_4M", tk.seq(4_000_000, seed=14, mask=0x01010101)
    yield "low_entropy_bits_3M", tk.seq(3_000_000, seed=15, mask=0x00010001)
    yield "zeros_6M", bytes(6_000_000)                                     # one block spans many MiB: windows without a final block
    yield "mixed", text[:1_500_000] + bytes(900_000) + tk.seq(800_000, seed=16) + text[2_000_000:3_300_000] + tk.seq(700_000, seed=17, mask=0x01010101)
    rep = np.tile(np.frombuffer(text[:70_001], dtype=np.uint8), 60).copy()
    rep[::4099] ^= 1
    yield "period_70001", rep.tobytes()


@pytest.mark.parametrize("name,data", list(_window_cases()), ids=[n for n, _ in _window_cases()])
def test_windows_give_the_bytes_of_the_whole_input(ctx, oracle, name, data):
    want = oracle.ring_encode(data)
    assert _ring(ctx, data) == want
    for window, piece in ((1 << 20, 300_001), (2 << 20, 1 << 20), (1 << 20, 4_000_000)):
        got, early = _windowed(ctx, data, window, piece)
        assert got == want, (name, window, piece)
        if name == "text_7M":
            assert early > len(want) // 2, "most of the stream must have left before finalize()"


def test_window_sizes_around_the_cut_rules(ctx, oracle):
    """lengths around a window's end and the ring's rounds (multiples of 16 KiB): the last 256 KiB + 16 KiB of a window
    are never final, and the very last window may be hardly longer than what was kept"""
    text = _text(3_300_000, seed=19)
    for n in (1 << 20, (1 << 20) + (1 << 19), (1 << 20) + (1 << 19) + 1, 2_097_152 - 1, 2_097_152, 2_097_152 + BLK, 2_097_152 + RING // 2 + BLK + 5,
              2_621_440, 2_621_441, 3_145_728 - BLK - 1, 3_300_000):
        data = text[:n]
        got, _ = _windowed(ctx, data, 1 << 20, 1 << 18)
        assert got == oracle.ring_encode(data), n


def test_windowed_stream_decodes_back(ctx):
    """32 MiB of text through 4 MiB windows: round trip, and the windows did leave early"""
    import lzfse_rust_amd as m
    data = _text(32 << 20, seed=23)
    got, early = _windowed(ctx, data, 4 << 20, 1 << 20)
    assert got == _ring(ctx, data)
    assert early > len(got) * 3 // 4
    out, st = ctx.decode_batch([got])
    assert st == [0] and out[0].tobytes() == data


def test_windows_cut_inside_long_matches(ctx, oracle):
    """Inputs whose bvx2 blocks span tens of MiB and end INSIDE an event: zeros (every event is a match of LONG_MATCH_LEN =
    105 LMDs, 10 000 is no multiple of that, so every block ends in the middle of a match) and runs of 2 400 .. 7 000 equal
    bytes (two to three LMDs per run). The window is cut inside the event: what is left of it opens the next window's first
    block, as the remainder opens the next block in Buffer::push (fse/buffer.rs:45-97)."""
    rng = np.random.default_rng(29)
    zeros = bytes(56_000_000)
    runs = np.repeat(rng.integers(0, 256, size=9000, dtype=np.uint8), rng.integers(2400, 7000, size=9000)).tobytes()
    for name, data in (("zeros", zeros), ("runs", runs)):
        want = oracle.ring_encode(data)
        got, early = _windowed(ctx, data, 4 << 20, 3_000_000)
        assert got == want, name
        assert early > 0, name


def test_windowed_writer_sink_failure_and_reuse(ctx, oracle):
    """a sink that fails in the middle of write(): its exception comes back from write() (io::Error of the inner writer),
    the handle stays failed, and the context encodes the next stream as if nothing had happened"""
    import lzfse_rust_amd as m
    data = _text(5_000_000, seed=31)

    class Bad:
        def __init__(self):
            self.n = 0

        def write(self, b):
            self.n += len(b)
            if self.n > 200_000:
                raise OSError("disk full")

    enc = m.LzfseRingEncoder(context=ctx, window=1 << 20)
    w = enc.writer(Bad())
    with pytest.raises(OSError):
        for o in range(0, len(data), 1 << 18):
            w.write(data[o:o + (1 << 18)])
    with pytest.raises(m.LzfseError):
        w.write(b"more")
    got, _ = _windowed(ctx, data, 1 << 20, 1 << 18)
    assert got == oracle.ring_encode(data)


def test_reserve_commit_is_feed_without_the_copy(ctx, oracle, snappy_raw):
    """lzfse_mi_estream_reserve / _commit (what LzfseRingEncoder::encode's copy(reader) becomes): the same bytes as feed,
    whatever the reads bring -- short reads, empty reads in the middle, a commit of less than the room -- and a commit of
    more than the room is refused."""
    import io
    import lzfse_rust_amd as m
    raw = (snappy_raw["lcet10.txt"] + snappy_raw["html"] + snappy_raw["kppkn.gtb"]) * 7          # 5 MB
    want = oracle.ring_encode(raw)

    class Choppy(io.RawIOBase):
        def __init__(self, data, sizes):
            self.d, self.p, self.sizes, self.k = data, 0, sizes, 0

        def readable(self):
            return True

        def readinto(self, b):
            n = min(len(b), self.sizes[self.k % len(self.sizes)], len(self.d) - self.p)
            self.k += 1
            b[:n] = self.d[self.p:self.p + n]
            self.p += n
            return n

    for sizes, window, read_size in (([1 << 20], 1 << 20, 1 << 20), ([70000, 1, 300000, 5], 2 << 20, 123457), ([1 << 22], 1 << 20, 1 << 22)):
        out = bytearray()
        u, v = m.LzfseRingEncoder(context=ctx, window=window, read_size=read_size).encode(Choppy(raw, sizes), _Into(out))
        assert (u, v) == (len(raw), len(want)) and bytes(out) == want, (sizes, window)
    w = m.LzfseRingEncoder(context=ctx, window=1 << 20).writer_bytes(bytearray())
    view = w._reserve(1000)
    n = len(view)
    view.release()
    with pytest.raises(m.LzfseError):
        w._commit(n + 1)
    w._commit(0)
    w.finalize()


class _Into:
    def __init__(self, out):
        self.out = out

    def write(self, b):
        self.out += b
