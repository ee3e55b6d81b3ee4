"""GPU parity tests (streaming decode): LzfseRingDecoder::decode(reader, writer) over the HIP path -- the same bytes and
the same errors as the slice path, whatever the sizes of the pieces the input arrives in (decode/ring_decoder.rs:58-68)."""
import ctypes as C
import glob
import io
import os

import numpy as np
import pytest

from oracle_py import rng_gen_vec

pytestmark = pytest.mark.gpu

BIG_CAP = 64 << 20


@pytest.fixture(scope="module")
def ctx():
    import lzfse_rust_amd as m
    return m.Context(0)


class PieceReader:
    """read() hands out the stream in pieces of the given sizes (cycled), then b""."""

    def __init__(self, data, sizes):
        self.data, self.sizes, self.pos, self.k = bytes(data), list(sizes), 0, 0

    def read(self, _n):
        n = self.sizes[self.k % len(self.sizes)]
        self.k += 1
        piece = self.data[self.pos:self.pos + n]
        self.pos += len(piece)
        return piece


def _stream(ctx, src, sizes, window=0):
    import lzfse_rust_amd as m
    out = io.BytesIO()
    dec = m.LzfseRingDecoder(context=ctx, window=window)
    u, v = dec.decode(PieceReader(src, sizes), out)
    return u, v, out.getvalue()


def _status(ctx, src, sizes, window=0):
    import lzfse_rust_amd as m
    try:
        _stream(ctx, src, sizes, window)
        return 0
    except m.LzfseError as e:
        return e.status


def test_stream_fixtures_any_piece_size(ctx, oracle, golden_dir):
    """Every fixture of the reference (all block kinds), pieces from 1 byte to the whole file, windows from one block up."""
    fs = []
    for sub in ("snappy", "special", "mutate"):
        fs += sorted(glob.glob(os.path.join(golden_dir, sub, "*.lzfse")))
    rng = np.random.default_rng(5)
    for f in fs:
        src = open(f, "rb").read()
        bad = oracle.decode_status(src, BIG_CAP)
        if bad:      # special/null.vx2.lzfse: a block the reference rejects
            assert _status(ctx, src, [7]) == bad, f
            continue
        want = oracle.decode(src)
        for window in (0, 1, 100000):
            sizes = [int(x) for x in rng.integers(1, max(2, len(src) // 3), size=7)]
            u, v, got = _stream(ctx, src, sizes, window)
            assert got == want, (f, window, sizes)
            assert (u, v) == (len(src), len(want)), f
    small = open(os.path.join(golden_dir, "mutate", "vx2.lzfse"), "rb").read()
    assert _stream(ctx, small, [1])[2] == oracle.decode(small)      # byte by byte
    assert _stream(ctx, b"bvx$", [1]) == (4, 0, b"")


def test_stream_windows_carry_the_match_window(ctx, oracle, snappy_raw):
    """Matches reach back up to 262 139 bytes across the boundary of two device calls: a 250 000-byte random page repeated
    (every match is about that far), text, and a long run; default and small windows."""
    page = rng_gen_vec(9, 250000)
    raws = [page * 40, snappy_raw["lcet10.txt"] * 25, bytes(9 << 20) + page + bytes(1 << 20) + page]
    for raw in raws:
        enc = oracle.encode(raw)
        for window, sizes in ((0, [1 << 20]), (300000, [70001, 13]), (1 << 20, [len(enc)])):
            u, v, got = _stream(ctx, enc, sizes, window)
            assert got == raw and (u, v) == (len(enc), len(raw)), (len(raw), window)


def test_stream_errors_match_the_slice_path(ctx, oracle, golden_dir, snappy_raw):
    """Damaged, cut and over-long streams: the status of LzfseDecoder::decode_bytes on the whole input, for any piece size
    (mutate_0.rs; decode/decoder.rs:93-95 for bytes behind bvx$)."""
    rng = np.random.default_rng(23)
    cases = []
    for k in ("raw", "vx1", "vx2", "vxn"):
        base = open(os.path.join(golden_dir, "mutate", k + ".lzfse"), "rb").read()
        for i in rng.choice(len(base), size=min(60, len(base)), replace=False):
            m = bytearray(base)
            m[i] ^= 1 << int(rng.integers(0, 8))
            cases.append(bytes(m))
    big = oracle.encode(snappy_raw["lcet10.txt"] * 3)      # 30+ blocks
    for _ in range(40):
        m = bytearray(big)
        if rng.random() < 0.4:
            m = m[: int(rng.integers(0, len(m)))]
        else:
            for _ in range(int(rng.integers(1, 4))):
                m[int(rng.integers(0, len(m)))] ^= 1 << int(rng.integers(0, 8))
        cases.append(bytes(m))
    enc = oracle.encode(snappy_raw["html"])
    cases += [enc[:-1], enc[:-4], enc[:-5], enc + b"\0", enc + enc, b"abcd" + enc, enc[:100], b"", b"bvx", b"bvx$bvx$", big + b"x"]
    n_err = 0
    for c in cases:
        want = oracle.decode_status(c, BIG_CAP)
        if want == 6:      # the oracle's buffer, not the stream: a header that promises more than BIG_CAP
            continue
        n_err += want != 0
        for window, sizes in ((0, [len(c) + 1]), (50000, [int(x) for x in rng.integers(1, 5000, size=5)])):
            got = _status(ctx, c, sizes, window)
            assert got == want, (got, want, len(c), window)
    assert n_err > 100
    assert _status(ctx, enc + b"\0", [len(enc), 1]) == oracle.decode_status(enc + b"\0", BIG_CAP) != 0


def test_stream_output_before_an_error_is_the_true_prefix(ctx, oracle, snappy_raw):
    """What the sink received before a late error is what the stream decodes to up to there (the ring decoder flushes as it
    goes), and the error is sticky."""
    import lzfse_rust_amd as m
    raw = snappy_raw["lcet10.txt"] * 12
    enc = bytearray(oracle.encode(raw))
    enc[-3000] ^= 0x55      # in the last block
    out = io.BytesIO()
    with pytest.raises(m.LzfseError):
        m.LzfseRingDecoder(context=ctx, window=1 << 20).decode(PieceReader(enc, [100000]), out)
    got = out.getvalue()
    assert len(got) >= 3 << 20 and raw.startswith(got)
    L = ctx._lib
    h = C.c_void_p()
    assert L.lzfse_mi_dstream_create(ctx._h, 0, C.byref(h)) == 0
    from lzfse_rust_amd import _native
    cb = _native.WRITE_FN(lambda _u, _p, _n: 0)
    a = np.frombuffer(b"bvxQ....", dtype=np.uint8)
    st = L.lzfse_mi_dstream_feed(h, a.ctypes.data, a.size, 0, cb, None)
    assert st == oracle.decode_status(b"bvxQ....", 100) != 0
    assert L.lzfse_mi_dstream_feed(h, a.ctypes.data, a.size, 1, cb, None) == st
    L.lzfse_mi_dstream_destroy(h)


def test_stream_sink_failure_travels_back(ctx, oracle, snappy_raw):
    class Full:
        def write(self, _b):
            raise OSError("disk full")

    import lzfse_rust_amd as m
    with pytest.raises(OSError):
        m.LzfseRingDecoder(context=ctx).decode(PieceReader(oracle.encode(snappy_raw["html"]), [4096]), Full())


def test_reader_pulls_the_decoded_bytes(ctx, oracle, golden_dir, snappy_raw):
    """LzfseRingDecoder::reader / reader_bytes (decode/ring_decoder.rs:75-90): read(n) gives n bytes until the stream ends,
    b"" afterwards; errors come out of the read that reaches them; after PayloadOverflow the reader is in State::Err
    (decode/reader_core.rs:62-76, 160-168)."""
    import lzfse_rust_amd as m
    rng = np.random.default_rng(3)
    dec = m.LzfseRingDecoder(context=ctx, window=200000, read_size=5000)
    for name in ("alice29.txt", "html_x_4", "fireworks.jpeg"):
        enc = open(os.path.join(golden_dir, "snappy", name + ".lzfse"), "rb").read()
        want = oracle.decode(enc)
        r = dec.reader(io.BytesIO(enc))
        got = bytearray()
        while True:
            n = int(rng.integers(1, 70000))
            piece = r.read(n)
            assert len(piece) == n or len(got) + len(piece) == len(want)
            if not piece:
                break
            got += piece
        assert bytes(got) == want and r.read(10) == b""
        assert isinstance(r.into_inner(), io.BytesIO)
    raw = snappy_raw["lcet10.txt"] * 9
    assert dec.reader_bytes(oracle.encode(raw)).read() == raw
    buf = bytearray(1000)
    assert dec.reader_bytes(oracle.encode(raw)).readinto(buf) == 1000 and bytes(buf) == raw[:1000]
    enc = oracle.encode(snappy_raw["html"])
    r = dec.reader_bytes(enc + b"\0")
    with pytest.raises(m.LzfseError) as e:
        r.read()
    assert e.value.status == 7
    with pytest.raises(m.LzfseError) as e:
        r.read(1)
    assert e.value.status == 5
    bad = oracle.encode(raw)[:-3000]
    r = dec.reader_bytes(bad)
    assert r.read(1 << 20) == raw[: 1 << 20]              # the stream is cut 3 MB further on
    with pytest.raises(m.LzfseError) as e:
        r.read()
    assert e.value.status == oracle.decode_status(bytes(bad), 1 << 23) != 0


def test_cut_stream_same_sink_whatever_the_feeds(ctx, oracle):
    """A multi-block stream that ends inside a block: the sound blocks in front of the cut reach the sink, and the
    totals say so, whether the stream is fed whole, in pieces or byte by byte (the reference decodes block by block,
    decode/decoder.rs:76-99: it has written them when the error comes)."""
    import lzfse_rust_amd as m
    rng = np.random.default_rng(41)
    words = [bytes(rng.integers(97, 123, size=int(k), dtype=np.uint8)) for k in rng.integers(2, 9, size=300)]
    raw = b" ".join(words[int(i)] for i in rng.integers(0, 300, size=90_000))[:400_000]
    enc = oracle.encode(raw)
    # block boundaries from the headers (bvx2: header size | payload sizes, fse/block.rs:108-136)
    cuts, pos = [], 0
    while enc[pos:pos + 4] == b"bvx2":
        p1, p2, p3 = (int.from_bytes(enc[pos + 8 + 8 * k:pos + 16 + 8 * k], "little") for k in range(3))
        cuts.append(pos)
        pos += (p3 & 0xFFFFFFFF) + ((p1 >> 20) & 0xFFFFF) + ((p2 >> 40) & 0xFFFFF)
    assert len(cuts) >= 4 and enc[pos:pos + 4] == b"bvx$"
    cut = cuts[3] + 100          # inside the fourth block
    damaged = enc[:cut]
    whole_blocks_raw = sum(int.from_bytes(enc[c + 4:c + 8], "little") for c in cuts[:3])
    results = []
    for sizes in ([len(damaged)], [4096], [1], [cuts[1] + 5, 7, 100_000]):
        out = io.BytesIO()
        dec = m.LzfseRingDecoder(context=ctx)
        with pytest.raises(m.LzfseError) as e:
            dec.decode(PieceReader(damaged, sizes), out)
        results.append((e.value.status, out.getvalue()))
    assert all(r == results[0] for r in results)
    assert results[0][0] == oracle.decode_status(damaged, len(raw)) == 8   # PayloadUnderflow
    assert results[0][1] == raw[:whole_blocks_raw]


def test_bvxn_header_that_promises_gigabytes(ctx, oracle):
    """A bvxn block whose header says 4 GiB over a payload of a few bytes: the reference grows a Vec as it produces and
    fails in the LZVN decoder at once; nothing here may size a buffer by the header (136 output bytes per payload byte
    is what a sound block can yield)."""
    import resource
    import lzfse_rust_amd as m
    payload = bytes([0xE3, 1, 2, 3, 0x06, 0, 0, 0, 0, 0, 0, 0])            # SmlL 3 literals, then end of stream
    good = b"bvxn" + (3).to_bytes(4, "little") + len(payload).to_bytes(4, "little") + payload + b"bvx$"
    assert oracle.decode(good) == bytes([1, 2, 3])
    liar = b"bvxn" + (0xFFFFFFFF).to_bytes(4, "little") + len(payload).to_bytes(4, "little") + payload + b"bvx$"
    liars = liar[:-4] * 40 + b"bvx$"
    for stream in (liar, liars):
        assert m.decode_size(stream, partial=True) <= 136 * len(stream)
        want = oracle.decode_status(stream, 1 << 16)
        assert want != 0
        before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
        with pytest.raises(m.LzfseError) as e:
            m.LzfseDecoder(context=ctx).decode_bytes(stream, bytearray())
        assert e.value.status == want
        for sizes in ([len(stream)], [5]):
            with pytest.raises(m.LzfseError) as e:
                m.LzfseRingDecoder(context=ctx).decode(PieceReader(stream, sizes), io.BytesIO())
            assert e.value.status == want
        assert resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - before < (256 << 10)   # KiB: no gigabyte was allocated


def test_fuzz_read(ctx, oracle):
    """test/src/fuzz_read.rs:8-33 at the reference's size: 8 MiB of Seq encoded by the slice encoder, read back through
    LzfseRingDecoder::reader in random-length reads of (gen % 0x20) * multiplier bytes until a read comes back short.
    Multipliers 0x100 / 0x1000 / 0x10000 with the reference's seeds 0 .. (a few each: its 0x100 seeds per multiplier only
    vary the read lengths; 1 and 0x10 are half a million Python calls a seed), plus one seed of 0x10."""
    import lzfse_rust_amd as m
    import test_kit as tk
    data = tk.seq(0x0080_0000)
    enc = bytearray()
    m.LzfseEncoder(context=ctx).encode_bytes(data, enc)
    assert bytes(enc) == oracle.encode(data)
    dec = m.LzfseRingDecoder(context=ctx)
    for mult, seeds in ((0x10, 1), (0x100, 2), (0x1000, 4), (0x10000, 6)):
        for seed in range(seeds):
            rdr = dec.reader(io.BytesIO(bytes(enc)))
            rng = tk.Rng(seed)
            got = bytearray()
            while True:
                n = (rng.gen() % 0x20) * mult
                piece = rdr.read(n)
                got += piece
                if len(piece) != n:
                    break
            rdr.into_inner()
            assert bytes(got) == data, (mult, seed)


def test_stream_objects_may_outlive_their_context(oracle, snappy_raw):
    """include/lzfse_mi.h: a stream object and its context may be destroyed in either order. Destroying the context first
    detaches the stream objects still alive: their calls fail with BAD_ARGUMENT, their destroy frees only their own buffers
    (round 3 handed them to the freed context)."""
    import lzfse_rust_amd as m
    from lzfse_rust_amd import _native
    raw = snappy_raw["alice29.txt"]
    enc = oracle.encode(raw)
    c = m.Context(0)
    lib = c._lib
    d, e = C.c_void_p(), C.c_void_p()
    assert lib.lzfse_mi_dstream_create(c._h, 1 << 20, C.byref(d)) == 0
    assert lib.lzfse_mi_estream_create(c._h, 1 << 20, C.byref(e)) == 0
    got = []
    cb = _native.WRITE_FN(lambda _u, p, n: got.append(bytes((C.c_uint8 * n).from_address(C.addressof(p.contents)))) and 0)
    a = np.frombuffer(enc, dtype=np.uint8)
    half = a.size // 2
    assert lib.lzfse_mi_dstream_feed(d, a.ctypes.data, half, 0, cb, None) == 0     # (grows the decoder's window buffers)
    r = np.frombuffer(raw, dtype=np.uint8)
    assert lib.lzfse_mi_estream_feed(e, r.ctypes.data, r.size, cb, None) == 0
    c.close()                                                                      # the context goes first
    assert lib.lzfse_mi_dstream_feed(d, a[half:].ctypes.data, a.size - half, 1, cb, None) == 11   # LZFSE_MI_BAD_ARGUMENT
    u, v = C.c_uint64(0), C.c_uint64(0)
    assert lib.lzfse_mi_estream_finish(e, cb, None, C.byref(u), C.byref(v)) == 11
    lib.lzfse_mi_dstream_destroy(d)
    lib.lzfse_mi_estream_destroy(e)
    # and the usual order still hands the buffers on: a second object on a live context works as the first did
    c2 = m.Context(0)
    for _ in range(2):
        out = io.BytesIO()
        m.LzfseRingDecoder(context=c2, window=1 << 20).decode(PieceReader(enc, [70000]), out)
        assert out.getvalue() == raw
    c2.set_option("stream_spare", 0)   # frees what the finished objects left, keeps nothing from now on
    out = io.BytesIO()
    m.LzfseRingDecoder(context=c2, window=1 << 20).decode(PieceReader(enc, [70000]), out)
    assert out.getvalue() == raw
    c2.close()


def test_writer_pieces_are_views_released_after_the_call(ctx, oracle, snappy_raw):
    """LzfseRingDecoder hands the writer views of the window buffer that die with the call: a sink that keeps one fails LOUDLY
    later (ValueError) instead of reading overwritten memory; zero_copy=False hands out bytes it may keep."""
    import lzfse_rust_amd as m
    raw = snappy_raw["html"]
    enc = oracle.encode(raw)

    class Keep:
        def __init__(self):
            self.pieces = []

        def write(self, b):
            self.pieces.append(b)

    k = Keep()
    m.LzfseRingDecoder(context=ctx, window=1 << 16).decode(PieceReader(enc, [9000]), k)
    assert k.pieces and all(isinstance(p, memoryview) for p in k.pieces)
    with pytest.raises(ValueError):
        bytes(k.pieces[0])                      # released: no silent read of a reused buffer
    k = Keep()
    m.LzfseRingDecoder(context=ctx, window=1 << 16, zero_copy=False).decode(PieceReader(enc, [9000]), k)
    assert all(isinstance(p, bytes) for p in k.pieces) and b"".join(k.pieces) == raw
    out = io.BytesIO()
    m.LzfseRingDecoder(context=ctx, window=1 << 16).decode(PieceReader(enc, [9000]), out)
    assert out.getvalue() == raw


def test_encoder_pieces_and_typed_sources(ctx, oracle, snappy_raw):
    """LzfseWriter hands its sink views that die with the call (as the decoder does); zero_copy=False hands out bytes a sink may
    keep (`pieces.append`). encode_bytes sizes its destination by the BYTES of a typed source, not by its element count."""
    import lzfse_rust_amd as m
    raw = snappy_raw["alice29.txt"] * 9
    want = oracle.ring_encode(raw)
    class Keep:
        def __init__(self):
            self.pieces = []

        def write(self, b):
            self.pieces.append(b)

    k = Keep()
    pieces = k.pieces
    w = m.LzfseRingEncoder(context=ctx, window=1 << 20, zero_copy=False).writer(k)
    for o in range(0, len(raw), 300000):
        w.write(raw[o:o + 300000])
    w.finalize()
    assert all(isinstance(p, bytes) for p in pieces) and b"".join(pieces) == want
    k = Keep()
    kept = k.pieces
    w = m.LzfseRingEncoder(context=ctx, window=1 << 20).writer(k)
    w.write(raw)
    w.finalize()
    assert kept and all(isinstance(p, memoryview) for p in kept)
    with pytest.raises(ValueError):
        bytes(kept[0])
    typed = np.frombuffer(snappy_raw["html"][:102400], dtype=np.uint32)          # 25 600 elements, 102 400 bytes
    out = bytearray(b"head")
    n = m.LzfseEncoder(context=ctx).encode_bytes(typed, out)
    assert bytes(out[4:]) == oracle.encode(snappy_raw["html"][:102400]) and n == len(out) - 4


def test_windows_in_the_background(oracle, snappy_raw):
    """Round 4: a stream object hands a full window to a helper thread and returns (stream.hip). Two stream objects of one context
    with windows in flight at the same time (a decoder whose sink is an encoder: transcoding, window sizes that do not line
    up), objects dropped and contexts destroyed while a window is in flight, and an error that a background window met: it
    arrives at a later call, after the windows before it have been written."""
    import lzfse_rust_amd as m
    from lzfse_rust_amd import _native
    raw = (snappy_raw["lcet10.txt"] + snappy_raw["plrabn12.txt"] + snappy_raw["alice29.txt"]) * 9      # 9.5 MB
    enc = oracle.encode(raw)
    c = m.Context(0)
    out = bytearray()
    w = m.LzfseRingEncoder(context=c, window=3 << 20).writer_bytes(out)
    u, v = m.LzfseRingDecoder(context=c, window=1 << 20).decode(PieceReader(enc, [300000, 17, 1 << 20]), w)
    w.finalize()
    assert (u, v) == (len(enc), len(raw)) and bytes(out) == oracle.ring_encode(raw)
    # dropped with a window in flight (no finalize): the object waits for its helper and goes
    w2 = m.LzfseRingEncoder(context=c, window=1 << 20).writer_bytes(bytearray())
    w2.write(raw[:4 << 20])
    del w2
    # the context destroyed with windows in flight: they finish (or are refused), later calls say BAD_ARGUMENT
    lib = c._lib
    d, e = C.c_void_p(), C.c_void_p()
    assert lib.lzfse_mi_dstream_create(c._h, 1 << 20, C.byref(d)) == 0
    assert lib.lzfse_mi_estream_create(c._h, 1 << 20, C.byref(e)) == 0
    cb = _native.WRITE_FN(lambda _u, p, n: 0)
    a = np.frombuffer(enc, dtype=np.uint8)
    r = np.frombuffer(raw, dtype=np.uint8)
    assert lib.lzfse_mi_dstream_feed(d, a.ctypes.data, a.size * 2 // 3, 0, cb, None) == 0
    assert lib.lzfse_mi_estream_feed(e, r.ctypes.data, 5 << 20, cb, None) == 0
    c.close()
    assert lib.lzfse_mi_dstream_feed(d, a.ctypes.data, 16, 1, cb, None) == 11
    assert lib.lzfse_mi_estream_feed(e, r.ctypes.data, 16, cb, None) == 11
    lib.lzfse_mi_dstream_destroy(d)
    lib.lzfse_mi_estream_destroy(e)
    # a damaged block far into the stream: the windows before it reach the sink, then the slice path's status
    c = m.Context(0)
    got = bytearray()

    class Keep:
        def write(self, b):
            got.extend(bytes(b))

    cut = bytearray(enc)
    at = len(enc) * 2 // 3
    cut[at:at + 4] = b"bvxQ"        # (four bytes of some block's payload)
    with pytest.raises(m.LzfseError) as ei:
        m.LzfseRingDecoder(context=c, window=1 << 20).decode(PieceReader(bytes(cut), [1 << 20]), Keep())
    assert ei.value.status == oracle.decode_status(bytes(cut), len(raw) + 64)
    assert len(got) >= 4 << 20 and raw.startswith(bytes(got))
    c.close()


def test_decode_reads_into_the_library_buffer(ctx, oracle, snappy_raw):
    """lzfse_mi_dstream_reserve / _commit (decode(reader, writer) with a reader that has readinto): the same bytes and the same
    statuses as feed, whatever the reads bring; a commit of more than was reserved is refused."""
    import lzfse_rust_amd as m
    raw = (snappy_raw["plrabn12.txt"] + snappy_raw["geo.protodata"]) * 6       # 3.6 MB
    enc = oracle.encode(raw)

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

    for sizes, read_size in (([1 << 20], 1 << 20), ([7, 100000, 1, 33333], 65536), ([1 << 24], 1 << 22)):
        out = io.BytesIO()
        u, v = m.LzfseRingDecoder(context=ctx, window=1 << 20, read_size=read_size).decode(Choppy(enc, sizes), out)
        assert (u, v) == (len(enc), len(raw)) and out.getvalue() == raw, sizes
    cut = enc[:len(enc) // 2]
    with pytest.raises(m.LzfseError) as ei:
        m.LzfseRingDecoder(context=ctx, window=1 << 20).decode(io.BytesIO(cut), io.BytesIO())
    assert ei.value.status == oracle.decode_status(cut, len(raw) + 64)
    lib = ctx._lib
    h, p = C.c_void_p(), C.c_void_p()
    assert lib.lzfse_mi_dstream_create(ctx._h, 0, C.byref(h)) == 0
    assert lib.lzfse_mi_dstream_reserve(h, 100, C.byref(p)) == 0 and p.value
    from lzfse_rust_amd import _native
    cb = _native.WRITE_FN(lambda _u, q, n: 0)
    assert lib.lzfse_mi_dstream_commit(h, 101, 0, cb, None) == 11      # LZFSE_MI_BAD_ARGUMENT
    lib.lzfse_mi_dstream_destroy(h)
