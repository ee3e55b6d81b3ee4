"""The reference's large-stream tests where the driver runs them (test/src/big_mem.rs:32-102, test/src/huge.rs:12-21): zeros of
0x7FFF_FFFF, 0x8000_0001 and 0x8000_0002 bytes through lzfse_mi_encode / lzfse_mi_decode against the oracle, a stream of more than
2^32 zeros through lzfse_mi_estream_* against the restated ring encoder, and a 1 GiB `cli -encode --plain | cli -decode` pipe. Zeros
cost the oracle next to nothing and the device about a second; the noise cases of big_mem.rs (20 s of oracle time per GiB) stay in
scripts/big_mem.py. Every test skips when the box is short of host memory."""
import ctypes as C
import hashlib
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _need_gib(g):
    have = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            have = min(have, int(lim))
    except OSError:
        pass
    if have < (g << 30):
        pytest.skip(f"needs ~{g} GiB of host memory")


@pytest.fixture(scope="module")
def ctx():
    import lzfse_rust_amd as m
    return m.Context(0)


@pytest.mark.parametrize("n", [0x7FFF_FFFF, 0x8000_0001, 0x8000_0002])
def test_big_zeros_slice(ctx, oracle, n):
    """big_mem.rs:84-102 (zeros): encode_bytes == the oracle's bytes, decode_bytes gives the zeros back. 0x8000_0001 / _0002 are
    still ONE block of the reference's front end (BLOCK_GUIDE + 3, frontend_bytes.rs:169-178)."""
    _need_gib(12)
    data = np.zeros(n, dtype=np.uint8)
    want = oracle.encode(data)
    outs, st = ctx.encode_batch([data])
    assert st[0] == 0, st
    assert outs[0].size == len(want) and outs[0].tobytes() == want
    del outs
    dec, st = ctx.decode_batch([want], caps=[n])
    assert st[0] == 0 and dec[0].size == n and not dec[0].any()


@pytest.mark.parametrize("n", [0x8000_0003, 0x8000_0004])
def test_big_zeros_slice_repositions(ctx, oracle, n):
    """big_mem.rs:92-102 at the first sizes the reference's front end matches in TWO blocks (more than BLOCK_GUIDE + 3 bytes:
    frontend_bytes.rs:348-375 reposition): LzfseEncoder::encode_bytes == the oracle's bytes (its restated reposition), and back.
    tests/test_gpu_reposition.py has the many-block cases at a small BLOCK_GUIDE; noise of these sizes: scripts/big_mem.py."""
    import lzfse_rust_amd as m
    _need_gib(16)
    data = np.zeros(n, dtype=np.uint8)
    want = oracle.encode(data)
    out = bytearray()
    got = m.LzfseEncoder(context=ctx).encode_bytes(data, out)
    assert got == len(want) and bytes(out) == want
    dec, st = ctx.decode_batch([want], caps=[n])
    assert st[0] == 0 and dec[0].size == n and not dec[0].any()


def test_big_zeros_stream_beyond_u32(ctx, oracle):
    """More than 2^32 zeros through LzfseWriter (lzfse_mi_estream_*: 64 MiB windows with carried state; a window's positions are
    relative) == the restated ring encoder (frontend_ring.rs), which is fed the same pieces."""
    import lzfse_rust_amd as m
    _need_gib(8)
    piece = np.zeros(64 << 20, dtype=np.uint8)
    n_pieces, tail = 65, 12345            # 4.06 GiB + a ragged end
    h = oracle.lib.lzo_ring_new(0, 0, 0, None)
    try:
        for _ in range(n_pieces):
            assert oracle.lib.lzo_ring_write(h, piece.ctypes.data, piece.size) == 0
        assert oracle.lib.lzo_ring_write(h, piece.ctypes.data, tail) == 0
        ptr, n = C.c_void_p(), C.c_size_t(0)
        assert oracle.lib.lzo_ring_finish(h, C.byref(ptr), C.byref(n)) == 0
        want = C.string_at(ptr, n.value)
    finally:
        oracle.lib.lzo_ring_free(h)
    got = bytearray()
    w = m.LzfseRingEncoder(context=ctx).writer_bytes(got)
    for _ in range(n_pieces):
        w.write(piece)
    w.write(piece[:tail])
    w.finalize()
    assert len(got) == len(want) and bytes(got) == want
    # and back through the stream decoder, a hashing sink
    total = n_pieces * piece.size + tail

    class Count:
        n, bad = 0, False

        def write(self, b):
            a = np.frombuffer(b, dtype=np.uint8)
            Count.bad |= bool(a.any())
            Count.n += a.size

    import io
    m.LzfseRingDecoder(context=ctx).decode(io.BytesIO(want), Count())
    assert Count.n == total and not Count.bad


def test_pipe_one_gib_through_the_cli():
    """huge.rs:12-21 at 1 GiB: masked noise > cli -encode --plain > cli -decode > check, two processes streaming a window at a time."""
    _need_gib(8)
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import stream_pipe as sp
    total = 1 << 30
    env = dict(os.environ, PYTHONPATH=ROOT)
    cli = [sys.executable, "-m", "lzfse_rust_amd.cli"]
    enc = subprocess.Popen(cli + ["-encode", "--plain"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)
    dec = subprocess.Popen(cli + ["-decode"], stdin=enc.stdout, stdout=subprocess.PIPE, env=env)
    enc.stdout.close()

    def feed():
        g = sp.Seq()
        try:
            for _ in range(total // sp.CHUNK):
                enc.stdin.write(g.piece().tobytes())
        finally:
            enc.stdin.close()

    th = threading.Thread(target=feed)
    th.start()
    g, got = sp.Seq(), 0
    want = hashlib.sha256()
    have = hashlib.sha256()
    try:
        while True:
            b = dec.stdout.read(4 << 20)
            if not b:
                break
            have.update(b)
            got += len(b)
    finally:
        th.join()
    for _ in range(total // sp.CHUNK):
        want.update(g.piece().tobytes())
    assert enc.wait(timeout=60) == 0 and dec.wait(timeout=60) == 0
    assert got == total and have.digest() == want.digest()
