"""Streaming decode (lzfse_mi_dstream_*) and streaming encode (lzfse_mi_estream_*) against the one-call forms on the same
stream: host-pointer rates, PCIe included.
    python scripts/stream_bench.py [MB]"""
import hashlib
import numpy as np
import io
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))

import lzfse_rust_amd as m
from bench import synth_text


class Sink:
    def __init__(self):
        self.n = 0

    def write(self, b):
        self.n += len(b)


class HashSink:
    def __init__(self):
        self.h, self.n = hashlib.sha256(), 0

    def write(self, b):
        self.h.update(b)
        self.n += len(b)


def main():
    mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    ctx = m.Context(0)
    raw = bytes(synth_text(mb << 20))
    encs, st = ctx.encode_batch([raw])
    assert st[0] == 0
    enc = encs[0].tobytes()
    for _ in range(2):
        t = time.perf_counter()
        outs, st = ctx.decode_batch([enc])
        dt = time.perf_counter() - t
    print(f"slice decode (host pointers): {len(raw) / dt / 1e6:9.1f} MB/s")
    for window in (1 << 20, 4 << 20, 16 << 20, 64 << 20):
        for _ in range(2):
            s = Sink()
            t = time.perf_counter()
            u, v = m.LzfseRingDecoder(context=ctx, window=window, read_size=1 << 20).decode(io.BytesIO(enc), s)
            dt = time.perf_counter() - t
        assert (u, v, s.n) == (len(enc), len(raw), len(raw))
        print(f"stream decode, window {window >> 20:3d} MiB: {len(raw) / dt / 1e6:9.1f} MB/s")
    for _ in range(2):
        s = Sink()
        t = time.perf_counter()
        u, v = m.LzfseRingDecoder(context=ctx, window=64 << 20, read_size=1 << 20, zero_copy=False).decode(io.BytesIO(enc), s)
        dt = time.perf_counter() - t
    print(f"stream decode, window  64 MiB: {len(raw) / dt / 1e6:9.1f} MB/s   (zero_copy=False: the writer is handed bytes it may keep -- one more copy into a fresh allocation per window)")
    # ---- encode: the ring / stream encoder's bytes, whole input in one call against windows ----
    for _ in range(2):
        t = time.perf_counter()
        outs, st = ctx.encode_batch([raw], ring=True)
        dt = time.perf_counter() - t
    want = outs[0].tobytes()
    print(f"ring encode, one call (host pointers): {len(raw) / dt / 1e6:9.1f} MB/s")
    for window in (4 << 20, 16 << 20, 64 << 20):
        s = HashSink()   # once for the bytes ...
        u, v = m.LzfseRingEncoder(context=ctx, window=window, read_size=1 << 20).encode(io.BytesIO(raw), s)
        assert (u, v) == (len(raw), len(want)) and s.h.digest() == hashlib.sha256(want).digest()
        for _ in range(2):   # ... and with a sink that only counts for the rate
            s = Sink()
            t = time.perf_counter()
            u, v = m.LzfseRingEncoder(context=ctx, window=window, read_size=1 << 20).encode(io.BytesIO(raw), s)
            dt = time.perf_counter() - t
        assert (u, v, s.n) == (len(raw), len(want), len(want))
        print(f"stream encode, window {window >> 20:3d} MiB: {len(raw) / dt / 1e6:9.1f} MB/s")

    # ---- the C entry points by themselves: no Python reader (pieces are views of one array), a sink that only counts ----
    import ctypes as C
    from lzfse_rust_amd import _native
    lib = ctx._lib
    a = np.frombuffer(raw, dtype=np.uint8)
    for window in (16 << 20, 64 << 20):
        for _ in range(2):
            h = C.c_void_p()
            assert lib.lzfse_mi_estream_create(ctx._h, window, C.byref(h)) == 0
            got = [0]
            cb = _native.WRITE_FN(lambda u, p, n: (got.__setitem__(0, got[0] + n), 0)[1])
            t = time.perf_counter()
            for o in range(0, a.size, 1 << 20):
                piece = a[o:o + (1 << 20)]
                assert lib.lzfse_mi_estream_feed(h, piece.ctypes.data, piece.size, cb, None) == 0
            u, v = C.c_uint64(0), C.c_uint64(0)
            assert lib.lzfse_mi_estream_finish(h, cb, None, C.byref(u), C.byref(v)) == 0
            dt = time.perf_counter() - t
            lib.lzfse_mi_estream_destroy(h)
        assert got[0] == len(want)
        print(f"stream encode, window {window >> 20:3d} MiB, lzfse_mi_estream_* called directly (1 MiB pieces, counting sink): {len(raw) / dt / 1e6:9.1f} MB/s")


if __name__ == "__main__":
    main()
