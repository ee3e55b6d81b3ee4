"""A stream of more than 2^31 (or 2^32) bytes through the windowed stream encoder against the restated ring encoder (SHA-256 of
both streams): the slice calls refuse such inputs (positions are 31 bits on the device), a window's positions are relative.
    python scripts/stream_big.py [GiB]        (about 15 s of oracle time per GiB; profiles/r03_stream_big.txt)"""
import hashlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import lzfse_rust_amd as m
import oracle_py
from bench import synth_text

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.2
kind = sys.argv[2] if len(sys.argv) > 2 else "text"      # text | mixed (text, low-entropy noise, noise, zeros, runs in turn)
base = np.frombuffer(synth_text(64 << 20, seed=3), dtype=np.uint8)
copies = int(gib * 16) + 1
raw = np.empty(copies * base.size, dtype=np.uint8)
rng = np.random.default_rng(5)
for c in range(copies):
    a = raw[c * base.size:(c + 1) * base.size]
    a[:] = base
    a[c % 251::251] ^= np.uint8(1 + c % 200)
    if kind == "mixed":
        k = c % 5
        if k == 1:
            a[:] = rng.integers(0, 4, size=a.size, dtype=np.uint8) * 85
        elif k == 2:
            a[: a.size // 2] = rng.integers(0, 256, size=a.size // 2, dtype=np.uint8)
        elif k == 3:
            a[a.size // 4: a.size // 4 * 3] = 0
        elif k == 4:
            r = np.repeat(rng.integers(0, 256, size=a.size // 600, dtype=np.uint8), rng.integers(3, 1200, size=a.size // 600))
            a[: min(a.size, r.size)] = r[: a.size]
print(f"{raw.size} bytes", flush=True)
O = oracle_py.Oracle()
t = time.time()
want = hashlib.sha256()
mv = memoryview(raw)
# the oracle through its handle API, 16 MiB per write; its output is hashed and dropped
h = O.lib.lzo_ring_new(0, 0, 0, None)
import ctypes as C
for o in range(0, raw.size, 16 << 20):
    a = raw[o:o + (16 << 20)]
    assert O.lib.lzo_ring_write(h, a.ctypes.data, a.size) == 0
ptr, n = C.c_void_p(), C.c_size_t(0)
assert O.lib.lzo_ring_finish(h, C.byref(ptr), C.byref(n)) == 0
want.update(C.string_at(ptr, n.value)); want_len = n.value
O.lib.lzo_ring_free(h)
print(f"oracle: {want_len} bytes, {time.time() - t:.0f} s", flush=True)


class Sink:
    def __init__(self, keep=False):
        self.h, self.n, self.parts = hashlib.sha256(), 0, [] if keep else None

    def write(self, b):
        self.h.update(b); self.n += len(b)
        if self.parts is not None:
            self.parts.append(bytes(b))


ctx = m.Context(0)
t = time.time()
s = Sink(keep=True)
w = m.LzfseRingEncoder(context=ctx).writer(s)
for o in range(0, raw.size, 8 << 20):
    w.write(mv[o:o + (8 << 20)])
w.finalize()
print(f"device: {s.n} bytes, {time.time() - t:.1f} s", flush=True)
assert s.n == want_len and s.h.digest() == want.digest()
print("equal")
# ... and back through the stream decoder
import io
enc = b"".join(s.parts)
t = time.time()
d = Sink()
u, v = m.LzfseRingDecoder(context=ctx, read_size=8 << 20).decode(io.BytesIO(enc), d)
print(f"stream decode: {v} bytes, {time.time() - t:.1f} s", flush=True)
assert (u, v) == (len(enc), raw.size) and d.h.digest() == hashlib.sha256(mv).digest()
print("round trip equal")
