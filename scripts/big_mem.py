"""The reference's big-memory tests of the slice encoder (test/src/big_mem.rs:2-110: noise and zeros of 512 MiB and of sizes around
2^31) on the device: encode_bytes == the oracle's bytes (SHA-256), decode_bytes gives the input back; sizes from 2^31 on are
refused by the slice calls (`reposition`, frontend_bytes.rs:348-375, is not built) -- the stream encoder takes those
(scripts/stream_big.py).
    python scripts/big_mem.py [quick]          (profiles/r03_big_mem.txt)"""
import hashlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.dirname(__file__))
import lzfse_rust_amd as m
import oracle_py
import stream_pipe

O = oracle_py.Oracle()
ctx = m.Context(0)


def rng_bytes(n):
    """Rng::default().gen_vec(n) (test_kit): the LCG's states as little-endian words"""
    stream_pipe.MASK = np.uint32(0xFFFFFFFF)
    g = stream_pipe.Seq()
    out = np.empty((n + stream_pipe.CHUNK - 1) // stream_pipe.CHUNK * stream_pipe.CHUNK, dtype=np.uint8)
    for o in range(0, out.size, stream_pipe.CHUNK):
        out[o:o + stream_pipe.CHUNK] = g.piece()
    return out[:n]


def one(name, data):
    n = data.size
    t = time.time()
    outs, st = ctx.encode_batch([data])
    t_enc = time.time() - t
    if n > 0x7FFFFFFF:
        assert st[0] == 9, st       # LZFSE_MI_UNSUPPORTED
        print(f"{name} {n:#x}: refused (LZFSE_MI_UNSUPPORTED), as documented", flush=True)
        return
    assert st[0] == 0, st
    enc = outs[0]
    t = time.time()
    want = O.encode(data)
    t_or = time.time() - t
    assert hashlib.sha256(enc).digest() == hashlib.sha256(want).digest(), f"{name} {n:#x}: encode differs"
    del want
    t = time.time()
    dec, st = ctx.decode_batch([enc], caps=[n])
    t_dec = time.time() - t
    assert st[0] == 0 and dec[0].size == n and hashlib.sha256(dec[0]).digest() == hashlib.sha256(data).digest()
    print(f"{name} {n:#x}: {enc.size} bytes == oracle, round trip ok; device encode {t_enc:.1f} s, decode {t_dec:.1f} s, oracle encode {t_or:.0f} s", flush=True)


sizes = [0x2000_0000] if len(sys.argv) > 1 else [0x2000_0000, 0x7FFF_FFFD, 0x7FFF_FFFF, 0x8000_0000]
big = rng_bytes(max(sizes))
for n in sizes:
    one("rng", big[:n])
del big
for n in sizes:
    one("zeros", np.zeros(n, dtype=np.uint8))
print("big_mem ok")
