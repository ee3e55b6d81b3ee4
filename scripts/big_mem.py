"""The reference's big-memory tests of the slice encoder (test/src/big_mem.rs:2-110: noise and zeros of 512 MiB, of sizes around 2^31
and of 8 GiB) on the device: encode_bytes == the oracle's bytes (SHA-256), decode_bytes gives the input back. Sizes beyond
0x8000_0002 are matched in several blocks by the reference's front end (`reposition`, frontend_bytes.rs:348-375) and by
lzfse_mi_encode (round 5: encode_slice_blocks, api.hip), a block per device call.
    python scripts/big_mem.py [quick | huge]          (profiles/r05_big_mem.txt; huge adds 0x2_0000_0000)"""
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
    if n > 0x8000_0002:
        out = bytearray()
        m.LzfseEncoder(context=ctx).encode_bytes(data, out)      # lzfse_mi_encode: the slice is matched in blocks
        enc = np.frombuffer(out, dtype=np.uint8)
    else:
        outs, st = ctx.encode_batch([data])
        assert st[0] == 0, st
        enc = outs[0]
    t_enc = time.time() - t
    t = time.time()
    want = O.encode(data)
    t_or = time.time() - t
    assert hashlib.sha256(enc).digest() == hashlib.sha256(want).digest(), f"{name} {n:#x}: encode differs"
    del want
    t = time.time()
    if n > 0xFFFF_FFFF:
        # (beyond 2^32 bytes: through the stream decoder, a window at a time)
        import io

        class H:
            h, n = hashlib.sha256(), 0

            def write(self, b):
                H.h.update(b); H.n += len(b)

        m.LzfseRingDecoder(context=ctx).decode(io.BytesIO(enc), H())
        t_dec = time.time() - t
        assert H.n == n and H.h.digest() == hashlib.sha256(data).digest()
    else:
        dec, st = ctx.decode_batch([enc], caps=[n])
        t_dec = time.time() - t
        assert st[0] == 0 and dec[0].size == n and hashlib.sha256(dec[0]).digest() == hashlib.sha256(data).digest()
    print(f"{name} {n:#x}: {enc.size} bytes == oracle, round trip ok; device encode {t_enc:.1f} s, decode {t_dec:.1f} s, oracle encode {t_or:.0f} s", flush=True)


mode = sys.argv[1] if len(sys.argv) > 1 else ""
sizes = [0x2000_0000] if mode == "quick" else [0x2000_0000, 0x7FFF_FFFD, 0x7FFF_FFFF, 0x8000_0000, 0x8000_0002, 0x8000_0003, 0x8000_0004]
if mode == "huge":
    sizes = [0x8000_0003, 0x2_0000_0000]
big = rng_bytes(max(sizes))
for n in sizes:
    one("rng", big[:n])
del big
for n in sizes:
    one("zeros", np.zeros(n, dtype=np.uint8))
print("big_mem ok")
