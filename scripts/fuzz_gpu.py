"""One-off randomised parity campaign on a GPU box (not part of the test-suite: the suite's cases are fixed):
    python scripts/fuzz_gpu.py [ROUNDS] [SEED] [walk|pipe|pipeck|big|ring|stream|chain]  (stream: 3 to 6 inputs of 1.2 .. 12 MiB per round through LzfseWriter with
                                                              windows of 1 .. 3 MiB and pieces of any size: the bytes are those of the restated ring
                                                              encoder on the whole input, and the windows did leave early;
                                                              ring: the streams are ALSO encoded with the ring / stream encoder's parse and
                                                              compared with the restated frontend_ring.rs; lengths reach across the 512 KiB ring;
                                                              walk: the diagnostic build, every stream through the parallel header walk first;
                                                              pipe: every stream of the tile kernel through the pipelined LZ kernel, K and tile size changing per round;
                                                              pipeck: the same in the diagnostic build, where every ticket's bytes are checksummed by their writer and
                                                              checked by the next ticket as it reads them (no hand-over may ever be refused);
                                                              big: 2 to 9 streams of 2 .. 24 MiB per round: pointer jumping, the parallel header walk and several
                                                              workgroups per stream, as the cost model mixes them)
Every round: ~120 streams of random structure and length (4 097 B .. 3 MiB, some at tile edges) are encoded by the device
and compared with the CPU restatement's bytes, decoded back, and a damaged copy of every stream (one byte changed, or cut
short) is decoded by both with the status codes and lengths compared."""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import lzfse_rust_amd as lz
import oracle_py

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
O = oracle_py.Oracle()
if len(sys.argv) > 3 and sys.argv[3] == "walk":
    ctx = lz.Context(0, diag=True)
    ctx.set_option("diag_walk", 1)
elif len(sys.argv) > 3 and sys.argv[3] in ("pipeck", "chain"):
    ctx = lz.Context(0, diag=True)
else:
    ctx = lz.Context(0)
PIPE = len(sys.argv) > 3 and sys.argv[3] in ("pipe", "pipeck")
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"
RING = len(sys.argv) > 3 and sys.argv[3] == "ring"
STREAM = len(sys.argv) > 3 and sys.argv[3] == "stream"
CHAIN = len(sys.argv) > 3 and sys.argv[3] == "chain"   # the diagnostic build, chain tiles of 1, 2 or 4 x 65 472 positions forced per round (a small call picks 1)
rng = np.random.default_rng(seed)
words = [bytes(rng.integers(97, 123, size=int(rng.integers(1, 12)), dtype=np.uint8)) for _ in range(800)]
TILE = 65472

def gen(kind, n):
    if kind == 0:
        return rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
    if kind == 1:
        return (rng.integers(0, int(rng.integers(2, 9)), size=n, dtype=np.uint8) * 31).astype(np.uint8).tobytes()
    if kind == 2:
        chunk = rng.integers(0, 256, size=int(rng.integers(5, 9000)), dtype=np.uint8)
        a = np.tile(chunk, n // chunk.size + 1)[:n].copy()
        idx = rng.integers(0, n, size=max(1, n // int(rng.integers(30, 8000))))
        a[idx] ^= rng.integers(1, 256, size=idx.size, dtype=np.uint8)
        return a.tobytes()
    if kind == 3:
        per = int(rng.integers(1, 300000))
        return (bytes(rng.integers(0, 256, size=per, dtype=np.uint8)) * (n // per + 1))[:n]
    if kind == 4:
        out = bytearray()
        while len(out) < n:
            out += words[int(rng.integers(0, 800))] + b" "
        return bytes(out[:n])
    if kind == 5:
        runs = bytearray()
        while len(runs) < n:
            runs += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 6000))
        return bytes(runs[:n])
    parts = bytearray()
    while len(parts) < n:
        parts += gen(int(rng.integers(0, 6)), int(rng.integers(100, 60000)))
    return bytes(parts[:n])

def length():
    r = rng.random()
    if r < 0.15:
        return int(TILE * rng.integers(1, 14 if CHAIN else 6) + 3 + rng.integers(-3, 70))
    if r < 0.3:
        return int(rng.integers(4097, 20000))
    if RING and r < 0.5:   # around the ring size and the round ends (multiples of 16 KiB beyond 512 KiB)
        return int(0x80000 + 0x4000 * rng.integers(-2, 40) + rng.integers(-40, 41))
    if r < 0.9:
        return int(rng.integers(20000, 600000))
    return int(rng.integers(600000, 3 << 20))

t0 = time.time(); total = 0


class Sink:
    def __init__(self):
        self.parts = []

    def write(self, b):
        self.parts.append(bytes(b))


def stream_round(rd):
    global total
    for _ in range(int(rng.integers(3, 7))):
        n = int(rng.integers(1_200_000, 12 << 20))
        kind = int(rng.integers(0, 8))
        raw = bytes(int(rng.integers(1, 4 << 20))) + gen(4, n // 2) if kind == 7 else gen(kind, n)
        window = int(rng.integers(1 << 20, 3 << 20))
        sink = Sink()
        w = lz.LzfseRingEncoder(context=ctx, window=window).writer(sink)
        o = 0
        while o < len(raw):
            k = int(rng.integers(1, 2_000_000)) if rng.random() < 0.8 else int(rng.integers(1, 5000))
            w.write(raw[o:o + k]); o += k
        early = sum(map(len, sink.parts))
        w.finalize()
        got = b"".join(sink.parts)
        want = O.ring_encode(raw)
        assert got == want, f"round {rd}: stream encode differs, {len(raw)} bytes of kind {kind}, window {window}"
        total += len(raw)
        print(f"  kind {kind} {len(raw)} B window {window}: {len(sink.parts)} pieces, {early} of {len(got)} B before finalize", flush=True)


for rd in range(rounds):
    if STREAM:
        stream_round(rd)
        print(f"round {rd}: ok, {time.time() - t0:.0f} s", flush=True)
        continue
    if CHAIN:
        ctx.set_option("diag_chain", int(rng.choice([0x10, 0x20, 0x40, 0x40, 0x41])))
    if PIPE:
        ctx.set_option("decode_pipe", int(rng.integers(2, 12)) | (int(rng.integers(0, 2)) << 8))
    if BIG:
        raws = [gen(int(rng.integers(0, 7)), int(rng.integers(2 << 20, 24 << 20))) for _ in range(int(rng.integers(2, 10)))]
    else:
        raws = [gen(int(rng.integers(0, 7)), max(4097, length())) for _ in range(120)]
    total += sum(map(len, raws))
    want = [O.encode(r) for r in raws]
    outs, st = ctx.encode_batch(raws)
    assert all(e == 0 for e in st), st
    for i, (r, o, w) in enumerate(zip(raws, outs, want)):
        assert o.tobytes() == w, f"round {rd}: encode differs, stream {i} of {len(r)} bytes"
    # a few of them again as a batch of their own: small batches of small streams are parsed in shorter segments (seg_for)
    pick = [int(x) for x in rng.choice(len(raws), size=min(6, len(raws)), replace=False)]
    souts, sst = ctx.encode_batch([raws[i] for i in pick], ring=bool(RING and rd & 1))
    for i, o in zip(pick, souts):
        assert o.tobytes() == (O.ring_encode(raws[i]) if (RING and rd & 1) else want[i]), f"round {rd}: small-batch encode differs, stream {i} of {len(raws[i])} bytes"
    if RING:
        rwant = [O.ring_encode(r) for r in raws]
        routs, rst = ctx.encode_batch(raws, ring=True)
        assert all(e == 0 for e in rst), rst
        for i, (r, o, w) in enumerate(zip(raws, routs, rwant)):
            assert o.tobytes() == w, f"round {rd}: ring encode differs, stream {i} of {len(r)} bytes"
        # small size classes too
        smalls = [gen(int(rng.integers(0, 7)), int(rng.integers(0, 4097))) for _ in range(60)]
        souts, sst = ctx.encode_batch(smalls, ring=True)
        for r, o in zip(smalls, souts):
            assert o.tobytes() == O.ring_encode(r), f"round {rd}: ring encode differs, small stream of {len(r)} bytes"
    dec, st2 = ctx.decode_batch(want)
    assert all(e == 0 for e in st2)
    for r, o in zip(raws, dec):
        assert o.tobytes() == r
    bad = []
    for w in want:
        b = bytearray(w)
        if rng.random() < 0.3:
            b = b[: int(rng.integers(4, len(b)))]
        else:
            k = int(rng.integers(0, len(b)))
            b[k] ^= int(rng.integers(1, 256))
        bad.append(bytes(b))
    caps = [len(r) + 64 for r in raws]
    got, gst = ctx.decode_batch(bad, caps=caps)
    for i, (b, cap, g, s) in enumerate(zip(bad, caps, got, gst)):
        ws = O.decode_status(b, cap)
        assert s == ws, f"round {rd}: status {s} vs {ws}, stream {i} of {len(b)} bytes"
        if s == 0:
            assert g.tobytes() == O.decode(b, cap=cap)
    print(f"round {rd}: {len(raws)} streams ok ({sum(map(len, raws)) / 1e6:.0f} MB), {time.time() - t0:.0f} s", flush=True)
assert ctx.pipe_refusals() == 0, "a hand-over of the pipelined LZ kernel was refused"
print(f"fuzz ok: {rounds} rounds, {total / 1e6:.0f} MB")
