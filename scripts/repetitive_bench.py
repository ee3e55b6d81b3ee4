"""Encode and decode throughput on highly repetitive inputs (device-resident batches of 256 x 1 MiB): the inputs on which
capped candidate records make the segment walkers ask for exact lengths most often, and on which a tile of the LZ stage is
mostly matches that overlap themselves.   python scripts/repetitive_bench.py      (profiles/r03_repetitive.txt)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lzfse_rust_amd as lz
ctx = lz.Context(0)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)

def periodic(per, n, mut):
    chunk = rng.integers(0, 256, size=per, dtype=np.uint8)
    a = np.tile(chunk, n // per + 1)[:n].copy()
    if mut:
        idx = rng.integers(0, n, size=n // mut)
        a[idx] ^= rng.integers(1, 256, size=idx.size, dtype=np.uint8)
    return a

cases = [("period 300", lambda: periodic(300, 1 << 20, 0)), ("period 1100", lambda: periodic(1100, 1 << 20, 0)),
         ("period 1100, 1 change per 700 B", lambda: periodic(1100, 1 << 20, 700)), ("period 40, 1 change per 60 B", lambda: periodic(40, 1 << 20, 60)),
         ("period 5000, 1 change per 300 B", lambda: periodic(5000, 1 << 20, 300)), ("period 250000", lambda: periodic(250000, 1 << 20, 0)),
         ("period 7", lambda: periodic(7, 1 << 20, 0)),
         ("runs of 3 .. 400 equal bytes", lambda: np.repeat(rng.integers(0, 256, size=12000, dtype=np.uint8), rng.integers(3, 400, size=12000))[:1 << 20].copy()),
         ("runs of 2 400 .. 7 000 equal bytes", lambda: np.repeat(rng.integers(0, 256, size=900, dtype=np.uint8), rng.integers(2400, 7000, size=900))[:1 << 20].copy()),
         ("zeros", lambda: np.zeros(1 << 20, dtype=np.uint8))]
only = [a[5:] for a in sys.argv[1:] if a.startswith('only=')]   # only=zeros only='period 1100' ...
for name, gen in cases:
    if only and name not in only:
        continue
    parts = [gen() for _ in range(4)] * 64          # 256 streams of 1 MiB
    n = parts[0].size
    src = torch.from_numpy(np.concatenate(parts)).to(dev)
    bound = (lz.encode_bound(n) + 255) & ~255
    dst = torch.empty(bound * len(parts), dtype=torch.uint8, device=dev)
    so = np.arange(len(parts), dtype=np.uint64) * n
    sl = np.full(len(parts), n, dtype=np.uint64)
    do = np.arange(len(parts), dtype=np.uint64) * bound
    dc = np.full(len(parts), bound, dtype=np.uint64)
    ctx.encode_batch_device(src.data_ptr(), so, sl, dst.data_ptr(), do, dc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ol, st = ctx.encode_batch_device(src.data_ptr(), so, sl, dst.data_ptr(), do, dc)
    t1 = time.perf_counter()
    assert all(int(e) == 0 for e in st)
    if len(sys.argv) > 1 and sys.argv[1] == "stages":   # where the encoder's time goes (one more pass, one lane, stage timers on)
        ctx.set_option("encode_lanes", 1); ctx.enable_timing(True)
        ctx.encode_batch_device(src.data_ptr(), so, sl, dst.data_ptr(), do, dc)
        print("      ", {k: round(v[0], 2) for k, v in sorted(ctx.timings().items()) if v[0] > 0.15})
        ctx.enable_timing(False); ctx.set_option("encode_lanes", 0)
    back = torch.empty(n * len(parts), dtype=torch.uint8, device=dev)
    el = np.asarray([int(x) for x in ol], dtype=np.uint64)
    ctx.decode_batch_device(dst.data_ptr(), do, el, back.data_ptr(), so, sl)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ol2, st2 = ctx.decode_batch_device(dst.data_ptr(), do, el, back.data_ptr(), so, sl)
    t3 = time.perf_counter()
    assert all(int(e) == 0 for e in st2) and torch.equal(back, src)
    print(f"{name:34s} encode {src.numel() / (t1 - t0) / 1e9:6.2f} GB/s, decode {src.numel() / (t3 - t2) / 1e9:6.2f} GB/s, ratio {src.numel() / float(sum(int(x) for x in ol)):.0f}")
