"""Where a small encode call's time goes: html / alice29 / urls alone and 16 copies, device-resident, stage timers beside the wall time.
    python scripts/small_encode_stages.py      (profiles/r04_small_encode.txt)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import lzfse_rust_amd as lz
ctx = lz.Context(0)
g = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "snappy")
dev = torch.device("cuda", 0)
for name in ("html", "alice29.txt", "urls.10K"):
    enc0 = open(os.path.join(g, name + ".lzfse"), "rb").read()
    raw = ctx.decode_batch([enc0])[0][0].tobytes()
    for R in (1, 16):
        n = len(raw); bound = (lz.encode_bound(n) + 255) & ~255; npad = (n + 255) & ~255
        d_raw = torch.from_numpy(np.tile(np.frombuffer(raw + bytes(npad - n), dtype=np.uint8), R)).to(dev)
        d_enc = torch.empty(bound * R, dtype=torch.uint8, device=dev)
        so = np.arange(R, dtype=np.uint64) * npad; sl = np.full(R, n, dtype=np.uint64)
        eo = np.arange(R, dtype=np.uint64) * bound; ec = np.full(R, bound, dtype=np.uint64)
        ctx.enable_timing(True)
        for _ in range(5):
            ctx.encode_batch_device(d_raw.data_ptr(), so, sl, d_enc.data_ptr(), eo, ec)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        N = 30
        for _ in range(N):
            ctx.encode_batch_device(d_raw.data_ptr(), so, sl, d_enc.data_ptr(), eo, ec)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / N
        tm = ctx.timings()
        ctx.enable_timing(False)
        print(f"{name:12s} x{R:<3d} encode call {wall * 1e3:6.3f} ms = {n * R / wall / 1e9:5.2f} GB/s   stages (ms): " + ", ".join(f"{k} {v[0]:.3f}" for k, v in sorted(tm.items()) if v[0] >= 0.004))
