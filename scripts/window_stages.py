"""What ONE window of a stream costs: a lone text stream of 4 / 16 / 64 MiB, decode and encode, device-resident (stage timers
beside the wall time of the call) and through host pointers (what lzfse_mi_dstream_* / _estream_* call per window).
    python scripts/window_stages.py      (profiles/r04_window_stages.txt)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import lzfse_rust_amd as lz
from bench import synth_text

ctx = lz.Context(0)
dev = torch.device("cuda", 0)
for mb in (4, 16, 64):
    raw = bytes(synth_text(mb << 20))
    n = len(raw); bound = (lz.encode_bound(n) + 255) & ~255; npad = (n + 255) & ~255
    d_raw = torch.from_numpy(np.frombuffer(raw + bytes(npad - n), dtype=np.uint8).copy()).to(dev)
    d_enc = torch.empty(bound, dtype=torch.uint8, device=dev)
    d_dec = torch.empty(npad + 64, dtype=torch.uint8, device=dev)
    so = np.zeros(1, dtype=np.uint64); sl = np.full(1, n, dtype=np.uint64)
    eo = np.zeros(1, dtype=np.uint64); ec = np.full(1, bound, dtype=np.uint64)
    el, st = ctx.encode_batch_device(d_raw.data_ptr(), so, sl, d_enc.data_ptr(), eo, ec)
    assert int(st[0]) == 0
    for what, call in (("decode", lambda: ctx.decode_batch_device(d_enc.data_ptr(), eo, el, d_dec.data_ptr(), so, sl)),
                       ("encode", lambda: ctx.encode_batch_device(d_raw.data_ptr(), so, sl, d_enc.data_ptr(), eo, ec))):
        ctx.enable_timing(True)
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        N = 10
        t0 = time.perf_counter()
        for _ in range(N):
            call()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / N
        tm = ctx.timings()
        ctx.enable_timing(False)
        print(f"{mb:3d} MiB {what} device-resident {wall * 1e3:7.3f} ms = {n / wall / 1e9:5.2f} GB/s   stages (ms): " +
              ", ".join(f"{k} {v[0]:.3f}" for k, v in sorted(tm.items()) if v[0] >= 0.02), flush=True)
    enc = ctx.encode_batch([raw])[0][0].tobytes()
    for what, call in (("decode", lambda: ctx.decode_batch([enc])), ("encode", lambda: ctx.encode_batch([raw]))):
        for _ in range(2):
            call()
        N = 5
        t0 = time.perf_counter()
        for _ in range(N):
            call()
        wall = (time.perf_counter() - t0) / N
        print(f"{mb:3d} MiB {what} host pointers   {wall * 1e3:7.3f} ms = {n / wall / 1e9:5.2f} GB/s", flush=True)
