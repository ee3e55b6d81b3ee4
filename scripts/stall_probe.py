"""Where does a 3 ms call sometimes take 10? (round-4 verdict: kppkn.gtb x 64 encode 4.55 GB/s, sd 123 %.) Every Snappy file alone as a
batch of R copies resident in HBM, N samples of the encode and of the decode call: wall time per sample, and for the slow ones
(> 1.5 x the median) what the stage timers say -- a slow sample whose kernels took their usual time waited on the host.
    python scripts/stall_probe.py [R] [N] [timing|spin|spin-nogc|sleep]
       timing: stage timers on (the calls then run as one pass, without the second lane); spin: half a second of the CPU port between
       the files, as bench.py --per-file does; spin-nogc: the same with Python's cyclic collector off; sleep: idle instead"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import lzfse_rust_amd as lz
import bench

R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
MODE = sys.argv[3] if len(sys.argv) > 3 else ""      # (+ "null": on the legacy default stream, e.g. spin-null)
TIMING = MODE == "timing"
if MODE == "spin-nogc":
    import gc
    gc.disable()
if MODE.startswith("spin"):
    from oracle_py import Oracle
    O = Oracle("liblzfse_oracle_native.so")
dev = torch.device("cuda:0")

ctx = lz.Context(0, diag="trace" in MODE)
if "trace" in MODE:     # the diagnostic build prints where the host's time went in every encode call (stderr)
    ctx.set_option("diag_stats", 4)
if "null" in MODE:      # as bench.py does: the context works on torch's current stream, which is the legacy default stream
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
names, streams = bench.load_corpus() if hasattr(bench, "load_corpus") else (None, None)
if names is None:
    import glob
    g = os.path.join(ROOT, "tests", "golden", "snappy")
    files = sorted(glob.glob(os.path.join(g, "*.lzfse")))
    names = [os.path.basename(f)[:-6] for f in files]
    streams = [open(f, "rb").read() for f in files]
raws, st = ctx.decode_batch(streams)
assert all(s == 0 for s in st)
if TIMING:
    ctx.enable_timing(True)
for name, r in zip(names, raws):
    raw = r.tobytes()
    if MODE.startswith("spin"):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.5:
            O.encode(raw)
    elif MODE == "sleep":
        time.sleep(0.5)
    B = bench.DeviceBatch(torch, dev, lz, [raw] * R)
    enc_len, est = ctx.encode_batch_device(B.d_raw.data_ptr(), B.raw_off, B.raw_len, B.d_enc.data_ptr(), B.enc_off, B.enc_cap)
    te, td, ke, kd = [], [], [], []
    for it in range(N + 2):
        torch.cuda.synchronize()
        if "trace" in MODE:
            print(f"sample {name} {it - 2}", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        ctx.encode_batch_device(B.d_raw.data_ptr(), B.raw_off, B.raw_len, B.d_enc.data_ptr(), B.enc_off, B.enc_cap)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if TIMING: ke.append(ctx.timings())
        t1b = time.perf_counter()
        ctx.decode_batch_device(B.d_enc.data_ptr(), B.enc_off, enc_len, B.d_dec.data_ptr(), B.raw_off, B.raw_len)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if TIMING: kd.append(ctx.timings())
        te.append(t1 - t0); td.append(t2 - t1b)
        if "trace" in MODE:
            print(f"wall {name} {it - 2} encode {1e3 * (t1 - t0):.3f} decode {1e3 * (t2 - t1b):.3f}", file=sys.stderr, flush=True)
    te, td = np.array(te[2:]) * 1e3, np.array(td[2:]) * 1e3
    ke, kd = ke[2:], kd[2:]
    for what, t, k in (("encode", te, ke), ("decode", td, kd)):
        med = float(np.median(t))
        slow = [i for i in range(len(t)) if t[i] > 1.5 * med]
        line = f"{name:28s} x{R} {what}: median {med:.3f} ms, mean {t.mean():.3f}, max {t.max():.3f}, sd {100 * t.std() / t.mean():.0f} %, {len(slow)} of {len(t)} samples > 1.5 x median"
        if slow and TIMING:
            i = slow[0]
            tot = lambda d: sum(v[0] for v in d.values())
            j = int(np.argsort(t)[len(t) // 2])
            line += f" | slow sample {i}: wall {t[i]:.3f} ms, kernels {tot(k[i]):.3f} ms; a median sample: wall {t[j]:.3f}, kernels {tot(k[j]):.3f}"
            worst = max(k[i], key=lambda s: k[i][s][0] - k[j].get(s, (0, 0))[0])
            line += f"; largest difference: {worst} {k[i][worst][0]:.3f} vs {k[j].get(worst, (0, 0))[0]:.3f} ms"
        print(line, flush=True)
    del B
