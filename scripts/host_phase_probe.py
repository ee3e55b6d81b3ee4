"""Host phases of one device-batch encode / decode call of Snappy x R (the diagnostic build, LZFSE_MI_OPT_DIAG_STATS & 4): python scripts/host_phase_probe.py R [lanes]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import glob
import numpy as np, torch
import lzfse_rust_amd as lz
import bench
R = int(sys.argv[1]); lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
ctx = lz.Context(0, diag=True)
ctx.set_option("encode_lanes", lanes); ctx.set_option("decode_lanes", lanes)
files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "snappy", "*.lzfse")))
raws, st = ctx.decode_batch([open(f, "rb").read() for f in files])
B = bench.DeviceBatch(torch, dev, lz, [r.tobytes() for r in raws] * R)
for it in range(4):
    if it == 3: ctx.set_option("diag_stats", 4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    enc_len, est = ctx.encode_batch_device(B.d_raw.data_ptr(), B.raw_off, B.raw_len, B.d_enc.data_ptr(), B.enc_off, B.enc_cap)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ctx.decode_batch_device(B.d_enc.data_ptr(), B.enc_off, enc_len, B.d_dec.data_ptr(), B.raw_off, B.raw_len)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"R={R} lanes={lanes} it={it}: encode {1e3 * (t1 - t0):.2f} ms, decode {1e3 * (t2 - t1):.2f} ms", flush=True)
