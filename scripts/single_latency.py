"""Latency of ONE small stream per call (html, 100 KB): host-pointer API and device-resident API."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
import lzfse_rust_amd as lz
ctx = lz.Context(0)
g = os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'snappy')
enc0 = open(os.path.join(g, 'html.lzfse'), 'rb').read()
raws, st = ctx.decode_batch([enc0])
raw = raws[0].tobytes()
enc = lz.LzfseEncoder(context=ctx); dec = lz.LzfseDecoder(context=ctx)
def t(f, n=50):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n
out = bytearray()
def e():
    out.clear(); enc.encode_bytes(raw, out)
e(); stream = bytes(out)
def d():
    o = bytearray(); dec.decode_bytes(stream, o)
te, td = t(e), t(d)
print(f"host API, one html stream ({len(raw)} B): encode {te*1e3:.3f} ms = {len(raw)/te/1e6:.1f} MB/s ; decode {td*1e3:.3f} ms = {len(raw)/td/1e6:.1f} MB/s")
dev = torch.device('cuda', 0)
d_raw = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
d_enc = torch.zeros(lz.encode_bound(len(raw)), dtype=torch.uint8, device=dev)
d_dec = torch.zeros(len(raw) + 64, dtype=torch.uint8, device=dev)
z = np.zeros(1, dtype=np.uint64); ln = np.array([len(raw)], dtype=np.uint64); cap = np.array([d_enc.numel()], dtype=np.uint64)
el, _ = ctx.encode_batch_device(d_raw.data_ptr(), z, ln, d_enc.data_ptr(), z, cap)
te = t(lambda: ctx.encode_batch_device(d_raw.data_ptr(), z, ln, d_enc.data_ptr(), z, cap))
td = t(lambda: ctx.decode_batch_device(d_enc.data_ptr(), z, el, d_dec.data_ptr(), z, ln))
print(f"device API, one html stream: encode {te*1e3:.3f} ms = {len(raw)/te/1e6:.1f} MB/s ; decode {td*1e3:.3f} ms = {len(raw)/td/1e6:.1f} MB/s")
