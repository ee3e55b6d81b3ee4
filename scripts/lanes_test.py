"""Wall time of batch calls through the library's own lane split (LZFSE_MI_LANES_ENC / _DEC are read once per process)."""
import sys, os, time, glob
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
import lzfse_rust_amd as lz
dev = torch.device('cuda', 0)
ctx = lz.Context(0)
timing = len(sys.argv) > 1 and sys.argv[1] == 'timing'
g = os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'snappy')
fs = sorted(glob.glob(g + '/*.lzfse'))
raws, st = ctx.decode_batch([open(f, 'rb').read() for f in fs])
batch = [r.tobytes() for r in raws] * 64
def layout(lens, align=256):
    off, o = [], 0
    for n in lens:
        off.append(o); o += (n + align - 1) // align * align
    return np.array(off, dtype=np.uint64), o
raw_len = np.array([len(r) for r in batch], dtype=np.uint64)
raw_off, tot = layout(raw_len)
enc_cap = np.array([lz.encode_bound(int(n)) for n in raw_len], dtype=np.uint64)
enc_off, etot = layout(enc_cap)
h = np.zeros(tot + 256, dtype=np.uint8)
for r, o in zip(batch, raw_off):
    h[int(o):int(o) + len(r)] = np.frombuffer(r, dtype=np.uint8)
d_raw = torch.from_numpy(h).to(dev)
d_enc = torch.zeros(etot + 256, dtype=torch.uint8, device=dev)
d_dec = torch.zeros(tot + 256, dtype=torch.uint8, device=dev)
ctx.enable_timing(timing)
total = int(raw_len.sum())
be = bd = 1e9
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    enc_len, est = ctx.encode_batch_device(d_raw.data_ptr(), raw_off, raw_len, d_enc.data_ptr(), enc_off, enc_cap)
    torch.cuda.synchronize(); be = min(be, time.perf_counter() - t0)
    t0 = time.perf_counter()
    ctx.decode_batch_device(d_enc.data_ptr(), enc_off, enc_len, d_dec.data_ptr(), raw_off, raw_len)
    torch.cuda.synchronize(); bd = min(bd, time.perf_counter() - t0)
assert torch.equal(d_dec[:tot], d_raw[:tot])
print(f"lanes enc={os.environ.get('LZFSE_MI_LANES_ENC')} dec={os.environ.get('LZFSE_MI_LANES_DEC')} timing={timing}: enc {be*1e3:.2f} ms {total/be/1e9:.2f} GB/s ; dec {bd*1e3:.2f} ms {total/bd/1e9:.2f} GB/s")
