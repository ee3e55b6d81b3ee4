"""Does running two half-batches concurrently (two contexts = two HIP streams, two host threads) beat one batch?"""
import sys, os, time, threading, glob
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
import lzfse_rust_amd as lz
dev = torch.device('cuda', 0)
NW = int(sys.argv[1]) if len(sys.argv) > 1 else 2
STAG = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
os.environ['LZFSE_MI_NO_SPLIT'] = '1'
ctxs = [lz.Context(0) for _ in range(NW)]
g = os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'snappy')
fs = sorted(glob.glob(g + '/*.lzfse'))
raws, st = ctxs[0].decode_batch([open(f, 'rb').read() for f in fs])
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
enc_len, est = ctxs[0].encode_batch_device(d_raw.data_ptr(), raw_off, raw_len, d_enc.data_ptr(), enc_off, enc_cap)
assert (est == 0).all()
total = int(raw_len.sum())

def run(parts, what):
    def work(ctx, idx):
        if what == 'enc':
            ctx.encode_batch_device(d_raw.data_ptr(), raw_off[idx], raw_len[idx], d_enc.data_ptr(), enc_off[idx], enc_cap[idx])
        else:
            ctx.decode_batch_device(d_enc.data_ptr(), enc_off[idx], enc_len[idx], d_dec.data_ptr(), raw_off[idx], raw_len[idx])
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ths = [threading.Thread(target=work, args=(ctxs[i], parts[i])) for i in range(len(parts))]
        for t in ths:
            t.start()
            if STAG: time.sleep(STAG)
        for t in ths: t.join()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best
n = len(batch)
allidx = np.arange(n)
for what in ('enc', 'dec'):
    t1 = run([allidx], what)
    parts = [allidx[i::NW] for i in range(NW)]
    t2 = run(parts, what)
    print(f"{what}: one batch {t1*1e3:.2f} ms = {total/t1/1e9:.2f} GB/s ; {NW} concurrent parts {t2*1e3:.2f} ms = {total/t2/1e9:.2f} GB/s")
assert torch.equal(d_dec[:tot], d_raw[:tot])
