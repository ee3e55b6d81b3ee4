import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import glob
import lzfse_rust_amd as m
ctx = m.Context(0)
g = os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'snappy')
fs = sorted(glob.glob(g + '/*.lzfse'))
raws, st = ctx.decode_batch([open(f, 'rb').read() for f in fs])
ctx.enable_timing(True)
names = [os.path.basename(f)[:-6] for f in fs]
def run(batch, dbg):
    os.environ["LZFSE_MI_CAND_DEBUG"] = str(dbg)
    for rep in range(2):
        outs, st = ctx.encode_batch(batch)
    t = ctx.timings()
    return {k: round(v[0], 3) for k, v in t.items() if k in ('enc_cand', 'enc_chain', 'enc_link', 'enc_spec', 'enc_block')}
allb = [r.tobytes() for r in raws] * 64
for dbg in (0, 1, 2, 3, 7):
    print('all dbg', dbg, run(allb, dbg))
for nm, r in zip(names, raws):
    b = [r.tobytes()] * 64
    mb = len(b[0]) * 64 / 1e6
    t = run(b, 0); t1 = run(b, 3)
    print(f"{nm:28s} {mb:7.1f} MB cand {t['enc_cand']:.3f} ms = {mb / t['enc_cand']:.1f} GB/s ; noLCP {t1['enc_cand']:.3f} ; chain {t['enc_chain']:.3f} spec {t['enc_spec']:.3f} block {t['enc_block']:.3f}")
