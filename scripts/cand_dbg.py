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
for nm, r in zip(names, raws):
    if False:
        continue
    b = [r.tobytes()] * 64
    print(nm, ' '.join(f"dbg{g}={run(b, g)['enc_cand']:.3f}" for g in (0, 1, 3, 7)))
