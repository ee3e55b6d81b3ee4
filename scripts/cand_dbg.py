import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import glob
import lzfse_rust_amd as m
ctx = m.Context(0)
g = os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'snappy')
fs = sorted(glob.glob(g + '/*.lzfse'))
raws, st = ctx.decode_batch([open(f, 'rb').read() for f in fs])
batch = [r.tobytes() for r in raws] * 64
ctx.enable_timing(True)
for dbg in (0, 8):
    os.environ["LZFSE_MI_CAND_DEBUG"] = str(dbg)
    for rep in range(2):
        outs, st = ctx.encode_batch(batch)
    print(dbg, {k: round(v[0], 2) for k, v in ctx.timings().items() if k in ('enc_cand', 'enc_chain', 'enc_link')})
