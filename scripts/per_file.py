import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import glob
import lzfse_rust_amd as m
ctx = m.Context(0)
g = os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'snappy')
fs = sorted(glob.glob(g + '/*.lzfse'))
raws, st = ctx.decode_batch([open(f, 'rb').read() for f in fs])
ctx.enable_timing(True)
R = 64
for f, r in zip(fs, raws):
    b = r.tobytes()
    for rep in range(2):
        outs, st = ctx.encode_batch([b] * R)
    te = ctx.timings()
    encs = [o.tobytes() for o in outs]
    for rep in range(2):
        d, st = ctx.decode_batch(encs, caps=[len(b)] * R)
    td = ctx.timings()
    mb = len(b) * R / 1e6
    print(f"{os.path.basename(f)[:-6]:28s} {mb:7.1f} MB  enc: " + " ".join(f"{k[4:]}={v[0]:.2f}" for k, v in te.items()) + "  | dec: " + " ".join(f"{k[4:]}={v[0]:.2f}" for k, v in td.items()))
