import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
os.environ["LZFSE_MI_WALK_STATS"] = "1"
import glob
import lzfse_rust_amd as m
ctx = m.Context(0)
g = os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'snappy')
fs = sorted(glob.glob(g + '/*.lzfse'))
raws, st = ctx.decode_batch([open(f, 'rb').read() for f in fs])
print([os.path.basename(f) for f in fs], file=sys.stderr)
ctx.enable_timing(True)
for rep in range(2):
    outs, st = ctx.encode_batch([r.tobytes() for r in raws])
    print(ctx.timings(), file=sys.stderr)
