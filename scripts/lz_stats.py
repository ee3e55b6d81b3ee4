import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
os.environ["LZFSE_MI_LZ_STATS"] = "1"
import glob
import lzfse_rust_amd as m
ctx = m.Context(0)
g = os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'snappy')
fs = sorted(glob.glob(g + '/*.lzfse'))
print([os.path.basename(f)[:-6] for f in fs], file=sys.stderr)
for v in ("0", "1"):
    os.environ["LZFSE_MI_LZ_VARIANT"] = v
    print("variant", v, file=sys.stderr)
    raws, st = ctx.decode_batch([open(f, 'rb').read() for f in fs])
