import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import glob
from oracle_py import Oracle
import lzfse_rust_amd as m
o = Oracle(); ctx = m.Context(0)
g = os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'snappy')
names = sorted(os.path.basename(f)[:-6] for f in glob.glob(g + '/*.lzfse'))
raws = {n: o.decode(open(os.path.join(g, n + '.lzfse'), 'rb').read()) for n in names}
outs, st = ctx.encode_batch([raws[n] for n in names])
def events(lms):
    pos = 0; out = []
    for l, mm, d in lms:
        out.append((pos, l, mm, d)); pos += l + mm
    return out
bad = 0
for i, n in enumerate(names):
    exp = o.encode(raws[n])
    got = outs[i].tobytes()
    if got != exp:
        bad += 1
        r2, lm = o.decode_lmds(got); _, lo = o.decode_lmds(exp)
        eg, eo = events(lm), events(lo)
        for k, (a, b) in enumerate(zip(eg, eo)):
            if a != b:
                print('BAD', n, 'roundtrip', r2 == raws[n], 'first diff lmd', k, 'pos', a[0], 'seg', a[0] // 4096, 'gpu', eg[k:k+3], 'oracle', eo[k:k+3]); break
print('bad streams:', bad)
