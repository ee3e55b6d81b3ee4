import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import glob
from oracle_py import Oracle
import lzfse_rust_amd as m
o = Oracle(); ctx = m.Context(0)
g = os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'snappy')
name = sys.argv[1] if len(sys.argv) > 1 else 'fireworks.jpeg'
raw = o.decode(open(os.path.join(g, name + '.lzfse'), 'rb').read())
exp, matches, blocks, packs = o.encode_trace(raw)
outs, st = ctx.encode_batch([raw])
got = outs[0].tobytes()
print('status', st, 'len', len(got), len(exp), 'equal', got == exp)
raw2, lm = o.decode_lmds(got)
print('roundtrip ok', raw2 == raw, 'n_lmds gpu', len(lm), 'oracle packs', len(packs))
# rebuild positions from lmd lists
def events(lms):
    pos = 0; out = []
    for l, mm, d in lms:
        out.append((pos, l, mm, d)); pos += l + mm
    return out
_, lo = o.decode_lmds(exp)
eg, eo = events(lm), events(lo)
for i, (a, b) in enumerate(zip(eg, eo)):
    if a != b:
        print('first diff at lmd', i, 'gpu', eg[max(0,i-2):i+3], 'oracle', eo[max(0,i-2):i+3]); break
print('oracle matches around:', [mm for mm in matches if abs(mm[1] - eo[i][0]) < 6000][:12])
