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
exp = {n: o.encode(raws[n]) for n in names}
def run(sel):
    outs, st = ctx.encode_batch([raws[n] for n in sel])
    return [(n, outs[i].tobytes() == exp[n]) for i, n in enumerate(sel)]
print(run(names))
print(run(['fireworks.jpeg', 'html']))
print(run(['html', 'fireworks.jpeg']))
print(run(['alice29.txt', 'fireworks.jpeg']))
print(run(['fireworks.jpeg', 'fireworks.jpeg']))
print(run(['paper-100k.pdf', 'html_x_4']))
