import sys, glob, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import lzfse_rust_amd as m
import oracle_py
O = oracle_py.Oracle()
raws = [O.decode(open(f, 'rb').read()) for f in sorted(glob.glob('/root/repo/tests/golden/snappy/*.lzfse'))]
c = m.Context(0, diag=True)
c.set_option("encode_lanes", 1)
c.set_option("diag_stats", 1)
outs, st = c.encode_batch(raws * 32)
outs, st = c.encode_batch(raws * 32)
