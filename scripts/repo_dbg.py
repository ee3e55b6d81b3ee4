"""Debug aid for slices of several blocks (encode_slice_blocks, api.hip) at a small BLOCK_GUIDE: python scripts/repo_dbg.py zeros N | text | noise"""
import sys, os, glob
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np
import lzfse_rust_amd as m
from oracle_py import Oracle, rng_gen_vec
o = Oracle()
G, S = 0x100000, 0x20000
ctx = m.Context(0, diag=True)
ctx.set_option("diag_guide", G | (S << 32))
ctx.set_option("diag_stats", int(os.environ.get("DIAG_STATS", "8")))
kind = sys.argv[1] if len(sys.argv) > 1 else "zeros"
if kind == "zeros":
    data = bytes(int(sys.argv[2]) if len(sys.argv) > 2 else 3_500_000)
elif kind == "noise":
    data = rng_gen_vec(3, 3_000_000)
elif kind == "period":
    data = (rng_gen_vec(6, int(sys.argv[2]) if len(sys.argv) > 2 else 70_000) * 60)[:int(sys.argv[3]) if len(sys.argv) > 3 else 4_200_000]
else:
    raw = {os.path.basename(f)[:-6]: o.decode(open(f, "rb").read()) for f in glob.glob("tests/golden/snappy/*.lzfse")}
    data = (raw["lcet10.txt"] + raw["alice29.txt"] + raw["urls.10K"]) * 4
    if len(sys.argv) > 2: data = data[:int(sys.argv[2])]
out = bytearray()
try:
    m.LzfseEncoder(context=ctx).encode_bytes(data, out)
except Exception as e:
    print("FAILED", e)
want = o.encode_guide(data, G, S)
print("encoded", len(out), "oracle", len(want), bytes(out) == want)
if bytes(out) != want and len(out):
    k = next(i for i in range(min(len(out), len(want))) if out[i] != want[i])
    print("first difference at byte", k)
    a, b = o.decode_lmds(bytes(out))[1], o.decode_lmds(want)[1]
    pos_a = pos_b = 0
    for i, (x, y) in enumerate(zip(a, b)):
        if tuple(x) != tuple(y):
            print("first differing LMD", i, "at raw position", pos_a, "device", tuple(x), "oracle", tuple(y))
            print("device next:", [tuple(int(q) for q in v) for v in a[i:i + 4]]); print("oracle next:", [tuple(int(q) for q in v) for v in b[i:i + 4]])
            break
        pos_a += int(x[0]) + int(x[1])
    print("lmds", len(a), len(b))
