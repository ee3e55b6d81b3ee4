"""Per-phase cycle counts of the LZ tile kernel (diagnostic build, LZFSE_MI_OPT_DIAG_STATS bit 4) on the Snappy corpus x 64."""
import sys, os, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import lzfse_rust_amd as lz
files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/snappy/*.lzfse")))
encs = [open(f, "rb").read() for f in files]
ctx = lz.Context(0, diag=True)
ctx.set_option("decode_lanes", 1)
ctx.decode_batch(encs * 64)
ctx.set_option("diag_stats", 4)
for f in files: print(os.path.basename(f), file=sys.stderr)
ctx.decode_batch(encs * 64)
