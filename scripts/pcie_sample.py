"""bench.py's pcie_inclusive sample by itself (384 streams = 32 copies of the Snappy files, 94 MB, host pointers in and out), a third of it and four times it.
    python scripts/pcie_sample.py [copies ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench
import lzfse_rust_amd as lz
ctx = lz.Context(0)
names, streams, hashes = bench.load_corpus_streams()
raws = [r.tobytes() for r in ctx.decode_batch(streams)[0]]
for copies in ([int(a) for a in sys.argv[1:]] or [10, 32, 128]):
    r = bench.pcie_inclusive(ctx, lz, raws * copies, 12 * copies)
    print(copies * 12, "streams", r["sample"].split(",")[1].strip(), "encode", r["encode_MBps"], "decode", r["decode_MBps"])
