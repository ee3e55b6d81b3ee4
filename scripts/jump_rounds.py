import sys
sys.path.insert(0,'/root/repo')
import bench, lzfse_rust_amd as lz
ctx = lz.Context(0, diag=True)
t = bench.synth_text(64 << 20, seed=1)
enc, st = ctx.encode_batch([t])
e = enc[0].tobytes()
ctx.set_option("decode_lanes", 1)
ctx.decode_batch([e])
ctx.set_option("diag_stats", 4)
ctx.decode_batch([e])
