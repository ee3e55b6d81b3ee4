"""Per-phase cycle counts of the LZ tile kernel (diagnostic build, LZFSE_MI_OPT_DIAG_STATS bit 4) on 64 x 4 MiB text."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import lzfse_rust_amd as lz
ctx = lz.Context(0, diag=True)
t = bench.synth_text(64 << 20, seed=1)
chunks = [t[i:i + (4 << 20)] for i in range(0, len(t), 4 << 20)] * 4
encs, st = ctx.encode_batch(chunks)
encs = [e.tobytes() for e in encs]
for variant in (1, 0):
    ctx.set_option("diag_lz_path", 0)
    ctx.set_option("diag_lz_tile", variant)
    ctx.set_option("decode_lanes", 1)
    ctx.decode_batch(encs[:8])
    ctx.set_option("diag_stats", 4)
    print(f"--- tile variant {variant}", file=sys.stderr)
    t0 = time.perf_counter()
    ctx.decode_batch(encs)
    print(f"decode_batch wall {time.perf_counter() - t0:.4f} s for {sum(map(len, chunks)) / 1e6:.0f} MB", file=sys.stderr)
    ctx.set_option("diag_stats", 0)
