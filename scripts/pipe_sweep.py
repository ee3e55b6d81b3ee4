"""LZ stage of decode with K workgroups per stream (LZFSE_MI_OPT_DECODE_PIPE): stage time of dec_lz for N streams of S MiB
of synthetic text, one pass (no sub-batches), over K and both tile sizes.
    python scripts/pipe_sweep.py [N] [S_MiB]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import lzfse_rust_amd as m
from bench import synth_text


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    mib = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    ctx = m.Context(0)
    ctx.set_option("decode_lanes", 1)
    raw = bytes(synth_text(n * mib << 20))
    chunks = [raw[i * (mib << 20):(i + 1) * (mib << 20)] for i in range(n)]
    encs, st = ctx.encode_batch(chunks)
    assert all(e == 0 for e in st)
    encs = [e.tobytes() for e in encs]
    ctx.enable_timing(True)
    pipes = [1] + [k | v << 8 for v in (0, 1) for k in (1, 2, 4, 8, 16, 32) if not (v == 1 and k > 8)] + [0]
    for pipe in pipes:
        ctx.set_option("decode_pipe", pipe)
        best = None
        for _ in range(3):
            outs, st = ctx.decode_batch(encs)
            t = ctx.timings()
            lz = t["dec_lz"][0]
            best = lz if best is None else min(best, lz)
        assert all(e == 0 for e in st) and all(o.tobytes() == c for o, c in zip(outs, chunks))
        name = "off" if pipe == 1 else "auto" if pipe == 0 else f"K={pipe & 0xFF:2d} {'1024/32K' if pipe >> 8 else ' 256/8K '}"
        print(f"{name:16s} dec_lz {best:7.3f} ms   ({n * mib * 1.048576 / best:6.1f} GB/s)", flush=True)


if __name__ == "__main__":
    main()
