"""dec_fse ms of the two entropy-stage kernels (diagnostic build: LZFSE_MI_OPT_DIAG_FSE) on batches of different sizes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import bench
import lzfse_rust_amd as lz

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ctx = lz.Context(0, diag=True)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
names, fixture_streams, hashes = bench.load_corpus_streams()
raws_np, st = ctx.decode_batch(fixture_streams)
raws = [r.tobytes() for r in raws_np]
text = bench.synth_text(64 << 20, seed=1)
cases = {"snappy x 192": raws * 192, "snappy x 340": raws * 340, "snappy x 512": raws * 512,
         "chunks 256 x 4 MiB": [text[i:i + (4 << 20)] for i in range(0, 64 << 20, 4 << 20)] * 16}
for name, batch in cases.items():
    B = bench.DeviceBatch(torch, dev, lz, batch)
    enc_len, est = ctx.encode_batch_device(B.d_raw.data_ptr(), B.raw_off, B.raw_len, B.d_enc.data_ptr(), B.enc_off, B.enc_cap)
    assert (est == 0).all()
    ctx.set_option("decode_lanes", 1)
    ctx.enable_timing(True)
    out = []
    for mode in (1, 2):
        ctx.set_option("diag_fse", mode)
        best = 1e9
        for _ in range(4):
            _, dst = ctx.decode_batch_device(B.d_enc.data_ptr(), B.enc_off, enc_len, B.d_dec.data_ptr(), B.raw_off, B.raw_len)
            torch.cuda.synchronize()
            assert (dst == 0).all()
            best = min(best, ctx.timings()["dec_fse"][0])
        assert torch.equal(B.d_dec[:B.raw_padded], B.d_raw[:B.raw_padded])
        out.append(best)
    ctx.set_option("diag_fse", 0)
    print(f"{name:24s} streams {len(batch):5d}  dec_fse ms: one block per workgroup {out[0]:.3f}   four blocks per workgroup {out[1]:.3f}")
