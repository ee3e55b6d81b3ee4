"""GPU parity at the sizes BASELINE.json's configs name: the HIP path through the C ABI against the
oracle's bytes, not against itself.

    config 2  one 64 MiB bvx2 stream produced by the ORACLE, decoded on the GPU          == the raw text
    config 3  64 MiB of text as ONE stream, encoded on the GPU                           == the oracle's bytes (SHA-256)
    config 5  perturbed copies of that text cut into independent 4 MiB streams           == the oracle's bytes per chunk

These sizes are the only ones that reach the stitcher over 32 k segments, 31-bit jump origins, the 5-tile link window
and > 800 blocks per stream."""
import hashlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def ctx():
    import lzfse_rust_amd as m
    return m.Context(0)


@pytest.fixture(scope="module")
def text64():
    import bench
    return bench.synth_text(64 << 20, seed=1)


@pytest.fixture(scope="module")
def text64_oracle_stream(oracle, text64):
    return oracle.encode(text64)


def test_config3_encode_64mib_single_stream_equals_oracle(ctx, text64, text64_oracle_stream):
    outs, st = ctx.encode_batch([text64])
    assert st[0] == 0
    assert len(outs[0]) == len(text64_oracle_stream)
    assert hashlib.sha256(outs[0].tobytes()).digest() == hashlib.sha256(text64_oracle_stream).digest()


def test_config2_decode_64mib_oracle_stream(ctx, text64, text64_oracle_stream):
    """One stream of > 800 bvx2 blocks written by the oracle: the default path choice (pointer jumping) at full size."""
    outs, st = ctx.decode_batch([text64_oracle_stream])
    assert st[0] == 0
    assert len(outs[0]) == len(text64)
    assert hashlib.sha256(outs[0].tobytes()).digest() == hashlib.sha256(text64).digest()


def test_config2_decode_64mib_tile_path(diag_ctx, text64, text64_oracle_stream):
    """The same stream through the per-stream tile kernel (what a batch of many such streams takes)."""
    diag_ctx.set_option("diag_lz_path", 0)
    try:
        outs, st = diag_ctx.decode_batch([text64_oracle_stream])
    finally:
        diag_ctx.set_option("diag_lz_path", -1)
    assert st[0] == 0
    assert hashlib.sha256(outs[0].tobytes()).digest() == hashlib.sha256(text64).digest()


def test_config5_chunks_4mib_each_equals_oracle(ctx, oracle, text64):
    """256 MiB = 4 perturbed copies of the text (bench.py's chunks1g recipe), 64 independent 4 MiB streams."""
    from lzfse_rust_amd import sharding
    base = np.frombuffer(text64, dtype=np.uint8)
    chunks = []
    for c in range(4):
        a = base.copy()
        a[c % 251::251] ^= np.uint8(1 + c)
        chunks += [a[o:o + n].tobytes() for o, n in sharding.chunk_bounds(a.size, 4 << 20)]
    assert len(chunks) == 64
    want = [oracle.encode(c) for c in chunks]
    outs, st = ctx.encode_batch(chunks)
    assert all(e == 0 for e in st)
    for i, (o, w) in enumerate(zip(outs, want)):
        assert o.tobytes() == w, f"chunk {i}"
    # and back: the oracle's streams through the GPU decoder, the GPU's through the oracle
    dec, st2 = ctx.decode_batch(want)
    assert all(e == 0 for e in st2)
    for i, (d, c) in enumerate(zip(dec, chunks)):
        assert d.tobytes() == c, f"chunk {i}"


def test_encode_rejects_streams_beyond_i32(ctx):
    """E20: inputs > 0x8000_0002 bytes (BLOCK_GUIDE + 3: more than one block of the reference's slice front end) need its
    reposition path (frontend_bytes.rs:348-375), which is not built: the stream is refused with LZFSE_MI_UNSUPPORTED before
    anything is read (lengths are host arrays). Up to that size a slice is one block: tests/test_gpu_big.py."""
    off = np.zeros(2, dtype=np.uint64)
    ln = np.array([0x8000_0003, 0xFFFF_FFFF_0], dtype=np.uint64)
    cap = np.array([4096, 4096], dtype=np.uint64)
    out_len, st = ctx.encode_batch_device(0, off, ln, 0, off, cap)   # no byte of a refused stream is touched
    assert list(st) == [9, 9] and list(out_len) == [0, 0]
