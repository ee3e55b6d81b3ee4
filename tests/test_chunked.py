"""Chunked container + one-process multi-device entry points + CLI (reference CLI lzfoo/main.rs)."""
import ctypes as C
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_frame_parsing_needs_no_device():
    """lzfse_mi_decode_chunked_size is header arithmetic (like decode::probe): works on CPU, rejects damaged frames."""
    from lzfse_rust_amd import _native
    L = _native.lib()
    assert L.lzfse_mi_chunked_bound(10 << 20, 0) >= (10 << 20)
    frame = bytearray(b"LZMC" + (1).to_bytes(2, "little") + bytes(2) + (4096).to_bytes(4, "little") + (2).to_bytes(4, "little")
                      + (5000).to_bytes(8, "little") + (4096).to_bytes(4, "little") + (10).to_bytes(4, "little")
                      + (904).to_bytes(4, "little") + (6).to_bytes(4, "little") + bytes(16))
    v = C.c_uint64(0)

    def size(b):
        a = np.frombuffer(bytes(b), dtype=np.uint8)
        return L.lzfse_mi_decode_chunked_size(a.ctypes.data, a.size, C.byref(v))
    assert size(frame) == 0 and v.value == 5000
    assert size(frame[:-1]) == 8            # payload underflow
    assert size(frame + b"x") == 7          # payload overflow
    bad = bytearray(frame); bad[0] = ord("X")
    assert size(bad) == 2                   # bad block (magic)
    bad = bytearray(frame); bad[16] ^= 1
    assert size(bad) == 11                  # table does not add up to raw_total
    assert size(frame[:10]) == 8


@pytest.mark.gpu
def test_chunked_round_trip_and_chunk_streams_are_plain_lzfse(oracle, snappy_raw):
    """Two contexts on the device (stand-ins for two devices: chunk c -> contexts[c mod 2]); every chunk stream inside the
    frame equals the oracle's stream of that chunk, so any LZFSE decoder reads it."""
    import lzfse_rust_amd as m
    ctxs = [m.Context(0), m.Context(0)]
    data = (snappy_raw["urls.10K"] + snappy_raw["html"]) * 7 + b"tail"
    chunk = 1 << 20
    frame = m.encode_chunked(ctxs, data, chunk).tobytes()
    assert frame[:4] == b"LZMC"
    n_chunks = int.from_bytes(frame[12:16], "little")
    assert n_chunks == (len(data) + chunk - 1) // chunk and int.from_bytes(frame[16:24], "little") == len(data)
    pos = 24 + 8 * n_chunks
    for c in range(n_chunks):
        raw_len = int.from_bytes(frame[24 + 8 * c:28 + 8 * c], "little")
        enc_len = int.from_bytes(frame[28 + 8 * c:32 + 8 * c], "little")
        assert frame[pos:pos + enc_len] == oracle.encode(data[c * chunk:c * chunk + raw_len]), c
        pos += enc_len
    assert pos == len(frame)
    assert m.decode_chunked(ctxs, frame).tobytes() == data
    assert m.decode_chunked(ctxs[:1], frame).tobytes() == data
    # a destination below lzfse_mi_chunked_bound that still holds the frame (chunks encoded into private buffers instead of
    # into the frame): the same frame; one that does not hold it: BUFFER_OVERFLOW (6)
    assert m.encode_chunked(ctxs, data, chunk, cap=len(frame) + 100).tobytes() == frame
    assert m.encode_chunked(ctxs, data, chunk, cap=len(frame)).tobytes() == frame
    with pytest.raises(m.LzfseError) as ei:
        m.encode_chunked(ctxs, data, chunk, cap=len(frame) - 1)
    assert ei.value.status == 6
    # default chunk size, empty input, damaged chunk
    assert m.decode_chunked(ctxs, m.encode_chunked(ctxs, data)).tobytes() == data
    assert m.decode_chunked(ctxs, m.encode_chunked(ctxs, b"")).tobytes() == b""
    bad = bytearray(frame)
    bad[24 + 8 * n_chunks + 40] ^= 0x55
    with pytest.raises(m.LzfseError):
        m.decode_chunked(ctxs, bytes(bad))


@pytest.mark.gpu
def test_chunked_over_every_visible_device(snappy_raw):
    """lzfse_mi_encode_chunked / _decode_chunked over one context per VISIBLE device (chunk c on device c mod n): the frame is the
    one a single context makes, whichever devices decode it. Skips on a one-GPU box; the first multi-GPU run exercises it."""
    import lzfse_rust_amd as m
    n = m.device_count()
    if n < 2:
        pytest.skip(f"{n} HIP device(s) visible: needs at least two")
    ctxs = [m.Context(k) for k in range(n)]
    data = (snappy_raw["lcet10.txt"] + snappy_raw["kppkn.gtb"]) * (2 * n) + b"end"
    chunk = 1 << 19
    frame = m.encode_chunked(ctxs, data, chunk).tobytes()
    assert frame == m.encode_chunked(ctxs[:1], data, chunk).tobytes()
    assert m.decode_chunked(ctxs, frame).tobytes() == data
    assert m.decode_chunked(ctxs[::-1], frame).tobytes() == data
    assert m.decode_chunked(ctxs[-1:], frame).tobytes() == data


def test_device_count_needs_no_device():
    import lzfse_rust_amd as m
    assert m.device_count() >= 0


@pytest.mark.gpu
def test_cli_like_lzfoo(tmp_path, snappy_raw, oracle):
    raw = snappy_raw["alice29.txt"] * 40
    src, enc, dec, plain = (tmp_path / n for n in ("in", "enc", "dec", "plain"))
    src.write_bytes(raw)
    env = dict(os.environ, PYTHONPATH=ROOT)
    run = lambda *a, **k: subprocess.run([sys.executable, "-m", "lzfse_rust_amd.cli", *a], env=env, capture_output=True, **k)
    r = run("-encode", "-i", str(src), "-o", str(enc), "-v")
    assert r.returncode == 0, r.stderr
    assert b"LZFSE encode" in r.stderr and b"Compression ratio:" in r.stderr and b"MB/s" in r.stderr
    assert enc.read_bytes()[:4] == b"LZMC" and enc.stat().st_size < len(raw) // 2
    r = run("-decode", "-i", str(enc), "-o", str(dec))
    assert r.returncode == 0 and dec.read_bytes() == raw and r.stderr == b""
    # stdin -> stdout, one ordinary LZFSE stream: lzfoo's own bytes (lzfoo/main.rs:89: LzfseRingEncoder::encode)
    r = run("-encode", "--plain", input=raw)
    assert r.returncode == 0 and r.stdout[:4] == b"bvx2" and r.stdout[-4:] == b"bvx$"
    assert r.stdout == oracle.ring_encode(raw)
    piped = r.stdout
    r = run("-encode", "--plain", "-i", str(src), "-o", str(plain), "-v")
    assert r.returncode == 0 and plain.read_bytes() == oracle.ring_encode(raw) and f"Input size: {len(raw)} B".encode() in r.stderr
    r = run("-decode", "-i", str(plain), "-o", str(dec), "-v")
    assert r.returncode == 0 and dec.read_bytes() == raw and f"Output size: {len(raw)} B".encode() in r.stderr
    r2 = run("decode", input=piped)
    assert r2.returncode == 0 and hashlib.sha256(r2.stdout).digest() == hashlib.sha256(raw).digest()
    r = run("-decode", input=b"bvx2garbage")
    assert r.returncode == 1 and b"Error: Decode" in r.stderr
