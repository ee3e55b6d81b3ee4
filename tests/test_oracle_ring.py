"""Pins the ring / stream half of the CPU oracle (oracle/lzfse_oracle.c, "ring frontend": LzfseRingEncoder::encode,
LzfseWriter, encode/frontend_ring.rs) against every known-answer test the reference holds for it
(/root/reference/src/encode/frontend_ring.rs:768-993) and against the facts that follow from its code: the output does
not depend on how the input is cut into writes, it decodes back, and below one ring it is the slice encoder's parse.
All CPU, no GPU."""
import numpy as np
import pytest

from oracle_py import rng_gen_vec
from test_oracle import EOS, ZERO_4097, raw_block
import test_kit as tk

T_RING = (0x10000, 0x200, 0x100)   # the KATs' test ring: frontend_ring.rs:704-717
RING, BLK = 0x80000, 0x4000        # encode/constants.rs:23-33


# ---- byte-exact vectors: frontend_ring.rs:768-844 (the same bytes as the slice path's, frontend_bytes.rs:455-531) ----

@pytest.mark.parametrize("n", [0, 1, 20])
def test_kat_ring_zero_raw(oracle, n):
    assert oracle.ring_encode(bytes(n)) == raw_block(bytes(n))


def test_kat_ring_zero_21(oracle):
    exp = bytes([0x62, 0x76, 0x78, 0x6E, 0x15, 0, 0, 0, 0x0C, 0, 0, 0, 0x68, 0x01, 0x00, 0xFC,
                 0x06, 0, 0, 0, 0, 0, 0, 0]) + EOS
    assert oracle.ring_encode(bytes(21)) == exp


def test_kat_ring_zero_4096(oracle):
    exp = (bytes([0x62, 0x76, 0x78, 0x6E, 0x00, 0x10, 0, 0, 0x2B, 0, 0, 0, 0x68, 0x01, 0x00])
           + bytes([0xF0, 0xFF]) * 15 + bytes([0xF0, 0x06, 0x06, 0, 0, 0, 0, 0, 0, 0]) + EOS)
    assert oracle.ring_encode(bytes(4096)) == exp


def test_kat_ring_zero_4097(oracle):
    assert oracle.ring_encode(bytes(4097)) == ZERO_4097


def test_kat_ring_rand_cutoff_magics(oracle):
    """frontend_ring.rs:846-859"""
    assert oracle.ring_encode(rng_gen_vec(0, 4096))[:4] == b"bvx-"
    assert oracle.ring_encode(rng_gen_vec(0, 4097))[:4] == b"bvx2"


# ---- parse KATs on the Dummy backend and the 64 KiB test ring: frontend_ring.rs:861-992 ----

def test_kat_match_short_zero_4(oracle):
    """:861-887: literals [0], lmds [(1, 3, 1)]"""
    assert [t[1:] for t in oracle.ring_kat(0, T_RING, bytes(T_RING[0]), 0, 4)] == [(1, 3, 1)]


def test_kat_match_short_zero_n(oracle):
    """:889-917 (ignored as expensive in the reference; every n here)"""
    z = bytes(T_RING[0])
    for n in range(5, T_RING[0]):
        got = oracle.ring_kat(0, T_RING, z, 0, n)
        assert got == [(0, 1, n - 1, 1)], n


def test_kat_match_long_overmatch_limit(oracle):
    """:919-950: one LMD, no literals, distance 1, and the match never passes the tail"""
    z = bytes(T_RING[0])
    for offset in range(T_RING[1] - 1):
        idx = T_RING[0] // 2 + offset
        got = oracle.ring_kat(1, T_RING, z, idx, 0)
        assert len(got) == 1, offset
        lit_pos, l, m, d = got[0]
        assert l == 0 and d == 1 and idx + m <= T_RING[0], (offset, got)


def test_kat_sandwich_n_short(oracle):
    """:952-992: 1 2 3 0 ... 0 1 2 3 -> literals [1, 2, 3, 0], lmds [(4, n - 7, 1), (0, 3, n - 3)]"""
    for n in list(range(10, 600)) + list(range(600, T_RING[0], 97)) + [T_RING[0] - 1]:
        ring = bytearray(T_RING[0])
        ring[0:3] = b"\x01\x02\x03"
        ring[n - 3:n] = b"\x01\x02\x03"
        got = oracle.ring_kat(0, T_RING, bytes(ring), 0, n)
        assert [t[1:] for t in got] == [(4, n - 7, 1), (0, 3, n - 3)], n
        assert got[0][0] == 0 and got[1][0] == n - 3


# ---- properties that follow from the reference's code ----

def _inputs():
    rng = np.random.default_rng(7)
    words = [bytes(rng.integers(97, 123, size=int(k), dtype=np.uint8)) for k in rng.integers(2, 9, size=500)]
    text = b" ".join(words[int(i)] for i in rng.integers(0, 500, size=400000))
    yield "text_1_9M", text[:1_900_000]
    yield "text_ring_exact", text[:RING]
    yield "text_ring_plus_blk", text[:RING + BLK]
    yield "text_ring_minus_1", text[:RING - 1]
    yield "noise_1_2M", tk.seq(1_200_000, seed=3)
    yield "low_entropy_1M", tk.seq(1_000_000, seed=5, mask=0x01010101)
    yield "zeros_1_5M", bytes(1_500_000)
    yield "zeros_then_text", bytes(700_000) + text[:300_000] + bytes(400_000)
    # a literal desert of more than half a ring between two copies: push_literal_overflow (frontend_ring.rs:257-272)
    yield "noise_sandwich", text[:40_000] + tk.seq(900_000, seed=9) + text[:40_000]
    per = bytes(rng.integers(0, 256, size=300_000, dtype=np.uint8))
    yield "period_300k", per * 4


@pytest.mark.parametrize("name,data", list(_inputs()), ids=[n for n, _ in _inputs()])
def test_ring_stream_decodes_and_ignores_write_sizes(oracle, name, data):
    enc = oracle.ring_encode(data)
    assert oracle.decode(enc) == data
    for piece in (1 << 14, 100_003, 7777):
        assert oracle.ring_encode(data, piece) == enc, piece
    rng = np.random.default_rng(len(data))
    cuts = np.sort(rng.integers(0, len(data), size=40))
    pieces = [data[a:b] for a, b in zip([0] + list(cuts), list(cuts) + [len(data)])]
    assert oracle.ring_encode_pieces(pieces) == enc


def test_ring_equals_slice_below_one_ring(oracle, snappy_raw):
    """Below RING_SIZE nothing is committed before flush (frontend_ring.rs:216-219), flush_select runs match_short over
    the whole input with head = 0 (:297-312,:401-450), and match_short differs from the slice loop
    (frontend_bytes.rs:160-268) only in the coarse compare, which reads past the end of the input: the parse is the
    slice parse unless two candidates both run to the end of the input. Every Snappy file below one ring agrees."""
    n_same = 0
    for name, raw in snappy_raw.items():
        if len(raw) < RING:
            assert oracle.ring_encode(raw) == oracle.encode(raw), name
            n_same += 1
    assert n_same >= 8


def test_ring_differs_from_slice_beyond_one_ring(oracle, snappy_raw):
    """... and beyond one ring it is another parse (the reason this oracle exists): literals that pass the ring head
    are pushed in 16 KiB pieces, so an incompressible input cuts its LMDs differently."""
    data = tk.seq(1_200_000, seed=3)
    ring, _m, _b, rp = oracle.ring_encode_trace(data)
    sl, _m2, _b2, sp = oracle.encode_trace(data)
    assert ring != sl and rp != sp
    assert oracle.decode(ring) == data


def test_ring_small_sizes_round_trip(oracle):
    rng = np.random.default_rng(11)
    for n in list(range(0, 70)) + [255, 256, 4095, 4096, 4097, 4098, 5000, 16383, 16384, 16385, 70000]:
        data = bytes(rng.integers(0, 4, size=n, dtype=np.uint8))
        enc = oracle.ring_encode(data)
        assert oracle.decode(enc) == data, n
        assert oracle.ring_encode(data, 3) == enc, n
