"""Pins the CPU oracle against every fixture / known-answer test the reference holds for
the slice path. All CPU, no GPU."""
import glob
import gzip
import hashlib
import json
import os

import numpy as np
import pytest

from oracle_py import OracleError, rng_gen_vec, seq_masked

EOS = bytes([0x62, 0x76, 0x78, 0x24])


def raw_block(payload):
    return bytes([0x62, 0x76, 0x78, 0x2D]) + len(payload).to_bytes(4, "little") + payload + EOS


# ---- encoder known-answer tests: src/encode/frontend_bytes.rs:455-546, encoder.rs:35-47 ----

def test_kat_test_string(oracle):
    exp = bytes([0x62, 0x76, 0x78, 0x2d, 0x04, 0, 0, 0, 0x74, 0x65, 0x73, 0x74, 0x62, 0x76, 0x78, 0x24])
    assert oracle.encode(b"test") == exp
    assert oracle.decode(exp) == b"test"  # src/decode/mod.rs:33-47


@pytest.mark.parametrize("n", [0, 1, 20])
def test_kat_zero_raw(oracle, n):
    assert oracle.encode(bytes(n)) == raw_block(bytes(n))


def test_kat_zero_21(oracle):
    exp = bytes([0x62, 0x76, 0x78, 0x6E, 0x15, 0, 0, 0, 0x0C, 0, 0, 0, 0x68, 0x01, 0x00, 0xFC,
                 0x06, 0, 0, 0, 0, 0, 0, 0]) + EOS
    assert oracle.encode(bytes(21)) == exp


def test_kat_zero_4096(oracle):
    exp = (bytes([0x62, 0x76, 0x78, 0x6E, 0x00, 0x10, 0, 0, 0x2B, 0, 0, 0, 0x68, 0x01, 0x00])
           + bytes([0xF0, 0xFF]) * 15 + bytes([0xF0, 0x06, 0x06, 0, 0, 0, 0, 0, 0, 0]) + EOS)
    assert oracle.encode(bytes(4096)) == exp


ZERO_4097 = bytes([
    0x62, 0x76, 0x78, 0x32, 0x01, 0x10, 0x00, 0x00, 0x04, 0x00, 0x00, 0x00, 0x00, 0x02,
    0x00, 0x70, 0x00, 0x00, 0x00, 0x00, 0x00, 0x0C, 0x00, 0x10, 0x83, 0x00, 0x00, 0x00,
    0x20, 0x00, 0x00, 0x08, 0x8F, 0xC0, 0x23, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00,
    0x00, 0x00, 0xC0, 0xA3, 0xF0, 0x68, 0x3C, 0x1A]) + bytes(15) + bytes([0xF0, 0xE8, 0x03]) \
    + bytes(71) + bytes([0x22, 0xCB, 0xFF, 0x01]) + EOS


def test_kat_zero_4097(oracle):
    """The only reference KAT through hash table -> LMD -> weights -> FSE -> header."""
    assert len(ZERO_4097) == 147
    assert oracle.encode(bytes(4097)) == ZERO_4097


def test_kat_rand_cutoff_magics(oracle):
    assert oracle.encode(rng_gen_vec(0, 4096))[:4] == b"bvx-"
    assert oracle.encode(rng_gen_vec(0, 4097))[:4] == b"bvx2"


def test_parse_kats(oracle):
    """frontend_bytes.rs:549-624 restated for the Fse match unit (MATCH_UNIT 4): n zeros
    => one literal, then one match (idx 1, len n-1, D 1)."""
    n = 5000
    _enc, matches, blocks, packs = oracle.encode_trace(bytes(n))
    assert matches == [(0, 1, n - 1, 1)]
    assert blocks == [(3, 1, n)]
    assert packs == [(1, 2359, 1), (0, 2359, 0), (0, n - 1 - 2 * 2359, 0)]


# ---- decoder fixtures: test/src/data.rs:33-100 ----

def _fixtures(golden):
    fs = []
    for sub in ("snappy", "special", "mutate"):
        fs += sorted(glob.glob(os.path.join(golden, sub, "*.lzfse")))
    return [f for f in fs if not f.endswith("null.vx2.lzfse")]


def test_decode_fixture_hashes(oracle, golden_dir):
    fs = _fixtures(golden_dir)
    assert len(fs) == 12 + 2 + 4
    for f in fs:
        raw = oracle.decode(open(f, "rb").read())
        assert hashlib.sha256(raw).digest() == open(f[:-6] + ".hash", "rb").read(), f


def test_null_vx2_is_rejected(oracle, golden_dir):
    # unused by the reference's tests; zero weight bytes => Weights::load_v2 underflows
    with pytest.raises(OracleError) as e:
        oracle.decode(open(os.path.join(golden_dir, "special", "null.vx2.lzfse"), "rb").read(), cap=1 << 20)
    assert e.value.status == 30


def _parse_lmd_text(text):
    out, cur = [], None
    for ln in text.split():
        k, v = ln.split(":")
        v = int(v)
        if k == "L":
            if cur:
                out.append(tuple(cur))
            cur = [v, 0, None]
        elif k == "M":
            cur[1] = v
        else:
            cur[2] = v
    out.append(tuple(cur))
    return out


def test_decode_lmd_golden_streams(oracle, golden_dir):
    """data/snappy/lmdy_output/*.lmd: L always, M and D only when M != 0, D substituted."""
    fs = sorted(glob.glob(os.path.join(golden_dir, "lmd", "*.lmd.gz")))
    assert len(fs) == 12
    for f in fs:
        name = os.path.basename(f)[:-7]
        exp = _parse_lmd_text(gzip.open(f, "rt").read())
        _raw, lmds = oracle.decode_lmds(open(os.path.join(golden_dir, "snappy", name + ".lzfse"), "rb").read())
        got = [(l, m, d if m else None) for (l, m, d) in lmds]
        assert got == exp, name


def test_synth_fixtures_decode_and_roundtrip(oracle, golden_dir):
    fs = sorted(glob.glob(os.path.join(golden_dir, "synth", "*.lzfse")))
    assert len(fs) == 50
    for f in fs:
        raw = oracle.decode(open(f, "rb").read())
        assert 60000 < len(raw) <= 1 << 20
        assert oracle.decode(oracle.encode(raw)) == raw


def test_encode_cross_check_b3(oracle, golden_dir, snappy_raw):
    """Two independent restatements of A.1 (this C oracle, SURVEY's session probe) agree."""
    b3 = json.load(open(os.path.join(golden_dir, "encoder_b3.json")))
    for name, raw in snappy_raw.items():
        enc = oracle.encode(raw)
        size, sha = b3[name]
        assert len(enc) == size, name
        assert hashlib.sha256(enc).hexdigest() == sha, name
        assert oracle.decode(enc) == raw


# ---- robustness: test/src/mutate_0.rs ----

@pytest.mark.parametrize("name", ["raw", "vx1", "vx2", "vxn"])
def test_mutate_bits_never_crash(oracle, golden_dir, name):
    data = bytearray(open(os.path.join(golden_dir, "mutate", name + ".lzfse"), "rb").read())
    n_ok = 0
    for i in range(len(data)):
        for b in range(8):
            data[i] ^= 1 << b
            st = oracle.decode_status(bytes(data), 1 << 16)
            n_ok += st == 0
            data[i] ^= 1 << b
    assert n_ok < len(data) * 8


def test_truncation_and_trailing(oracle, snappy_raw):
    enc = oracle.encode(snappy_raw["html"])
    for cut in (1, 3, 4, 5, 100, len(enc) - 40):
        assert oracle.decode_status(enc[:-cut], 1 << 20) != 0
    assert oracle.decode_status(enc + b"\0", 1 << 20) == 7  # PayloadOverflow decoder.rs:93-95
    assert oracle.decode_status(b"abcd" + enc, 1 << 20) == 2  # BadBlock


def test_n_raw_bytes_off_by_one(oracle, snappy_raw):
    """fse/test.rs:434,458: n_raw_bytes +-1 => BadLmdPayload."""
    enc = bytearray(oracle.encode(snappy_raw["html"]))
    n = int.from_bytes(enc[4:8], "little")
    for d in (-1, 1):
        enc[4:8] = (n + d).to_bytes(4, "little")
        assert oracle.decode_status(bytes(enc), 1 << 20) == 22


# ---- round trips on the reference's synthetic generators (test/src/random_0.rs, pattern_*) ----

@pytest.mark.parametrize("seed", [0, 1, 2])
def test_roundtrip_low_entropy_noise(oracle, seed):
    raw = seq_masked(seed, 0x01010101, 1 << 20)
    assert oracle.decode(oracle.encode(raw)) == raw


@pytest.mark.parametrize("n", [4097, 4098, 5000, 39999, 40000, 40001, 65536, 100000])
def test_roundtrip_sizes_random(oracle, n):
    raw = rng_gen_vec(n, n)
    assert oracle.decode(oracle.encode(raw)) == raw


@pytest.mark.parametrize("n", list(range(0, 64)) + [4095, 4096, 4097])
def test_roundtrip_small(oracle, n):
    raw = bytes((i * 7 + (i >> 3)) & 0xFF for i in range(n))
    enc = oracle.encode(raw)
    assert oracle.decode(enc) == raw
    assert oracle.decode_size(enc) == n


def test_roundtrip_block_boundaries(oracle):
    """> 10 000 LMDs and > 40 000 literals force mid-stream bvx2 cuts (fse/backend.rs:76-90)."""
    rng = np.random.default_rng(5)
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(3, 9)), dtype=np.uint8)) for _ in range(3000)]
    text = b" ".join(words[int(i)] for i in rng.integers(0, 3000, size=200000))
    enc, _m, blocks, _p = oracle.encode_trace(text)
    assert len(blocks) > 3 and any(b[0] == 10000 for b in blocks)
    assert oracle.decode(enc) == text
    noise = rng.integers(0, 256, size=150000, dtype=np.uint8).tobytes()
    enc, _m, blocks, _p = oracle.encode_trace(noise)
    assert any(b[1] == 40000 for b in blocks)
    assert oracle.decode(enc) == noise


def test_long_match_split(oracle):
    raw = bytes(300000)  # one match of 299 999 => ceil(/2359) packs
    enc, matches, blocks, packs = oracle.encode_trace(raw)
    assert matches == [(0, 1, 299999, 1)]
    assert sum(p[1] for p in packs) == 299999 and max(p[1] for p in packs) == 2359
    assert oracle.decode(enc) == raw


# ---- low-level known answers ----

def test_weight_code_roundtrip(oracle):
    """fse/weight_encoder.rs:44-51 for every encodable weight 0..1047 (table sums ignored)."""
    import ctypes as C
    for base in range(0, 1048, 4):
        w = np.zeros(360, dtype=np.uint16)
        w[104:108] = np.minimum(np.arange(base, base + 4), 1047)  # U table: allowed sum 1024
        if int(w.sum()) > 1024:
            w[105:108] = 0
        buf = np.zeros(640, dtype=np.uint8)
        n = oracle.lib.lzo_weights_store_v2(w.ctypes.data, buf.ctypes.data)
        out = np.zeros(360, dtype=np.uint16)
        st = oracle.lib.lzo_weights_load_v2(buf.ctypes.data, n, out.ctypes.data)
        if int(w.sum()) <= 1024:
            assert st == 0 and (out == w).all()
        else:
            assert st == 27


def test_normalize_m1_invariants(oracle):
    """fse/weights.rs:366-385: sums exactly to the state count, non-zero stays non-zero."""
    rng = np.random.default_rng(1)
    for n_sym, states in ((20, 64), (64, 256), (256, 1024)):
        for _ in range(200):
            w = rng.integers(0, 50, size=n_sym).astype(np.uint16)
            w[rng.integers(0, n_sym, size=n_sym // 2)] = 0
            if w.sum() == 0:
                continue
            nz = w != 0
            oracle.lib.lzo_normalize_m1(w.ctypes.data, n_sym, int(w.sum()), states)
            assert int(w.sum()) == states
            assert ((w != 0) == nz).all()


# ---- reposition (frontend_bytes.rs:348-375): slices that the front end matches in several blocks ----
G_SMALL, S_SMALL = 0x100000, 0x20000     # BLOCK_GUIDE / SLACK of the test hook: 1 MiB / 128 KiB (the limit lies 917 501 into a block)


def test_reposition_one_block_below_the_guide(oracle, snappy_raw):
    """A slice of up to BLOCK_GUIDE + 3 bytes is ONE block whatever the guide (:169-178): lzo_encode_guide == lzo_encode there, and
    lzo_encode itself is the guide of the reference (0x7FFF_FFFF / 0x1000_0000)."""
    text = snappy_raw["lcet10.txt"] + snappy_raw["html"]
    for n in (4097, 100000, len(text)):
        assert oracle.encode_guide(text[:n], G_SMALL, S_SMALL) == oracle.encode(text[:n])
    big = text * 9                                    # 4.8 MB: five blocks at the small guide, one at the reference's
    assert oracle.encode_guide(big, 0x7FFFFFFF, 0x10000000) == oracle.encode(big)
    assert oracle.encode_guide(big[:G_SMALL + 3], G_SMALL, S_SMALL) == oracle.encode(big[:G_SMALL + 3])


def test_reposition_zeros_by_hand(oracle):
    """Zeros of BLOCK_GUIDE + 4 bytes, the first size that repositions: position 1 matches position 0 to the END OF THE BLOCK
    (BLOCK_GUIDE - 1 bytes, :253), the walk stands at BLOCK_GUIDE, the slice moves up to MAX_MATCH_DISTANCE below it, and the
    second block -- short: 4 + 262 139 bytes -- visits ONE position, whose match of 4 bytes is flushed as pending (:271-285).
    One block of G + 4 zeros would be one match of G + 3 bytes: the LMD streams differ in their last two entries."""
    n = G_SMALL + 4
    got = oracle.decode_lmds(oracle.encode_guide(bytes(n), G_SMALL, S_SMALL))[1]
    one = oracle.decode_lmds(oracle.encode(bytes(n)))[1]
    total = lambda lmds: sum(int(l) + int(m) for l, m, _ in lmds)
    assert total(got) == n and total(one) == n
    flat = lambda lmds: [(int(l), int(m), int(d)) for l, m, d in lmds]
    g, o1 = flat(got), flat(one)
    k = (G_SMALL - 1) // 2359
    assert g[0] == (1, 2359, 1) and g[:k] == o1[:k]
    assert g[k:] == [(0, (G_SMALL - 1) - 2359 * k, 1), (0, 4, 1)]           # the rest of the first block's match, then the 4-byte match
    assert o1[k:] == [(0, (G_SMALL + 3) - 2359 * k, 1)]


@pytest.mark.parametrize("kind", ["text", "zeros", "noise", "masked", "desert", "period", "runs"])
def test_reposition_round_trips(oracle, snappy_raw, kind):
    """Five blocks and more at the small guide: the stream decodes back (the decoder is pinned by the reference's fixtures), for
    matches that run past a block's limit or to its end (zeros, periods, runs), literal deserts that pass the next block's
    head -- pushed as they are, the pending match dropped (:361-367) -- and plain text."""
    from oracle_py import rng_gen_vec, seq_masked
    rng = np.random.default_rng(5)
    text = (snappy_raw["lcet10.txt"] + snappy_raw["alice29.txt"] + snappy_raw["urls.10K"]) * 4
    data = {
        "text": text,
        "zeros": bytes(3_500_000),
        "noise": rng_gen_vec(3, 3_000_000),
        "masked": seq_masked(2, 0x03030303, 3_000_000),
        "desert": rng_gen_vec(4, 1_500_000) + text[:2_000_000] + rng_gen_vec(5, 900_000) + bytes(700_000),
        "period": rng_gen_vec(6, 70_000) * 60,
        "runs": b"".join(bytes([int(rng.integers(0, 256))]) * int(rng.integers(2000, 90000)) for _ in range(80)),
    }[kind]
    enc = oracle.encode_guide(data, G_SMALL, S_SMALL)
    assert oracle.decode(enc) == data
    if kind in ("zeros", "period", "noise", "desert"):
        assert enc != oracle.encode(data)      # (a match across a block's limit, or a literal desert: not what one block makes)


def test_reposition_guide_conditions(oracle):
    """The reference's own assertions on the two constants (:166-168,359): lzo_encode_guide refuses what they forbid."""
    from oracle_py import OracleError
    for g, s in ((0x100000, 255), (0x100000, 0x80001), (0x40000, 0x100), (0x80000000, 0x10000000)):
        with pytest.raises(OracleError):
            oracle.encode_guide(bytes(5000), g, s)
