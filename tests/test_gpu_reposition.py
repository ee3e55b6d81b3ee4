"""E20: slices that the reference's front end matches in several blocks (frontend_bytes.rs:160-211 match_any, :348-375 reposition; a
slice of more than BLOCK_GUIDE + 3 = 0x8000_0002 bytes). The device takes a block per call and lzfse_mi_encode carries the walk's
state, the positions the reference never pushed and the unclosed bvx2 block between the calls (encode_slice_blocks, api.hip). With
the reference's constants that needs inputs beyond 2 GiB (tests/test_gpu_big.py has one); here the diagnostic build is given a
BLOCK_GUIDE of 1 MiB and a SLACK of 128 KiB (LZFSE_MI_OPT_DIAG_GUIDE), and the oracle the same (lzo_encode_guide), so that inputs of
a few MiB reposition several times: text, zeros and periods (matches that run past a block's limit, or to its end), noise (literal
deserts that pass the next block's head: pushed as they are, the pending match dropped), block-limit edges."""
import numpy as np
import pytest

from oracle_py import rng_gen_vec, seq_masked

pytestmark = pytest.mark.gpu
G, S = 0x100000, 0x20000


@pytest.fixture()
def guided(diag_ctx):
    diag_ctx.set_option("diag_guide", G | (S << 32))
    yield diag_ctx
    diag_ctx.set_option("diag_guide", 0)


def _encode(ctx, data):
    import lzfse_rust_amd as m
    out = bytearray()
    m.LzfseEncoder(context=ctx).encode_bytes(data, out)
    return bytes(out)


def test_reposition_matches_oracle(guided, oracle, snappy_raw):
    text = (snappy_raw["lcet10.txt"] + snappy_raw["alice29.txt"] + snappy_raw["urls.10K"]) * 4   # 5 MB: five blocks
    rng = np.random.default_rng(8)
    cases = {
        "text": text,
        "zeros": bytes(3_500_000),
        "noise": rng_gen_vec(3, 3_000_000),
        "masked": seq_masked(2, 0x03030303, 3_000_000),
        "desert": rng_gen_vec(4, 1_500_000) + text[:2_000_000] + rng_gen_vec(5, 900_000) + bytes(700_000),
        "period": rng_gen_vec(6, 70_000) * 60,
        "html": snappy_raw["html"] * 37,
        "runs": b"".join(bytes([int(rng.integers(0, 256))]) * int(rng.integers(2000, 90000)) for _ in range(80)),
        "dense": (np.tile(rng.integers(0, 256, size=8, dtype=np.uint8), 400_000) ^ (rng.random(3_200_000) < 0.1).astype(np.uint8)).tobytes(),
    }
    for k in (2, 3, 4, 5, 1000):                      # one block up to G + 3 bytes, two beyond
        cases[f"zeros_edge_{k}"] = bytes(G + k)
        cases[f"text_edge_{k}"] = text[:G + k]
    lim = G - S - 3
    for k in (-2, 0, 1, 2):                          # a match that ends around the first block's limit
        cases[f"limit_{k}"] = rng_gen_vec(9, lim + k - 5000) + bytes(5000) + text[:1_200_000]
    for name, data in cases.items():
        want = oracle.encode_guide(data, G, S)
        got = _encode(guided, data)
        assert got == want, (name, len(data), len(got), len(want))
        assert oracle.decode(got) == data, name
    # and with the reference's own constants these are single blocks: the ordinary path
    guided.set_option("diag_guide", 0)
    assert _encode(guided, cases["text"]) == oracle.encode(cases["text"])


def test_reposition_fuzz(guided, oracle, snappy_raw):
    rng = np.random.default_rng(31)
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(1, 12)), dtype=np.uint8)) for _ in range(500)]

    def gen(kind, n):
        if kind == 0:
            return rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        if kind == 1:
            return (rng.integers(0, int(rng.integers(2, 9)), size=n, dtype=np.uint8) * 31).astype(np.uint8).tobytes()
        if kind == 2:
            per = int(rng.integers(1, 400000))
            return (bytes(rng.integers(0, 256, size=per, dtype=np.uint8)) * (n // per + 1))[:n]
        if kind == 3:
            out = bytearray()
            while len(out) < n:
                out += words[int(rng.integers(0, 500))] + b" "
            return bytes(out[:n])
        if kind == 4:
            return bytes(n)
        parts = bytearray()
        while len(parts) < n:
            parts += gen(int(rng.integers(0, 5)), int(rng.integers(1000, 700000)))
        return bytes(parts[:n])

    for it in range(24):
        data = gen(int(rng.integers(0, 6)), int(rng.integers(G + 4, 4 * G)))
        assert _encode(guided, data) == oracle.encode_guide(data, G, S), (it, len(data))
