"""Restatement of the reference's test_kit generators (test infrastructure; /root/reference/test_kit/src/*.rs) so that its
integration suite (test/src/*.rs) can be restated against the C ABI. numpy where the sequences are long.

    Rng            rng.rs:6-18     LCG state' = state * 1103515245 + 12345 (mod 2^32); gen() returns the NEW state
    Rng.gen_vec    rng.rs:42-58    bytes of the CURRENT state first (LE), then advance
    Seq / masked   seq.rs:8-38     bytes of (rng.gen() & mask), low byte first
    Useq           useq.rs:5-42    up to 10 923 528 bytes in which every 4-byte window is unique
    Cycle          cycle.rs:4-15   1, 2, ..., 255, 0, 1, ...
    Fibonacci      fibonacci.rs    0 1 1 2 3 5 ... until u32 overflow
    build_match_inc / build_match_dec   slices.rs:2-26
"""
import numpy as np


class Rng:
    def __init__(self, seed=0):
        self.s = seed & 0xFFFFFFFF

    def gen(self):
        self.s = (self.s * 1103515245 + 12345) & 0xFFFFFFFF
        return self.s

    def gen_vec(self, length):
        n4 = length // 4
        out = np.empty(n4 + 1, dtype=np.uint32)
        for i in range(n4):
            out[i] = self.s
            self.gen()
        out[n4] = self.s
        return out.view(np.uint8)[:length].tobytes()


def lcg_states(seed, count):
    """states[k] = state after k + 1 steps from `seed` (what k + 1 calls of Rng.gen() return), vectorised by doubling."""
    out = np.empty(count, dtype=np.uint64)
    a, c = np.uint64(1103515245), np.uint64(12345)
    mask = np.uint64(0xFFFFFFFF)
    s = np.uint64(seed & 0xFFFFFFFF)
    if count == 0:
        return out.astype(np.uint32)
    out[0] = (s * a + c) & mask
    filled = 1
    # x_{n+k} = A_k x_n + C_k with (A_k, C_k) the k-fold composition
    ak, ck = a, c
    while filled < count:
        m = min(filled, count - filled)
        out[filled:filled + m] = (out[:m] * ak + ck) & mask
        filled += m
        ck = (ak * ck + ck) & mask
        ak = (ak * ak) & mask
    return out.astype(np.uint32)


def seq(length, seed=0, mask=0xFFFFFFFF):
    """Iterator::take(Seq::masked(Rng::new(seed), mask), length) as bytes (Seq::default(): seed 0, no mask)."""
    n4 = (length + 3) // 4
    st = lcg_states(seed, n4) & np.uint32(mask)
    return st.view(np.uint8)[:length].tobytes()


def useq(length):
    """Useq::default().take(length): 4-byte groups (u0, u1, u2, 0); u2 counts up, wrapping pushes u1, then u0."""
    out = bytearray()
    u = [1, 2, 3, 0]
    n = 0
    while len(out) < length:
        if n == 4:
            u[2] = (u[2] + 1) & 0xFF
            if u[2] == 0:
                u[1] += 1
                u[2] = u[1] + 1
                if u[1] == 0xFE:
                    u[0] += 1
                    if u[0] == 0xFD:
                        break
                    u[1] = u[0] + 1
                    u[2] = u[1] + 1
            n = 0
        out.append(u[n])
        n += 1
    return bytes(out)


def cycle(length):
    return (np.arange(1, length + 1, dtype=np.uint64) & 0xFF).astype(np.uint8).tobytes()


def fibonacci():
    u, v = 0, 1
    out = []
    while True:
        if u == 0 and v == 0:
            return out
        if v == 0:
            out.append(u)
            u = 0
            continue
        n, o = u, v
        v = n + o if n + o <= 0xFFFFFFFF else 0
        u = o
        out.append(n)


def build_match_inc(size, index, match_index, match_len):
    assert match_index < index <= size and match_len <= size - index
    distance = index - match_index
    assert distance <= 255
    s = bytearray(size)
    for i in range(match_len + distance):
        s[match_index + i] = (i % distance) + 1
    return bytes(s)


def build_match_dec(size, index, match_index, match_len):
    assert match_index < index <= size and match_len <= match_index
    distance = index - match_index
    assert distance <= 255
    s = bytearray(size)
    for i in range(match_len + distance):
        s[index - i - 1] = (i % distance) + 1
    return bytes(s)


def patchwork(seed, rounds, shift, base_len=0x100):
    """test/src/patchwork_0.rs / patchwork_1.rs: a 256-byte random base, then `rounds` times: copy a random earlier
    slice (off = (gen >> 16) % top, len = (gen >> shift) % (top - off)) to the end."""
    v = bytearray(seq(base_len))
    rng = Rng(seed)
    for _ in range(rounds):
        top = len(v)
        off = (rng.gen() >> 16) % top
        ln = (rng.gen() >> shift) % (top - off)
        v += v[off:off + ln]
    return bytes(v)
