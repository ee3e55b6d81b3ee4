"""Boundary behaviour of the C ABI on the GPU: argument checking, capacities, context independence and reuse
(`&mut self` discipline of LzfseEncoder / LzfseDecoder: encoder.rs:14-18, decoder.rs:17-21 -- a context is not thread-safe,
distinct contexts are independent, results never depend on earlier calls)."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_argument_checking_and_empty_calls(oracle):
    import lzfse_rust_amd as m
    from lzfse_rust_amd import _native
    L = _native.lib()
    c = m.Context(0)
    n = C.c_size_t(0)
    buf = np.zeros(64, dtype=np.uint8)
    assert L.lzfse_mi_encode(None, buf.ctypes.data, 10, buf.ctypes.data, 64, C.byref(n)) == 11       # BAD_ARGUMENT
    assert L.lzfse_mi_encode(c._h, buf.ctypes.data, 10, buf.ctypes.data, 64, None) == 11
    assert L.lzfse_mi_decode_batch(c._h, 0, None, None, None, None, None, None) == 0                    # nothing to do
    assert L.lzfse_mi_encode_batch_device(c._h, 0, None, None, None, None, None, None, None, None) == 0
    h = C.c_void_p()
    assert L.lzfse_mi_create(10 ** 6, C.byref(h)) == 10 and not h.value                                 # NO_DEVICE
    assert L.lzfse_mi_create(-1, C.byref(h)) == 10
    assert L.lzfse_mi_set_option(c._h, 12345, 0) == 11
    assert L.lzfse_mi_set_option(c._h, 1, 99) == 11
    # an empty input is a stream of its own: an empty raw block + EOS (frontend_bytes.rs:63-77 and its KAT :455-462)
    outs, st = c.encode_batch([b""])
    assert st[0] == 0 and outs[0].tobytes() == oracle.encode(b"") == b"bvx-" + bytes(4) + b"bvx$"
    back, st = c.decode_batch([outs[0].tobytes(), b"bvx$"])
    assert list(st) == [0, 0] and len(back[0]) == 0 and len(back[1]) == 0


def test_capacities(oracle, snappy_raw):
    import lzfse_rust_amd as m
    from lzfse_rust_amd import _native
    L = _native.lib()
    c = m.Context(0)
    raw = snappy_raw["html"]
    want = oracle.encode(raw)
    src = np.frombuffer(raw, dtype=np.uint8)
    n = C.c_size_t(0)
    for cap, expect in ((len(want), 0), (len(want) - 1, 6), (16, 6), (0, 6)):
        dst = np.zeros(max(cap, 1), dtype=np.uint8)
        st = L.lzfse_mi_encode(c._h, src.ctypes.data, src.size, dst.ctypes.data, cap, C.byref(n))
        assert st == expect, (cap, st)
        if st == 0:
            assert n.value == len(want) and dst[:n.value].tobytes() == want
    enc = np.frombuffer(want, dtype=np.uint8)
    for cap, expect in ((len(raw), 0), (len(raw) + 100, 0), (len(raw) - 1, 6)):
        dst = np.zeros(cap, dtype=np.uint8)
        st = L.lzfse_mi_decode(c._h, enc.ctypes.data, enc.size, dst.ctypes.data, cap, C.byref(n))
        assert st == expect, (cap, st)
        if st == 0:
            assert n.value == len(raw) and dst[:n.value].tobytes() == raw
    # small size classes obey the capacity as well
    outs, st = c._host_batch(L.lzfse_mi_encode_batch, [raw[:100], raw[:100]], [200, 20])
    assert list(st) == [0, 6] and outs[0].tobytes() == oracle.encode(raw[:100])


def test_contexts_are_independent_across_threads(oracle, snappy_raw):
    import lzfse_rust_amd as m
    names = list(snappy_raw)
    want = {n: oracle.encode(snappy_raw[n]) for n in names}
    errors = []

    def worker(k):
        try:
            c = m.Context(0)
            for rep in range(3):
                order = names[k:] + names[:k]
                encs, st = c.encode_batch([snappy_raw[n] for n in order] * (4 + k))
                assert all(s == 0 for s in st)
                for n, e in zip(order * (4 + k), encs):
                    assert e.tobytes() == want[n], n
                decs, st = c.decode_batch([want[n] for n in order])
                assert all(s == 0 for s in st)
                for n, d in zip(order, decs):
                    assert d.tobytes() == snappy_raw[n], n
            c.close()
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))
    ts = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_context_reuse_and_recreation(oracle, snappy_raw):
    """Scratch grows and is reused; results never depend on earlier calls (the reference resets its table per call,
    frontend_bytes.rs:113-119); contexts can be created and destroyed repeatedly."""
    import lzfse_rust_amd as m
    raw_small, raw_big = snappy_raw["html"], snappy_raw["urls.10K"] * 9
    w_small, w_big = oracle.encode(raw_small), oracle.encode(raw_big)
    for _ in range(6):
        c = m.Context(0)
        for raws, wants in (([raw_small], [w_small]), ([raw_big] * 12, [w_big] * 12), ([raw_small] * 3, [w_small] * 3),
                            ([b"x" * 5000, raw_small, b""], [oracle.encode(b"x" * 5000), w_small, oracle.encode(b"")])):
            encs, st = c.encode_batch(raws)
            assert all(s == 0 for s in st)
            assert [e.tobytes() for e in encs] == wants
            decs, st = c.decode_batch(wants)
            assert all(s == 0 for s in st) and [d.tobytes() for d in decs] == raws
        c.close()
