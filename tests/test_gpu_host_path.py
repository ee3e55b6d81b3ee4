"""GPU tests of the host-pointer entry points (lzfse_mi_encode_batch / _decode_batch): inputs and outputs travel through pinned
staging in 16 MiB granules copied by several threads, encoded streams are packed on the device first -- none of which may
show in the results: empty and tiny streams, streams of several granules, failed streams in the middle of a batch."""
import numpy as np
import pytest

from oracle_py import rng_gen_vec

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import lzfse_rust_amd as m
    return m.Context(0)


def test_host_batch_of_mixed_sizes(ctx, oracle, snappy_raw):
    big = (snappy_raw["lcet10.txt"] * 100)[: 40 << 20]            # 2.5 granules in, 1 out
    noise = rng_gen_vec(5, 20 << 20)                                # incompressible: 20 MiB out as well
    raws = [b"", b"x", bytes(5000), big, snappy_raw["html"], noise, b"", snappy_raw["urls.10K"] * 3, bytes(range(256)) * 40]
    encs, st = ctx.encode_batch(raws)
    assert all(e == 0 for e in st)
    for r, e in zip(raws, encs):
        assert e.tobytes() == oracle.encode(r), len(r)
    streams = [e.tobytes() for e in encs]
    outs, st = ctx.decode_batch(streams)
    assert all(e == 0 for e in st) and all(o.tobytes() == r for o, r in zip(outs, raws))


def test_host_batch_with_failed_streams_in_the_middle(ctx, oracle, snappy_raw):
    import lzfse_rust_amd as m
    raws = [snappy_raw["alice29.txt"] * 20, snappy_raw["html"], snappy_raw["kppkn.gtb"] * 30, snappy_raw["geo.protodata"]]
    encs = [oracle.encode(r) for r in raws]
    cut = encs[1][: len(encs[1]) // 2]
    srcs = [encs[0], cut, encs[2], b"bvxQ", encs[3], b""]
    caps = [len(raws[0]), len(raws[1]), 1000, 10, len(raws[3]), 10]
    want = [oracle.decode_status(s, c) for s, c in zip(srcs, caps)]
    assert want[0] == 0 and want[1] != 0 and want[2] == 6 and want[3] != 0 and want[4] == 0 and want[5] != 0
    outs, st = ctx.decode_batch(srcs, caps=caps)
    assert list(st) == want
    assert outs[0].tobytes() == raws[0] and outs[4].tobytes() == raws[3] and all(len(outs[i]) == 0 for i in (1, 2, 3, 5))
    # encode: a destination that is too small for one stream of the batch (the packed copy has a hole there)
    lib = ctx._lib
    n = len(raws)
    import ctypes as C
    arrs = [np.frombuffer(r, dtype=np.uint8) for r in raws]
    capsE = [lib.lzfse_mi_encode_bound(len(r)) for r in raws]
    capsE[2] = 100
    dst = [np.empty(c, dtype=np.uint8) for c in capsE]
    sp = (C.c_void_p * n)(*[a.ctypes.data for a in arrs]); sl = (C.c_size_t * n)(*[a.size for a in arrs])
    dp = (C.c_void_p * n)(*[d.ctypes.data for d in dst]); dc = (C.c_size_t * n)(*capsE)
    ol = (C.c_size_t * n)(); stE = (C.c_int * n)()
    assert lib.lzfse_mi_encode_batch(ctx._h, n, sp, sl, dp, dc, ol, stE) == 0
    assert list(stE) == [0, 0, 6, 0] and ol[2] == 0
    for i in (0, 1, 3):
        assert dst[i][: ol[i]].tobytes() == encs[i]


def test_host_batch_cut_in_two_halves(ctx, oracle, snappy_raw):
    """A call of 512 MiB and more (inputs + capacities) runs as two halves on two contexts, one step apart; the results
    and the error details are those of one call, in the caller's order."""
    import lzfse_rust_amd as m
    big = (snappy_raw["lcet10.txt"] * 40)[: 12 << 20]
    raws = [big[i:] + big[:i] for i in range(0, 24 * 997, 997)]       # 24 different streams of 12 MiB
    encs, st = ctx.encode_batch(raws)                                   # 288 MiB in + 288 MiB of capacity
    assert all(e == 0 for e in st)
    want = [oracle.encode(r) for r in raws[:2]] + [None] * 21 + [oracle.encode(raws[-1])]
    for e, w in zip(encs, want):
        if w is not None:
            assert e.tobytes() == w
    streams = [e.tobytes() for e in encs]
    streams[3] = b"bvxQ" + streams[3][4:]           # first half: BadBlock with its magic as the detail
    streams[20] = streams[20][:-7]                   # second half: cut short
    outs, st = ctx.decode_batch(streams, caps=[2 * len(r) for r in raws])      # 90 MiB in + 576 MiB of capacity
    assert [i for i, e in enumerate(st) if e] == [3, 20] and st[3] == 2 and st[20] == oracle.decode_status(streams[20], 2 * len(raws[20]))
    assert ctx.error_detail(3) == int.from_bytes(b"bvxQ", "little")
    for i, (o, r) in enumerate(zip(outs, raws)):
        if i not in (3, 20):
            assert o.tobytes() == r, i


def test_two_contexts_in_two_threads(oracle, snappy_raw):
    """Distinct contexts are independent: two host threads, each with its own context, encode and decode
    different batches at the same time (ctypes releases the GIL during the calls)."""
    import threading
    import lzfse_rust_amd as m
    names = sorted(snappy_raw)
    batches = [[snappy_raw[n] for n in names[:6]] * 3, [snappy_raw[n] for n in names[6:]] * 3 + [bytes(3 << 20)]]
    want = [[oracle.encode(r) for r in b] for b in batches]
    errors = []

    def work(k):
        try:
            c = m.Context(0)
            for _ in range(6):
                encs, st = c.encode_batch(batches[k])
                assert all(e == 0 for e in st)
                assert [e.tobytes() for e in encs] == want[k]
                outs, st = c.decode_batch(want[k])
                assert all(e == 0 for e in st) and [o.tobytes() for o in outs] == batches[k]
        except Exception as e:      # noqa: BLE001
            errors.append((k, repr(e)))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_host_batch_beyond_the_staging_budget(ctx, snappy_raw):
    """Outputs below 512 MiB go through the library's pinned staging, which is kept until the context goes: a call stages at most
    512 MiB of them (HOST_STAGE_BUDGET, api.hip) and the rest travel straight into the caller's buffers. 1.4 GiB of 4 .. 64 MiB
    streams either way: the bytes are the same as those of small calls (the encoder's own bytes are checked against the oracle elsewhere)."""
    import os
    if os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") < (24 << 30):
        pytest.skip("needs ~6 GiB of host memory")
    text = (snappy_raw["lcet10.txt"] + snappy_raw["urls.10K"] + snappy_raw["kppkn.gtb"]) * 60     # 79 MB
    rng = np.random.default_rng(11)
    sizes = [64 << 20] * 10 + [int(x) for x in rng.integers(4 << 20, 33 << 20, size=40)]
    rng.shuffle(sizes)
    raws = [text[(k * 999983) % 1000000:][:n] for k, n in enumerate(sizes)]
    assert sum(map(len, raws)) > (1400 << 20)
    encs, st = ctx.encode_batch(raws)
    assert all(e == 0 for e in st)
    ref = {}
    for k in (0, 7, 23, 49):       # the same streams from calls of their own
        one, st1 = ctx.encode_batch([raws[k]])
        assert st1[0] == 0 and one[0].tobytes() == encs[k].tobytes(), k
        ref[k] = one[0].tobytes()
    streams = [e.tobytes() for e in encs]
    del encs
    outs, st = ctx.decode_batch(streams)
    assert all(e == 0 for e in st)
    for k, (o, r) in enumerate(zip(outs, raws)):
        assert len(o) == len(r) and o.tobytes() == r, k
