"""What a FRESH destination costs the host-pointer calls: `decode_bytes(&[u8]) -> Vec<u8>` hands over memory that has never been
touched (a large malloc is an mmap), and a transfer into such pages faults them in one by one.
lzfse_mi_decode of one text stream into (a) a buffer used before, (b) fresh memory every call, (c) fresh memory populated first
with madvise(MADV_POPULATE_WRITE) by 1 / 8 threads (timed apart: inside the library this runs under the kernels).
    python scripts/fresh_dst.py      (profiles/r04_fresh_dst.txt)"""
import ctypes as C, mmap, os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import lzfse_rust_amd as lz
from lzfse_rust_amd import _native
from bench import synth_text

libc = C.CDLL(None, use_errno=True)
libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
MADV_POPULATE_WRITE = 23
ctx = lz.Context(0)
lib = _native.lib()


def fresh(n):
    return mmap.mmap(-1, n + 4096, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)


def addr(m):
    return C.addressof(C.c_char.from_buffer(m))


def populate(a, n, threads):
    a0 = a & ~4095
    n = (a + n + 4095 & ~4095) - a0
    per = ((n // threads) + (2 << 20) - 1) & ~((2 << 20) - 1)
    ts = []
    for t in range(threads):
        lo, hi = t * per, min(n, (t + 1) * per)
        if lo >= hi:
            break
        th = threading.Thread(target=lambda lo=lo, hi=hi: libc.madvise(a0 + lo, hi - lo, MADV_POPULATE_WRITE))
        th.start(); ts.append(th)
    for th in ts:
        th.join()


for mb in (16, 64, 256):
    raw = bytes(synth_text(mb << 20))
    enc = ctx.encode_batch([raw])[0][0].tobytes()
    n = len(raw)
    out_len = C.c_size_t(0)

    def call(dst_addr, what="decode"):
        if what == "decode":
            st = lib.lzfse_mi_decode(ctx._h, enc, len(enc), C.c_void_p(dst_addr), n, C.byref(out_len))
        else:
            st = lib.lzfse_mi_encode(ctx._h, raw, n, C.c_void_p(dst_addr), lz.encode_bound(n), C.byref(out_len))
        assert st == 0, st

    keep = fresh(max(n, lz.encode_bound(n)))
    ka = addr(keep)
    for what, size in (("decode", n), ("encode", lz.encode_bound(n))):
        call(ka, what); call(ka, what)
        t = time.perf_counter(); call(ka, what); dt_used = time.perf_counter() - t
        res = []
        for mode in ("fresh", "pop1", "pop8"):
            best = None
            for _ in range(3):
                m = fresh(size)
                a = addr(m)
                tp = 0.0
                if mode != "fresh":
                    t = time.perf_counter(); populate(a, size, 1 if mode == "pop1" else 8); tp = time.perf_counter() - t
                t = time.perf_counter(); call(a, what); dt = time.perf_counter() - t
                if best is None or dt + tp < best[0] + best[1]:
                    best = (dt, tp)
                del a
                m.close()
            res.append((mode, best))
        print(f"{mb:4d} MiB {what}: used buffer {dt_used * 1e3:7.2f} ms = {n / dt_used / 1e9:5.2f} GB/s | " +
              " | ".join(f"{mode} call {b[0] * 1e3:7.2f} ms" + (f" + populate {b[1] * 1e3:6.2f} ms" if b[1] else "") for mode, b in res), flush=True)
