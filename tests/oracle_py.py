"""ctypes binding of the CPU oracle (oracle/liblzfse_oracle.so). Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")


def _load(name="liblzfse_oracle.so"):
    path = os.path.join(_ORACLE_DIR, name)
    srcs = [os.path.join(_ORACLE_DIR, f) for f in ("lzfse_oracle.c", "lzfse_oracle.h", "lzo_bench.c")]
    if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(f) for f in srcs if os.path.exists(f)):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, name], stdout=subprocess.DEVNULL)
    return C.CDLL(path)


LMD_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32)
MATCH_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32)
BLOCK_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32)
PACK_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32)


class Trace(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("lmd", LMD_CB), ("match", MATCH_CB), ("block", BLOCK_CB),
                ("pack", PACK_CB)]


class Oracle:
    def __init__(self, name="liblzfse_oracle.so"):
        lib = _load(name)
        self.lib = lib
        lib.lzo_encode_bound.restype = C.c_size_t
        lib.lzo_encode_bound.argtypes = [C.c_size_t]
        lib.lzo_encode.restype = C.c_int
        lib.lzo_encode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                   C.POINTER(C.c_size_t), C.c_void_p]
        lib.lzo_decode.restype = C.c_int
        lib.lzo_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                   C.POINTER(C.c_size_t), C.c_void_p]
        lib.lzo_decode_size.restype = C.c_int
        lib.lzo_decode_size.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
        lib.lzo_candidates.restype = C.c_int
        lib.lzo_candidates.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        lib.lzo_encode_guide.restype = C.c_int
        lib.lzo_encode_guide.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_uint32, C.c_uint32]
        lib.lzo_table_rows.restype = C.c_int
        lib.lzo_table_rows.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        lib.lzo_normalize_m1.restype = None
        lib.lzo_normalize_m1.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.lzo_weights_store_v2.restype = C.c_uint32
        lib.lzo_weights_store_v2.argtypes = [C.c_void_p, C.c_void_p]
        lib.lzo_weights_load_v2.restype = C.c_int
        lib.lzo_weights_load_v2.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        lib.lzo_ring_new.restype = C.c_void_p
        lib.lzo_ring_new.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.lzo_ring_write.restype = C.c_int
        lib.lzo_ring_write.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.lzo_ring_finish.restype = C.c_int
        lib.lzo_ring_finish.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        lib.lzo_ring_free.restype = None
        lib.lzo_ring_free.argtypes = [C.c_void_p]
        lib.lzo_ring_encode.restype = C.c_int
        lib.lzo_ring_encode.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                        C.POINTER(C.c_size_t), C.c_void_p]
        lib.lzo_ring_kat.restype = C.c_int
        lib.lzo_ring_kat.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t,
                                     C.c_uint32, C.c_uint32, C.c_void_p]

    @staticmethod
    def _buf(data):
        if isinstance(data, np.ndarray):
            a = np.ascontiguousarray(data, dtype=np.uint8)
        else:
            a = np.frombuffer(bytes(data), dtype=np.uint8)
        return a, a.ctypes.data if a.size else None

    def encode_bound(self, n):
        return self.lib.lzo_encode_bound(n)

    def encode(self, data, trace=None):
        a, p = self._buf(data)
        cap = self.encode_bound(a.size)
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        st = self.lib.lzo_encode(p, a.size, out.ctypes.data, cap, C.byref(n),
                                 C.byref(trace) if trace is not None else None)
        if st != 0:
            raise OracleError(st)
        return out[: n.value].tobytes()

    def decode_size(self, data):
        a, p = self._buf(data)
        v = C.c_uint64(0)
        st = self.lib.lzo_decode_size(p, a.size, C.byref(v))
        if st != 0:
            raise OracleError(st)
        return v.value

    def decode(self, data, cap=None, trace=None, as_array=False):
        a, p = self._buf(data)
        if cap is None:
            cap = self.decode_size(a)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        n = C.c_size_t(0)
        st = self.lib.lzo_decode(p, a.size, out.ctypes.data, cap, C.byref(n),
                                 C.byref(trace) if trace is not None else None)
        if st != 0:
            raise OracleError(st)
        return out[: n.value] if as_array else out[: n.value].tobytes()

    def decode_status(self, data, cap):
        a, p = self._buf(data)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        n = C.c_size_t(0)
        return self.lib.lzo_decode(p, a.size, out.ctypes.data, cap, C.byref(n), None)

    def decode_lmds(self, data):
        """Decoded LMD stream [(L, M, D)] with D substituted (lmdy_output golden shape)."""
        out = []

        def cb(_ctx, l, m, d):
            out.append((l, m, d))

        tr = Trace(None, LMD_CB(cb), MATCH_CB(), BLOCK_CB(), PACK_CB())
        raw = self.decode(data, trace=tr)
        return raw, out

    def encode_trace(self, data):
        matches, blocks, packs = [], [], []
        tr = Trace(None, LMD_CB(),
                   MATCH_CB(lambda _c, li, idx, ln, d: matches.append((li, idx, ln, d))),
                   BLOCK_CB(lambda _c, a, b, c: blocks.append((a, b, c))),
                   PACK_CB(lambda _c, l, m, d: packs.append((l, m, d))))
        enc = self.encode(data, trace=tr)
        return enc, matches, blocks, packs

    # ---- ring / stream encoder (LzfseRingEncoder::encode, LzfseWriter) ----

    def encode_guide(self, data, guide, slack):
        """lzo_encode with another BLOCK_GUIDE / SLACK (frontend_bytes.rs:19-23): the front end repositions (:348-375) on small inputs."""
        a, p = self._buf(data)
        cap = self.encode_bound(a.size)
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        st = self.lib.lzo_encode_guide(p, a.size, out.ctypes.data, cap, C.byref(n), guide, slack)
        if st != 0:
            raise OracleError(st)
        return out[: n.value].tobytes()

    def ring_encode(self, data, piece=0, trace=None):
        """The stream LzfseRingEncoder::encode / LzfseWriter produce for `data` (fed `piece` bytes at a time)."""
        a, p = self._buf(data)
        cap = self.encode_bound(a.size) + 65536
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        st = self.lib.lzo_ring_encode(p, a.size, piece, out.ctypes.data, cap, C.byref(n),
                                      C.byref(trace) if trace is not None else None)
        if st != 0:
            raise OracleError(st)
        return out[: n.value].tobytes()

    def ring_encode_pieces(self, pieces):
        """The same through the handle API, one write per element of `pieces`."""
        h = self.lib.lzo_ring_new(0, 0, 0, None)
        try:
            for pc in pieces:
                a, p = self._buf(pc)
                st = self.lib.lzo_ring_write(h, p, a.size)
                if st != 0:
                    raise OracleError(st)
            ptr, n = C.c_void_p(), C.c_size_t(0)
            st = self.lib.lzo_ring_finish(h, C.byref(ptr), C.byref(n))
            if st != 0:
                raise OracleError(st)
            return C.string_at(ptr, n.value)
        finally:
            self.lib.lzo_ring_free(h)

    def ring_encode_trace(self, data, piece=0):
        matches, blocks, packs = [], [], []
        tr = Trace(None, LMD_CB(),
                   MATCH_CB(lambda _c, li, idx, ln, d: matches.append((li, idx, ln, d))),
                   BLOCK_CB(lambda _c, a, b, c: blocks.append((a, b, c))),
                   PACK_CB(lambda _c, l, m, d: packs.append((l, m, d))))
        enc = self.ring_encode(data, piece, trace=tr)
        return enc, matches, blocks, packs

    def ring_kat(self, mode, ring, ring_data, idx0, n):
        """frontend_ring.rs:861-992 set-ups on the test ring `ring` = (size, block, limit); returns the Dummy backend's
        pushes as (literal_len, match_len, distance) and the literal positions."""
        out = []
        tr = Trace(None, LMD_CB(), MATCH_CB(lambda _c, li, idx, ln, d: out.append((li, idx - li, ln, d))),
                   BLOCK_CB(), PACK_CB())
        a, p = self._buf(ring_data)
        st = self.lib.lzo_ring_kat(mode, ring[0], ring[1], ring[2], p, a.size, idx0, n, C.byref(tr))
        if st != 0:
            raise OracleError(st)
        return out

    def candidates(self, data):
        a, p = self._buf(data)
        mi = np.empty(a.size, dtype=np.uint32)
        fl = np.empty(a.size, dtype=np.uint32)
        st = self.lib.lzo_candidates(p, a.size, mi.ctypes.data, fl.ctypes.data)
        if st != 0:
            raise OracleError(st)
        return mi[: a.size - 3], fl[: a.size - 3]


    def table_rows(self, data):
        a, p = self._buf(data)
        rows = np.empty((a.size, 4), dtype=np.uint32)
        st = self.lib.lzo_table_rows(p, a.size, rows.ctypes.data)
        if st != 0:
            raise OracleError(st)
        return rows[: a.size - 3]


class OracleError(Exception):
    def __init__(self, status):
        super().__init__(f"oracle status {status}")
        self.status = status


# ---- restatement of test_kit generators (test_kit/src/rng.rs:14-58, seq.rs:25-34) ----

def rng_gen_vec(seed, length):
    """Rng::new(seed).gen_vec(length): emits the CURRENT state as LE u32 then advances."""
    n4 = length // 4
    out = np.empty(n4 + 1, dtype=np.uint32)
    s = seed & 0xFFFFFFFF
    for i in range(n4 + 1):
        out[i] = s
        s = (s * 1103515245 + 12345) & 0xFFFFFFFF
    return out.view(np.uint8)[:length].tobytes()


def seq_masked(seed, mask, length):
    """Seq::masked(Rng::new(seed), mask): bytes of (rng.gen() & mask), low byte first."""
    n4 = (length + 3) // 4
    out = np.empty(n4, dtype=np.uint32)
    s = seed & 0xFFFFFFFF
    for i in range(n4):
        s = (s * 1103515245 + 12345) & 0xFFFFFFFF
        out[i] = s & mask
    return out.view(np.uint8)[:length].tobytes()
