"""Host-side mirror of lzfse_rust's slice API over the MI355X C ABI.

    LzfseEncoder.encode_bytes(src, dst) -> int     src/encode/encoder.rs:49-53
    LzfseDecoder.decode_bytes(src, dst) -> int     src/decode/decoder.rs:61-69
    LzfseRingEncoder.encode(reader, writer) / .writer(inner) / .writer_bytes(vec)   src/encode/ring_encoder.rs:55-97
    LzfseRingDecoder.decode(reader, writer) / .reader(inner) / .reader_bytes(bytes) src/decode/ring_decoder.rs:57-90
    encode_bytes / decode_bytes (free functions)   src/encode/mod.rs:58-60, src/decode/mod.rs:49-51

Same names, argument meaning and error behaviour: `dst` (a bytearray, the Vec<u8>) is appended
to, never cleared, and the number of appended bytes is returned; decode errors raise LzfseError
carrying the status code that maps 1:1 on crate::Error. All compute runs on the GPU.
"""
import ctypes as C

import numpy as np

from . import _native

OK = 0
BAD_READER_STATE = 5
BUFFER_OVERFLOW = 6
PAYLOAD_OVERFLOW = 7


class LzfseError(Exception):
    """crate::Error (src/error/mod.rs:40-61) / io::Error for the encoder."""

    def __init__(self, status, detail=0):
        self.status = int(status)
        self.detail = int(detail)  # BadBlock(magic) / BadLmdCount(n) / BadLiteralCount(n), else 0
        msg = _native.lib().lzfse_mi_status_string(self.status).decode()
        super().__init__(f"lzfse status {self.status}: {msg}" + (f" (0x{self.detail:08X})" if self.detail else ""))


def _check(st):
    if st != OK:
        raise LzfseError(st)


def device_count():
    """lzfse_mi_device_count: HIP devices visible to the process (0 without one)."""
    return int(_native.lib().lzfse_mi_device_count())


class Context:
    """One HIP device + stream + scratch (lzfse_mi_ctx)."""

    OPTIONS = {"encode_lanes": 1, "decode_lanes": 2, "stagger": 3, "decode_pipe": 4, "stream_spare": 5, "diag_lz_path": 100, "diag_lz_tile": 101, "diag_stats": 102, "diag_chain": 103, "diag_walk": 104, "diag_pipe_scatter": 105, "diag_guide": 106}

    def __init__(self, device=0, diag=False):
        self._lib = _native.lib(diag=diag)
        h = C.c_void_p()
        _check(self._lib.lzfse_mi_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lzfse_mi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name, value):
        """lzfse_mi_set_option; the diag_* options need Context(diag=True)."""
        _check(self._lib.lzfse_mi_set_option(self._h, self.OPTIONS[name], int(value)))

    def pipe_refusals(self):
        """lzfse_mi_get_info(LZFSE_MI_INFO_PIPE_REFUSALS): times the pipelined LZ stage of decode was given up (0 = in use)."""
        v = C.c_int64(0)
        _check(self._lib.lzfse_mi_get_info(self._h, 1, C.byref(v)))
        return v.value

    # -- stream / timing plumbing --
    def set_stream(self, hip_stream_ptr):
        _check(self._lib.lzfse_mi_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    def enable_timing(self, on=True):
        _check(self._lib.lzfse_mi_enable_timing(self._h, 1 if on else 0))

    def timings(self):
        t = _native.Timings()
        _check(self._lib.lzfse_mi_get_timings(self._h, C.byref(t)))
        return {t.names[i].decode(): (float(t.ms[i]), int(t.launches[i])) for i in range(t.n_stages)}

    def error_detail(self, stream_index=0):
        """u32 payload of Error::BadBlock / FseErrorKind::BadLmdCount / BadLiteralCount for a stream of the last call."""
        v = C.c_uint32(0)
        _check(self._lib.lzfse_mi_last_error_detail(self._h, int(stream_index), C.byref(v)))
        return v.value

    # -- host-pointer batch --
    def _host_batch(self, fn, srcs, caps):
        """One call of a host-pointer batch entry point. The outputs are views into ONE buffer, and the pointer arrays are made
        by numpy: with tens of thousands of small streams a Python-level array per stream costs more than the device call."""
        n = len(srcs)
        arrs = [s if (isinstance(s, np.ndarray) and s.dtype == np.uint8 and s.ndim == 1 and s.flags.c_contiguous)
                else np.frombuffer(s if isinstance(s, (bytes, bytearray, memoryview, np.ndarray)) else bytes(s), dtype=np.uint8) for s in srcs]
        lens = np.fromiter((a.size for a in arrs), dtype=np.uint64, count=n)
        sp = np.fromiter((a.__array_interface__["data"][0] if a.size else 0 for a in arrs), dtype=np.uint64, count=n)
        dc = np.asarray(caps, dtype=np.uint64).reshape(n)
        room = np.maximum(dc, 1)
        offs = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum((room + np.uint64(63)) & ~np.uint64(63), out=offs[1:])
        big = np.empty(int(offs[n]) + 64, dtype=np.uint8)
        dp = np.uint64(big.ctypes.data) + offs[:n]
        ol = np.zeros(n, dtype=np.uint64)
        st = np.zeros(n, dtype=np.int32)
        vpp, szp, ip = C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int)
        _check(fn(self._h, n, sp.ctypes.data_as(vpp), lens.ctypes.data_as(szp), dp.ctypes.data_as(vpp), dc.ctypes.data_as(szp),
                  ol.ctypes.data_as(szp), st.ctypes.data_as(ip)))
        o, e = offs.tolist(), ol.tolist()
        return [big[o[i]: o[i] + e[i]] for i in range(n)], st.tolist()

    def encode_batch(self, srcs, ring=False):
        """ring=True: the streams LzfseRingEncoder::encode / LzfseWriter produce (encode/frontend_ring.rs: another parse)."""
        caps = [self._lib.lzfse_mi_encode_bound(len(s)) for s in srcs] if len(srcs) < 64 else _encode_bounds(self._lib, srcs)
        return self._host_batch(self._lib.lzfse_mi_encode_ring_batch if ring else self._lib.lzfse_mi_encode_batch, srcs, caps)

    def decode_batch(self, srcs, caps=None):
        if caps is None:
            if len(srcs) < 64:
                caps = [decode_size(s, partial=True) for s in srcs]
            else:   # (many streams: no numpy / ctypes object per header walk)
                srcs = [s if isinstance(s, (bytes, np.ndarray)) else bytes(s) for s in srcs]
                v, size_of, caps = C.c_uint64(0), self._lib.lzfse_mi_decode_size, []
                ref = C.byref(v)
                for s in srcs:
                    size_of(s if isinstance(s, bytes) else s.ctypes.data, len(s), ref)
                    caps.append(v.value)
        return self._host_batch(self._lib.lzfse_mi_decode_batch, srcs, caps)

    # -- device-resident batch: raw device pointers (e.g. torch tensor.data_ptr()) --
    def _device_batch(self, fn, d_src, src_off, src_len, d_dst, dst_off, dst_cap):
        n = len(src_off)
        so = np.ascontiguousarray(src_off, dtype=np.uint64)
        sl = np.ascontiguousarray(src_len, dtype=np.uint64)
        do = np.ascontiguousarray(dst_off, dtype=np.uint64)
        dc = np.ascontiguousarray(dst_cap, dtype=np.uint64)
        ol = np.zeros(n, dtype=np.uint64)
        st = np.zeros(n, dtype=np.int32)
        u64p, ip = C.POINTER(C.c_uint64), C.POINTER(C.c_int)
        _check(fn(self._h, n, C.c_void_p(d_src), so.ctypes.data_as(u64p), sl.ctypes.data_as(u64p),
                  C.c_void_p(d_dst), do.ctypes.data_as(u64p), dc.ctypes.data_as(u64p), ol.ctypes.data_as(u64p),
                  st.ctypes.data_as(ip)))
        return ol, st

    def decode_batch_device(self, d_src, src_off, src_len, d_dst, dst_off, dst_cap):
        return self._device_batch(self._lib.lzfse_mi_decode_batch_device, d_src, src_off, src_len, d_dst, dst_off,
                                  dst_cap)

    def encode_batch_device(self, d_src, src_off, src_len, d_dst, dst_off, dst_cap, ring=False):
        fn = self._lib.lzfse_mi_encode_ring_batch_device if ring else self._lib.lzfse_mi_encode_batch_device
        return self._device_batch(fn, d_src, src_off, src_len, d_dst, dst_off, dst_cap)


def _encode_bounds(lib, srcs):
    """lzfse_mi_encode_bound for many inputs at once (the library's formula, checked against the library on one of them)"""
    n = np.fromiter((len(s) for s in srcs), dtype=np.uint64, count=len(srcs))
    caps = n + n // np.uint64(2) + n // np.uint64(4) + np.uint64(4096)
    assert int(caps[0]) == lib.lzfse_mi_encode_bound(int(n[0]))
    return caps


def encode_small(src):
    """Host-side size classes (n <= 4096): raw / LZVN blocks, exactly as the reference's CPU path."""
    a = np.frombuffer(bytes(src), dtype=np.uint8)
    out = np.empty(a.size + 64, dtype=np.uint8)
    n = C.c_size_t(0)
    _check(_native.lib().lzfse_mi_encode_small(a.ctypes.data if a.size else None, a.size, out.ctypes.data, out.size, C.byref(n)))
    return out[: n.value].tobytes()


def encode_bound(n):
    return _native.lib().lzfse_mi_encode_bound(int(n))


def decode_size(src, partial=False):
    """decode::probe: sum of n_raw_bytes over the block headers (src/decode/probe.rs:11-35). partial=True: no raise on
    a header-level error; the size of the blocks before it (what the decoder needs to reach the reference's error)."""
    a = np.frombuffer(bytes(src) if not isinstance(src, np.ndarray) else src, dtype=np.uint8)
    v = C.c_uint64(0)
    st = _native.lib().lzfse_mi_decode_size(a.ctypes.data if a.size else None, a.size, C.byref(v))
    if not partial:
        _check(st)
    return v.value


def _as_u8(src):
    """`src` as a numpy byte array without a copy where the buffer protocol allows it"""
    if isinstance(src, np.ndarray) and src.dtype == np.uint8 and src.ndim == 1 and src.flags.c_contiguous:
        return src
    try:
        return np.frombuffer(src, dtype=np.uint8)
    except (TypeError, ValueError):
        return np.frombuffer(bytes(src), dtype=np.uint8)


_resize = C.pythonapi.PyByteArray_Resize
_resize.argtypes, _resize.restype = [C.py_object, C.c_ssize_t], C.c_int


def _into_tail(ctx, fn, src, dst, cap):
    """One single-stream call (lzfse_mi_encode / lzfse_mi_decode) whose destination is the tail of bytearray `dst` itself: the
    Vec<u8> of the reference is written where it lies. `dst` grows by `cap` without its new bytes being touched
    (PyByteArray_Resize: `dst += bytes(cap)` would fault every new page in, and `dst += out.tobytes()` copy the result twice on
    top -- 64 MiB decoded: 32 ms that way, 6 ms this way, scripts/fresh_dst.py) and is cut back to what the call produced; an
    error leaves it as it was. Returns (status, bytes appended)."""
    a = _as_u8(src)
    old = len(dst)
    _resize(dst, old + max(int(cap), 1))     # (BufferError when a memoryview of `dst` is alive: as for `dst +=`)
    n = C.c_size_t(0)
    st = BAD_READER_STATE
    try:
        tail = (C.c_uint8 * max(int(cap), 1)).from_buffer(dst, old)
        try:
            st = fn(ctx._h, a.ctypes.data if a.size else None, a.size, C.addressof(tail), int(cap), C.byref(n))
        finally:
            del tail
    finally:
        _resize(dst, old + (n.value if st == OK else 0))
    return st, (n.value if st == OK else 0)


class LzfseEncoder:
    """src/encode/encoder.rs:14-54."""

    def __init__(self, device=0, context=None):
        self._ctx = context or Context(device)

    def encode_bytes(self, src, dst):
        """Appends the LZFSE stream of `src` to bytearray `dst`; returns bytes appended."""
        if type(dst) is bytearray:
            st, n = _into_tail(self._ctx, self._ctx._lib.lzfse_mi_encode, src, dst, self._ctx._lib.lzfse_mi_encode_bound(int(_as_u8(src).size)))
            _check(st)
            return n
        outs, st = self._ctx.encode_batch([src])
        _check(st[0])
        dst += outs[0].tobytes()
        return len(outs[0])


class LzfseDecoder:
    """src/decode/decoder.rs:17-99."""

    def __init__(self, device=0, context=None):
        self._ctx = context or Context(device)

    def decode_bytes(self, src, dst):
        """Appends the decoded bytes of stream `src` to bytearray `dst`; returns bytes appended."""
        if type(dst) is bytearray:
            fn, cap = self._ctx._lib.lzfse_mi_decode, decode_size(src, partial=True)
            st, n = _into_tail(self._ctx, fn, src, dst, cap)
            if st == BUFFER_OVERFLOW:   # (see below)
                a = _as_u8(src)
                st, n = _into_tail(self._ctx, fn, src, dst, cap + self._ctx._lib.lzfse_mi_decode_headroom(a.ctypes.data if a.size else None, a.size))
            if st != OK:
                raise LzfseError(st, self._ctx.error_detail(0))
            return n
        outs, st = self._ctx.decode_batch([src])
        if st[0] == BUFFER_OVERFLOW:
            # `dst` is a Vec in the reference: a block that produces more than its header says runs to its last LMD and
            # fails there (fse/fse_core.rs:132-140); give it the room to reach that error
            a = np.frombuffer(bytes(src), dtype=np.uint8)
            room = self._ctx._lib.lzfse_mi_decode_headroom(a.ctypes.data if a.size else None, a.size)
            outs, st = self._ctx.decode_batch([src], caps=[decode_size(src, partial=True) + room])
        if st[0] != OK:
            raise LzfseError(st[0], self._ctx.error_detail(0))
        dst += outs[0].tobytes()
        return len(outs[0])


class LzfseRingEncoder:
    """src/encode/ring_encoder.rs:17-97. encode(reader, writer) -> (bytes read, bytes written); encode_bytes is the slice
    encoder's (ring_encoder.rs:71-73); writer(inner) / writer_bytes(vec) give the Write front ends. The streams are the
    ring front end's (encode/frontend_ring.rs), not the slice encoder's bytes."""

    def __init__(self, device=0, context=None, read_size=1 << 20, window=0, zero_copy=True):
        self._ctx = context or Context(device)
        self._read_size = read_size
        self._window = window    # input bytes per device call (0 = 64 MiB)
        # True: the sink gets a view of the library's buffer that is released after the call (a sink that keeps one fails loudly
        # later); False: bytes it may keep, e.g. `pieces.append` (one copy more)
        self._zero_copy = zero_copy

    def encode(self, reader, writer):
        w = LzfseWriter(self._ctx, writer, self._window, self._zero_copy)
        n_in = 0
        readinto = getattr(reader, "readinto", None)
        if readinto is not None:
            # copy(reader) of the reference reads straight into its ring (ring_encoder.rs:55-67): so does this, into the window
            # buffer the library says (lzfse_mi_estream_reserve / _commit)
            while True:
                view = w._reserve(self._read_size)
                try:
                    k = readinto(view)
                finally:
                    view.release()
                w._commit(k or 0)
                if not k:
                    break
                n_in += k
        else:
            while True:
                piece = reader.read(self._read_size)
                if not piece:
                    break
                n_in += w.write(piece)
        w.finalize()
        return n_in, w.bytes_out

    def encode_bytes(self, src, dst):
        return LzfseEncoder(context=self._ctx).encode_bytes(src, dst)

    def writer(self, inner):
        """LzfseRingEncoder::writer (ring_encoder.rs:79-84)"""
        return LzfseWriter(self._ctx, inner, self._window, self._zero_copy)

    def writer_bytes(self, vec):
        """LzfseRingEncoder::writer_bytes (ring_encoder.rs:91-96): `vec` (a bytearray) is appended to"""
        return LzfseWriterBytes(self._ctx, vec, self._window)


class LzfseWriter:
    """encode/writer.rs:12-75: write(buf) takes all of buf, flush() does nothing, finalize() ends the stream and returns
    `inner`. The stream reaches `inner` a window at a time (`window` input bytes per device call, 0 = 64 MiB): the blocks
    no later input can change leave during write(), the rest at finalize(). Dropping a writer without finalize() loses
    the end of the stream, as in the reference (writer.rs:36-38)."""

    def __init__(self, context, inner, window=0, zero_copy=True):
        self._ctx, self._inner = context, inner
        self._lib = context._lib
        self._h = C.c_void_p()
        self._failure = []
        self._zero_copy = zero_copy

        def _write(_user, p, n):
            try:
                # (as in LzfseRingDecoder: a view of the library's buffer, released after the call, so that a sink that keeps it
                # fails loudly later instead of reading a reused buffer; file objects, hashers, `bytearray +=` copy what they need)
                if not n:
                    self._sink(b"")
                elif not self._zero_copy:
                    self._sink(C.string_at(p, n))   # bytes the sink may keep (one copy more; LzfseRingDecoder's zero_copy=False does the same)
                else:
                    view = memoryview((C.c_uint8 * n).from_address(C.addressof(p.contents)))
                    try:
                        self._sink(view)
                    finally:
                        view.release()
                return 0
            except Exception as e:   # the sink's error travels back through the C layer as LZFSE_MI_IO
                self._failure.append(e)
                return 1

        self._cb = _native.WRITE_FN(_write)
        _check(self._lib.lzfse_mi_estream_create(context._h, window, C.byref(self._h)))
        self.bytes_out = 0

    def write(self, buf):
        try:
            a = np.frombuffer(buf, dtype=np.uint8)     # (no copy: the library takes the bytes into its own buffer)
        except (TypeError, ValueError):
            a = np.frombuffer(bytes(buf), dtype=np.uint8)
        st = self._lib.lzfse_mi_estream_feed(self._h, a.ctypes.data if a.size else None, a.size, self._cb, None)
        if self._failure:
            raise self._failure.pop()
        _check(st)
        return a.size

    def _reserve(self, want):
        """lzfse_mi_estream_reserve: a writable view of where the next input bytes go (release it before _commit)"""
        p, room = C.c_void_p(), C.c_size_t(0)
        st = self._lib.lzfse_mi_estream_reserve(self._h, int(want), C.byref(p), C.byref(room), self._cb, None)
        if self._failure:
            raise self._failure.pop()
        _check(st)
        return memoryview((C.c_uint8 * room.value).from_address(p.value)).cast("B")

    def _commit(self, n):
        _check(self._lib.lzfse_mi_estream_commit(self._h, int(n)))

    def flush(self):
        pass

    def _sink(self, piece):
        self._inner.write(piece)

    def finalize(self):
        u, v = C.c_uint64(0), C.c_uint64(0)
        try:
            st = self._lib.lzfse_mi_estream_finish(self._h, self._cb, None, C.byref(u), C.byref(v))
        finally:
            self.close()
        if self._failure:
            raise self._failure.pop()
        _check(st)
        self.bytes_out = v.value
        return self._inner

    def close(self):
        if self._h:
            self._lib.lzfse_mi_estream_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LzfseWriterBytes(LzfseWriter):
    """encode/writer_bytes.rs:12-78: the same over a Vec<u8> (here a bytearray); finalize() returns it."""

    def _sink(self, piece):
        self._inner += piece


class LzfseRingDecoder:
    """src/decode/ring_decoder.rs:17-68: decode(reader, writer) -> (bytes read, bytes written). `reader.read(n)` returns
    b"" at the end of the input, `writer.write(b)` takes the output in pieces; `window` = raw bytes per device call.
    The writer is handed a memoryview of the library's window buffer that is valid DURING the write call only and released
    after it (file objects, hashers, `bytearray +=` are fine; a sink that keeps the object gets a ValueError when it uses it
    later, one that keeps an export of it makes the call fail); `zero_copy=False` hands out bytes it may keep instead."""

    def __init__(self, device=0, context=None, window=0, read_size=1 << 20, zero_copy=True):
        self._ctx = context or Context(device)
        self._window = window
        self._read_size = read_size
        self._zero_copy = zero_copy

    def decode(self, reader, writer):
        lib = self._ctx._lib
        h = C.c_void_p()
        _check(lib.lzfse_mi_dstream_create(self._ctx._h, self._window, C.byref(h)))
        failure = []

        def _write(_user, p, n):
            try:
                # A view of the library's window buffer, valid during the call only -- and RELEASED after it, so that a writer
                # that keeps the object it was handed (list.append, a queue) gets a ValueError when it touches it later instead
                # of reading memory that has been overwritten or freed. zero_copy=False hands out bytes instead (one copy and a
                # fresh allocation per piece: 64 MiB windows then decode at 3.2 instead of 10.3 GB/s, profiles/r04_stream_bench.txt)
                if not n:
                    writer.write(b"")
                elif not self._zero_copy:
                    writer.write(bytes((C.c_uint8 * n).from_address(C.addressof(p.contents))))
                else:
                    view = memoryview((C.c_uint8 * n).from_address(C.addressof(p.contents)))
                    try:
                        writer.write(view)
                    finally:
                        view.release()   # (BufferError if the writer still holds an export of it, e.g. np.frombuffer: reported below)
                return 0
            except Exception as e:   # the sink's error travels back through the C layer as LZFSE_MI_IO
                failure.append(e)
                return 1

        cb = _native.WRITE_FN(_write)
        readinto = getattr(reader, "readinto", None)
        try:
            while True:
                if readinto is not None:
                    # (the reference reads straight into its ring, ring_decoder.rs:57-67: so does this, into the library's input buffer)
                    p = C.c_void_p()
                    st = lib.lzfse_mi_dstream_reserve(h, self._read_size, C.byref(p))
                    if st != OK:
                        raise LzfseError(st, 0)
                    view = memoryview((C.c_uint8 * self._read_size).from_address(p.value)).cast("B")
                    try:
                        k = readinto(view) or 0
                    finally:
                        view.release()
                    st = lib.lzfse_mi_dstream_commit(h, k, 0 if k else 1, cb, None)
                else:
                    piece = reader.read(self._read_size)
                    a = np.frombuffer(piece, dtype=np.uint8)
                    k = a.size
                    st = lib.lzfse_mi_dstream_feed(h, a.ctypes.data if a.size else None, a.size, 0 if a.size else 1, cb, None)
                if failure:
                    raise failure[0]
                if st != OK:
                    raise LzfseError(st, 0)
                if not k:
                    break
            u, v = C.c_uint64(0), C.c_uint64(0)
            _check(lib.lzfse_mi_dstream_totals(h, C.byref(u), C.byref(v)))
            return u.value, v.value
        finally:
            lib.lzfse_mi_dstream_destroy(h)

    def reader(self, inner):
        """LzfseRingDecoder::reader (decode/ring_decoder.rs:75-80)."""
        return LzfseReader(self._ctx, inner, self._window, self._read_size)

    def reader_bytes(self, data):
        """LzfseRingDecoder::reader_bytes (decode/ring_decoder.rs:82-89)."""
        import io
        return LzfseReader(self._ctx, io.BytesIO(bytes(data)), self._window, self._read_size)


class LzfseReader:
    """LzfseReader / LzfseReaderBytes (decode/ring_decoder.rs:75-90, 135-170; Read impl decode/reader_core.rs:170-188): a
    reader over the decoded bytes of `inner`. read(n) returns n bytes unless the stream ends first, b"" from then on; a
    decode error is raised by the read that reaches it; after PayloadOverflow (bytes behind bvx$) the reader is in
    State::Err and every further read is BadReaderState (reader_core.rs:62-76, 160-168)."""

    def __init__(self, context, inner, window=0, read_size=1 << 20):
        self._ctx, self._inner, self._read_size = context, inner, read_size
        self._lib = context._lib
        self._h = C.c_void_p()
        _check(self._lib.lzfse_mi_dstream_create(context._h, window, C.byref(self._h)))
        self._out = bytearray()
        self._pos = 0
        self._done = False
        self._error = None
        self._cb = _native.WRITE_FN(self._sink)

    def _sink(self, _user, p, n):
        self._out += C.string_at(p, n)
        return 0

    def _fill(self):
        piece = self._inner.read(self._read_size)
        a = np.frombuffer(piece, dtype=np.uint8)
        st = self._lib.lzfse_mi_dstream_feed(self._h, a.ctypes.data if a.size else None, a.size, 0 if a.size else 1, self._cb, None)
        if st != OK:
            self._error = BAD_READER_STATE if st == PAYLOAD_OVERFLOW else st
            raise LzfseError(st, self._ctx.error_detail(0) if st != PAYLOAD_OVERFLOW else 0)
        if not a.size:
            self._done = True

    def read(self, n=-1):
        if self._error is not None:
            raise LzfseError(self._error)
        while not self._done and (n < 0 or len(self._out) - self._pos < n):
            self._fill()
        end = len(self._out) if n < 0 else min(len(self._out), self._pos + n)
        piece = bytes(self._out[self._pos:end])
        self._pos = end
        if self._pos > (1 << 22):      # drop what has been handed out
            del self._out[:self._pos]
            self._pos = 0
        return piece

    def readinto(self, b):
        piece = self.read(len(b))
        b[:len(piece)] = piece
        return len(piece)

    def into_inner(self):
        self.close()
        return self._inner

    def close(self):
        if self._h:
            self._lib.lzfse_mi_dstream_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def encode_bytes(src, dst):
    return LzfseEncoder().encode_bytes(src, dst)


def decode_bytes(src, dst):
    return LzfseDecoder().decode_bytes(src, dst)


# ---- chunked container over one or several devices (include/lzfse_mi.h, chunked section) ----

def _ctx_array(contexts):
    arr = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    return arr


def encode_chunked(contexts, src, chunk=0, cap=None):
    """One large input -> "LZMC" frame of independent LZFSE streams, chunk c on contexts[c mod len(contexts)]. `cap`: the
    destination's size (default lzfse_mi_chunked_bound, with which every chunk is encoded straight into the frame; a smaller
    one that still holds the frame works through private buffers)."""
    lib = contexts[0]._lib
    a = np.frombuffer(src, dtype=np.uint8) if not isinstance(src, np.ndarray) else src
    if cap is None:
        cap = lib.lzfse_mi_chunked_bound(a.size, chunk)
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    _check(lib.lzfse_mi_encode_chunked(_ctx_array(contexts), len(contexts), a.ctypes.data if a.size else None, a.size, chunk,
                                       out.ctypes.data, cap, C.byref(n)))
    return out[: n.value]


def decode_chunked(contexts, frame):
    lib = contexts[0]._lib
    a = np.frombuffer(frame, dtype=np.uint8) if not isinstance(frame, np.ndarray) else frame
    raw = C.c_uint64(0)
    _check(lib.lzfse_mi_decode_chunked_size(a.ctypes.data if a.size else None, a.size, C.byref(raw)))
    out = np.empty(max(raw.value, 1), dtype=np.uint8)
    n = C.c_size_t(0)
    _check(lib.lzfse_mi_decode_chunked(_ctx_array(contexts), len(contexts), a.ctypes.data, a.size, out.ctypes.data, raw.value,
                                       C.byref(n)))
    return out[: n.value]
