"""What the host-pointer path can count on: pageable vs registered vs pinned transfers on this box (hipMemcpy, libamdhip64 by ctypes)."""
import ctypes as C, time
import numpy as np
rt = C.CDLL("libamdhip64.so")
rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
rt.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
rt.hipHostUnregister.argtypes = [C.c_void_p]
rt.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        a = time.perf_counter(); f(); best = min(best, time.perf_counter() - a)
    return best
for mb in (8, 94, 1024):
    n = mb << 20
    d = C.c_void_p(); assert rt.hipMalloc(C.byref(d), n) == 0
    a = np.ones(n, dtype=np.uint8)
    b = np.empty(n, dtype=np.uint8); b[:] = 1
    h2d = t(lambda: rt.hipMemcpy(d, a.ctypes.data, n, 1))
    d2h = t(lambda: rt.hipMemcpy(b.ctypes.data, d, n, 2))
    reg = t(lambda: (rt.hipHostRegister(a.ctypes.data, n, 0), rt.hipHostUnregister(a.ctypes.data)), reps=2)
    assert rt.hipHostRegister(a.ctypes.data, n, 0) == 0
    h2d_r = t(lambda: rt.hipMemcpy(d, a.ctypes.data, n, 1))
    rt.hipHostUnregister(a.ctypes.data)
    p = C.c_void_p(); assert rt.hipHostMalloc(C.byref(p), n, 0) == 0
    h2d_p = t(lambda: rt.hipMemcpy(d, p, n, 1))
    d2h_p = t(lambda: rt.hipMemcpy(p, d, n, 2))
    cp = t(lambda: C.memmove(p, a.ctypes.data, n))
    print(f"{mb:5d} MiB: pageable H2D {n/h2d/1e9:6.1f} GB/s  D2H {n/d2h/1e9:6.1f} | register+unregister {reg*1e3:7.2f} ms ({n/reg/1e9:5.1f} GB/s)  registered H2D {n/h2d_r/1e9:6.1f} | pinned H2D {n/h2d_p/1e9:6.1f}  D2H {n/d2h_p/1e9:6.1f} | memcpy pageable->pinned (1 thread) {n/cp/1e9:5.1f} GB/s")

# many separate pageable buffers (what a batch of caller streams is): per-call cost of direct copies
rt.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
rt.hipDeviceSynchronize.argtypes = []
for cnt, each in ((384, 245 << 10), (3072, 245 << 10), (64, 4 << 20)):
    bufs = [np.ones(each, dtype=np.uint8) for _ in range(cnt)]
    outs = [np.empty(each, dtype=np.uint8) for _ in range(cnt)]
    for o in outs: o[:] = 0
    d = C.c_void_p(); assert rt.hipMalloc(C.byref(d), cnt * each) == 0
    def h2d():
        for k, b in enumerate(bufs): rt.hipMemcpyAsync(d.value + k * each, b.ctypes.data, each, 1, None)
        rt.hipDeviceSynchronize()
    def d2h():
        for k, b in enumerate(outs): rt.hipMemcpyAsync(b.ctypes.data, d.value + k * each, each, 2, None)
        rt.hipDeviceSynchronize()
    a, b2 = t(h2d), t(d2h)
    rc = [rt.hipHostRegister(b.ctypes.data, each, 0) for b in bufs[:4]]
    print(f"{cnt} buffers x {each >> 10} KiB: direct pageable H2D {cnt*each/a/1e9:5.1f} GB/s ({a*1e3:.2f} ms)  D2H {cnt*each/b2/1e9:5.1f} GB/s ({b2*1e3:.2f} ms)  hipHostRegister rc {rc}")

# registered caller memory seen from the device: hipHostGetDevicePointer + a device-side copy (blit) reading / writing it
rt.hipHostGetDevicePointer.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint]
n = 94 << 20
a = np.ones(n, dtype=np.uint8); b = np.empty(n, dtype=np.uint8); b[:] = 0
d = C.c_void_p(); assert rt.hipMalloc(C.byref(d), n) == 0
t0 = time.perf_counter()
r1 = rt.hipHostRegister(a.ctypes.data, n, 0); r2 = rt.hipHostRegister(b.ctypes.data, n, 0)
t1 = time.perf_counter()
pa, pb = C.c_void_p(), C.c_void_p()
g1 = rt.hipHostGetDevicePointer(C.byref(pa), a.ctypes.data, 0); g2 = rt.hipHostGetDevicePointer(C.byref(pb), b.ctypes.data, 0)
print("register rc", r1, r2, f"{(t1-t0)*1e3:.3f} ms; devptr rc", g1, g2, hex(pa.value or 0), hex(a.ctypes.data))
if g1 == 0 and g2 == 0:
    rd = t(lambda: (rt.hipMemcpy(d, pa, n, 3), rt.hipDeviceSynchronize()))
    wr = t(lambda: (rt.hipMemcpy(pb, d, n, 3), rt.hipDeviceSynchronize()))
    print(f"device-side copy from registered host memory {n/rd/1e9:5.1f} GB/s, to it {n/wr/1e9:5.1f} GB/s; data ok {bool((b == 1).all())}")
