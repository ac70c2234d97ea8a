"""PCIe link rates of the box, pinned host memory: H2D alone, D2H alone, both at once on two streams (what the host-pointer entries can hope for)."""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so"); vp = C.c_void_p
def chk(r): assert r == 0, r
N = 1 << 30
h1, h2, d1, d2 = vp(), vp(), vp(), vp()
chk(hip.hipHostMalloc(C.byref(h1), C.c_size_t(N), 0)); chk(hip.hipHostMalloc(C.byref(h2), C.c_size_t(N), 0))
chk(hip.hipMalloc(C.byref(d1), C.c_size_t(N))); chk(hip.hipMalloc(C.byref(d2), C.c_size_t(N)))
C.memset(h1, 1, N); C.memset(h2, 2, N)
s1, s2 = vp(), vp(); chk(hip.hipStreamCreateWithFlags(C.byref(s1), 1)); chk(hip.hipStreamCreateWithFlags(C.byref(s2), 1))
def run(h2d, d2h, reps=4):
    chk(hip.hipDeviceSynchronize()); t0 = time.perf_counter()
    for _ in range(reps):
        if h2d: chk(hip.hipMemcpyAsync(d1, h1, C.c_size_t(N), 1, s1))
        if d2h: chk(hip.hipMemcpyAsync(h2, d2, C.c_size_t(N), 2, s2))
    chk(hip.hipDeviceSynchronize()); return (time.perf_counter() - t0) / reps
run(True, True, 1)
a, b, c = run(True, False), run(False, True), run(True, True)
print("pinned 1 GiB: H2D %.1f GB/s, D2H %.1f GB/s, both at once %.1f + %.1f GB/s (%.1f ms per pair)" % (N / a / 1e9, N / b / 1e9, N / c / 1e9, N / c / 1e9, c * 1e3))
# pageable source for comparison
import numpy as np
p = np.ones(N, dtype=np.uint8)
chk(hip.hipDeviceSynchronize()); t0 = time.perf_counter(); chk(hip.hipMemcpy(d1, p.ctypes.data_as(vp), C.c_size_t(N), 1)); t1 = time.perf_counter()
chk(hip.hipMemcpy(p.ctypes.data_as(vp), d1, C.c_size_t(N), 2)); t2 = time.perf_counter()
print("pageable 1 GiB: H2D %.1f GB/s, D2H %.1f GB/s" % (N / (t1 - t0) / 1e9, N / (t2 - t1) / 1e9))
t0 = time.perf_counter(); C.memmove(h2, h1, N); t1 = time.perf_counter()
print("host memcpy pinned->pinned, one thread: %.1f GB/s" % (N / (t1 - t0) / 1e9))
