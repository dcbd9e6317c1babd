"""Where do the 4 ms of a device tableau handle go?  Times the HIP calls lpx_tableau_create makes (via ctypes)."""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
hip.hipSetDevice(0)
p = C.c_void_p(); hip.hipMalloc(C.byref(p), 1 << 20); hip.hipFree(p)
def t(f, n=32):
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) * 1e3 / n
keep = []
def malloc(sz):
    q = C.c_void_p(); assert hip.hipMalloc(C.byref(q), C.c_size_t(sz)) == 0; keep.append(q)
def hostmalloc(sz):
    q = C.c_void_p(); assert hip.hipHostMalloc(C.byref(q), C.c_size_t(sz), 0) == 0
def stream():
    s = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0; keep.append(s)
print("hipMalloc 8.4 MB      %.3f ms" % t(lambda: malloc(8_400_000)))
print("hipMalloc 0.7 MB      %.3f ms" % t(lambda: malloc(700_000)))
print("hipHostMalloc 264 B   %.3f ms" % t(lambda: hostmalloc(264)))
print("hipStreamCreate       %.3f ms" % t(stream))
