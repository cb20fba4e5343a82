"""Times hipMalloc / first touch of large buffers (tools probe, not product code)."""
import ctypes, sys, time
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipSetDevice(0)
def one(gb, keep=False):
    p = ctypes.c_void_p()
    t0 = time.time()
    rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(int(gb * (1 << 30))))
    t1 = time.time()
    hip.hipMemset(p, 0, ctypes.c_size_t(int(gb * (1 << 30)))); hip.hipDeviceSynchronize()
    t2 = time.time()
    if not keep:
        hip.hipFree(p)
    print("hipMalloc %6.1f GB rc=%d %.3f s, memset %.3f s, free %.3f s" % (gb, rc, t1 - t0, t2 - t1, time.time() - t2), flush=True)
    return p
for gb in (1, 4, 16, 30, 32, 48, 64, 80, 30, 80, 114, 30):
    one(gb)
print("-- kept allocations (like a build): 80, 30, 2.5, 0.2")
ps = [one(g, keep=True) for g in (80, 30, 2.5, 0.2)]
print("-- one block of 113 GB instead")
for p in ps: hip.hipFree(p)
one(113, keep=True)
