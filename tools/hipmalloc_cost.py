#!/usr/bin/env python3
"""what a large hipMalloc / hipFree costs on the box (the graph unit allocates ~100 GB of scratch per run)"""
import ctypes, time
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
p = ctypes.c_void_p()
hip.hipMalloc(ctypes.byref(p), 1 << 20); hip.hipFree(p)          # context
for gb in (1, 4, 8, 16, 32):
    t0 = time.perf_counter(); rc = hip.hipMalloc(ctypes.byref(p), gb << 30); t1 = time.perf_counter()
    hip.hipMemset(p, 0, gb << 30); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
    hip.hipFree(p); t3 = time.perf_counter()
    print(f"{gb:3d} GiB: hipMalloc {1e3 * (t1 - t0):8.1f} ms (rc {rc}), first memset {1e3 * (t2 - t1):8.1f} ms, hipFree {1e3 * (t3 - t2):8.1f} ms")
