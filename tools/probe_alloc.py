#!/usr/bin/env python3
"""tools/probe_alloc.py -- what hipMalloc / hipFree / hipMemset cost by size on this box (the command line's scan allocates
several hundred buffers of 0.1-1 GB)."""
import ctypes
import time

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
hip.hipDeviceSynchronize()
for mb in (16, 128, 1024, 4096):
    ptrs = []
    n = 40 if mb <= 1024 else 10
    t0 = time.time()
    for i in range(n):
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), mb << 20) == 0
        ptrs.append(p)
    t1 = time.time()
    for p in ptrs:
        hip.hipMemset(p, 0, mb << 20)
    hip.hipDeviceSynchronize()
    t2 = time.time()
    for p in ptrs:
        hip.hipFree(p)
    t3 = time.time()
    print("%5d MB x %d: hipMalloc %.2f ms each, first touch (memset) %.2f ms each = %.0f GB/s, hipFree %.2f ms each"
          % (mb, n, (t1 - t0) / n * 1e3, (t2 - t1) / n * 1e3, n * mb / 1024 / (t2 - t1), (t3 - t2) / n * 1e3), flush=True)
# again, after the frees: does memory that was used before come back slower?
for rep in range(2):
    t0 = time.time()
    ptrs = []
    for i in range(100):
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), 1 << 30) == 0
        ptrs.append(p)
    t1 = time.time()
    for p in ptrs:
        hip.hipFree(p)
    t2 = time.time()
    print("100 x 1 GB (100 GB held): hipMalloc %.2f ms each, hipFree %.2f ms each" % ((t1 - t0) * 10, (t2 - t1) * 10), flush=True)
