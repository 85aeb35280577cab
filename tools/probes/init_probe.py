#!/usr/bin/env python3
"""GPU box: what the first 0.2 s of the counting thread are made of (library load, HIP runtime start, first allocation)"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t0 = time.perf_counter()
from jasper_amd import _lib
t1 = time.perf_counter()
if os.environ.get("NO_TORCH_RUNTIME"):
    _lib._share_hip_runtime_with_torch = lambda: None
L = _lib.lib()
t2 = time.perf_counter()
n = C.c_int(0)
L.jasper_device_count(C.byref(n))
t3 = time.perf_counter()
f, t = C.c_uint64(0), C.c_uint64(0)
L.jasper_device_mem_info(0, C.byref(f), C.byref(t))
t4 = time.perf_counter()
from jasper_amd import KmerTable
tb = KmerTable(37, min_slots=int(1.25 * 296e6))
t5 = time.perf_counter()
tb.sync()
t6 = time.perf_counter()
print("import _lib %.3f | dlopen (+ torch's HIP runtime: %s) %.3f | hipGetDeviceCount %.3f | hipSetDevice + hipMemGetInfo %.3f | table create (2^29 slots) %.3f | sync %.3f | total %.3f s"
      % (t1 - t0, "no" if os.environ.get("NO_TORCH_RUNTIME") else "yes", t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t6 - t0))
