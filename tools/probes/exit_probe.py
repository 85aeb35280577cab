#!/usr/bin/env python3
"""GPU box: what a process's exit costs as a function of the device memory it holds (the parent times the child's start-to-end,
the child reports when it called os._exit).  python tools/probes/exit_probe.py"""
import ctypes, os, subprocess, sys, time
if len(sys.argv) > 2 and sys.argv[1] == "child":
    gb = float(sys.argv[2]); touch = sys.argv[3] == "1"
    hip = ctypes.CDLL("libamdhip64.so")
    t0 = time.time()
    n = ctypes.c_int(0); hip.hipGetDeviceCount(ctypes.byref(n))
    t1 = time.time()
    ptrs = []
    left = gb
    while left > 0:
        sz = min(left, 16.0)
        p = ctypes.c_void_p()
        rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(int(sz * (1 << 30))))
        assert rc == 0, rc
        if touch:
            hip.hipMemset(p, 0, ctypes.c_size_t(int(sz * (1 << 30))))
        ptrs.append(p); left -= sz
    hip.hipDeviceSynchronize()
    t2 = time.time()
    sys.stdout.write("%.6f %.3f %.3f\n" % (time.time(), t1 - t0, t2 - t1)); sys.stdout.flush()
    os._exit(0)
for gb, touch in ((0.001, 0), (4, 0), (4, 1), (16, 1), (48, 0), (48, 1), (4, 1), (0.001, 0)):
    time.sleep(2.0)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(gb), str(touch)], capture_output=True, text=True)
    t1 = time.time()
    f = p.stdout.split()
    print("%6.3f GB touched %d: init %s s, malloc(+memset) %s s, exit %.3f s, whole %.3f s" % (gb, touch, f[1], f[2], t1 - float(f[0]), t1 - t0), flush=True)
