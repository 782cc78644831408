#!/usr/bin/env python3
"""Stream a known byte count through 4-byte loads, 16-byte loads and 4-byte stores (run under
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE by tools/prof_traffic.sh) to calibrate the traffic counters."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import conftest  # noqa
import my_slam_amd as M
L = M.lib()
L.orbx_debug_stream.argtypes = [C.c_size_t, C.c_int, C.c_int]
L.orbx_debug_stream.restype = C.c_int
BYTES = 1 << 30
for mode in (0, 1, 2):
    rc = L.orbx_debug_stream(BYTES, mode, 3)
    print("mode", mode, "rc", rc, "bytes", BYTES)
