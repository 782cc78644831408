#!/usr/bin/env python3
"""Per-level time of k_resize_fused summed over workgroups (library built with -DFUSE_TRACE)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import conftest  # noqa
import my_slam_amd as M, my_slam_amd.synth as synth
W, H, n, B = (int(a) for a in (sys.argv[1:5] + ["640", "480", "1000", "64"][len(sys.argv) - 1:]))
fr = synth.stream(3, W, H, B)
e = M.ORBextractor(n, max_width=W, max_height=H, max_batch=B); e.set_batch_chunk(0)
L = M.lib(); L.orbx_debug_fuse_trace.argtypes = [C.c_void_p, C.c_int]
e.extract_batch(fr)
buf = np.zeros(16, np.uint64)
L.orbx_debug_fuse_trace(buf.ctypes.data, 1)
e.extract_batch(fr); torch.cuda.synchronize()
L.orbx_debug_fuse_trace(buf.ctypes.data, 0)
wg = int(buf[15])
print("workgroups", wg)
for i in range(8):
    if buf[i]: print("  fused level %d: %.2f us per workgroup" % (i, float(buf[i]) / 100.0 / max(wg, 1)))
