#!/usr/bin/env python3
"""Section timing of k_octree's workgroup (frame 0, level 0).  Needs a liborbx.so built with -DOCT_TRACE:
   make -C my-slam_amd clean && make -C my-slam_amd HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DOCT_TRACE"
usage: oct_trace.py W H nfeatures"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import conftest  # noqa
import my_slam_amd as M
import my_slam_amd.synth as synth
W, H, n = (int(a) for a in sys.argv[1:4])
img = synth.texture(2, W, H)
e = M.ORBextractor(n, max_width=W, max_height=H)
for _ in range(3):
    k, d = e(img)
buf = (C.c_ulonglong * 256)()
L = M.lib()
L.orbx_debug_oct_trace.argtypes = [C.c_void_p]
assert L.orbx_debug_oct_trace(buf) == 0
tp = buf[0] & 0xFFFFFFFF; ncand = buf[0] >> 32
names = {0: "start", 1: "roots", 2: "first count loop", 3: "S4 scan/cutoff", 4: "S5 children + S7 survivors", 5: "slots (phase A)",
         6: "slots (phase B)", 7: "key loop", 8: "final move", 9: "selection"}
print(f"{W}x{H} n={n}: level-0 candidates {ncand}, keypoints {len(k)}, {tp} stamps")
prev = None; tot = {}
for i in range(1, tp + 1):
    tag = buf[i] >> 56; t = buf[i] & ((1 << 56) - 1)
    if prev is not None:
        us = (t - prev) / 100.0
        tot[tag] = tot.get(tag, 0) + us
        print(f"  {names[tag]:30s} {us:7.2f} us")
    prev = t
print("totals:", {names[k]: round(v, 2) for k, v in tot.items()}, "sum", round(sum(tot.values()), 2))
