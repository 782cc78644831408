#!/usr/bin/env python3
"""Per-phase shader-clock split of k_fast_cells / k_describe (liborbx.so built with -DORBX_TRACE).
usage: phase_trace.py W H nfeatures batch"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import conftest  # noqa
import my_slam_amd as M
import my_slam_amd.synth as synth
W, H, n, B = (int(a) for a in sys.argv[1:5])
fr = synth.stream(3, W, H, B)
e = M.ORBextractor(n, max_width=W, max_height=H, max_batch=B)
L = M.lib()
have = [nm for nm in ("orbx_debug_fast_trace", "orbx_debug_desc_trace") if hasattr(L, nm)]
for nm in have:
    getattr(L, nm).argtypes = [C.c_void_p, C.c_int]
e.extract_batch(fr)
buf = (C.c_ulonglong * 8)()
for nm in have:
    getattr(L, nm)(buf, 1)
e.extract_batch(fr)
torch.cuda.synchronize()
for name, nm, ph in (("fast", "orbx_debug_fast_trace", ["header + tile load", "stage 1 (reject + queue)", "stage 2 (scores)", "NMS", "output"]),
                     ("describe", "orbx_debug_desc_trace", ["header + patch load", "IC angle", "blur rows", "blur columns", "sincos + rBRIEF + store"])):
    if nm not in have:
        continue
    fn = getattr(L, nm)
    assert fn(buf, 0) == 0
    tot = sum(buf[i] for i in range(6)); waves = buf[7]
    print(f"{name}: {waves} waves, {tot / max(waves, 1):.0f} clocks per wave")
    for i, p in enumerate(ph):
        print(f"   {p:28s} {100.0 * buf[i] / max(tot, 1):5.1f} %   {buf[i] / max(waves, 1):8.0f} clk/wave")
