#!/usr/bin/env python3
"""Per-level shader-clock split of k_resize_tiles (liborbx.so built with -DORBX_TRACE).  usage: tile_trace.py W H nfeatures batch"""
import ctypes as C, os, sys
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
fn = L.orbx_debug_tile_trace
fn.argtypes = [C.c_void_p, C.c_int]
d_fr = torch.from_numpy(fr).cuda()
cap = e.cap
KP = M.KP_DTYPE.itemsize
d_k = torch.zeros(B * cap * KP, dtype=torch.uint8, device="cuda"); d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
d_c = torch.zeros(B, dtype=torch.int32, device="cuda"); d_s = torch.zeros(B, dtype=torch.int32, device="cuda")
def run():
    e.extract_batch_device(d_fr.data_ptr(), B, W, H, d_fr.stride(1), d_fr.stride(0), d_k.data_ptr(), d_d.data_ptr(), d_c.data_ptr(), d_s.data_ptr(), None)
    torch.cuda.synchronize()
run(); run()
buf = (C.c_ulonglong * 8)()
fn(buf, 1)
run()
assert fn(buf, 0) == 0
tot = sum(buf[i] for i in range(6)); waves = buf[7]
print(f"tiles: {waves} waves, {tot / max(waves, 1):.0f} clocks per wave (100 MHz ticks if s_memtime is the constant clock)")
for i in range(6):
    print(f"   slot {i} {'writeout' if i == 5 else 'level a+%d' % (i + 1):12s} {100.0 * buf[i] / max(tot, 1):5.1f} %   {buf[i] / max(waves, 1):8.0f} clk/wave")

import numpy as np
sp = (C.c_ulonglong * (3 * 8192))()
L.orbx_debug_tile_span.argtypes = [C.c_void_p]
assert L.orbx_debug_tile_span(sp) == 0
a = np.frombuffer(sp, dtype=np.uint64).reshape(-1, 3)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
st = (a[:, 0] - t0).astype(np.float64) / 100.0; en = (a[:, 1] - t0).astype(np.float64) / 100.0      # us
print(f"waves {len(a)}: start p50 {np.median(st):.1f} p90 {np.percentile(st, 90):.1f} max {st.max():.1f} us; end p50 {np.median(en):.1f} max {en.max():.1f}; life p50 {np.median(en - st):.1f} p90 {np.percentile(en - st, 90):.1f}")
for t in np.arange(0, en.max(), en.max() / 12):
    print(f"   t = {t:5.1f} us: {int(((st <= t) & (en > t)).sum())} waves resident")
hw = a[:, 2] & 0xFFFFFFFF; xcc = (a[:, 2] >> 32) & 0xF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3
key = xcc * 1000 + se * 100 + sh * 20 + cu
u, c = np.unique(key, return_counts=True)
print(f"distinct (xcc, se, sh, cu): {len(u)}; waves per CU min {c.min()} p50 {int(np.median(c))} max {c.max()}")
