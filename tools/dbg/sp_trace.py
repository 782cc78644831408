#!/usr/bin/env python3
"""Per-phase shader-clock split of k_best2_mfma_sp (orbm_mfma.hip built with -DSP_TRACE): one bench-shaped dense match launch."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import conftest  # noqa
import my_slam_amd as M
B, cap = 63, 1007
rng = np.random.default_rng(5)
desc = torch.from_numpy(rng.integers(0, 256, (B + 1, cap, 32), dtype=np.uint8)).cuda()
kps = torch.zeros((B + 1) * cap * M.KP_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
n = torch.full((B + 1,), cap, dtype=torch.int32, device="cuda")
m12 = torch.zeros((B, cap), dtype=torch.int32, device="cuda"); nm = torch.zeros(B, dtype=torch.int32, device="cuda")
mt = M.ORBmatcher(0.9, False, max_queries=cap, max_train=cap, max_pairs=1)
L = M.lib(); fn = L.orbm_debug_sp_trace; fn.argtypes = [C.c_void_p, C.c_int]
KP = M.KP_DTYPE.itemsize
def run():
    mt.match_batch_device(desc[1:].data_ptr(), kps.data_ptr() + cap * KP, n[1:].data_ptr(), desc.data_ptr(), kps.data_ptr(), n.data_ptr(), cap, B, m12.data_ptr(), nm.data_ptr())
    torch.cuda.synchronize()
run(); run()
buf = (C.c_ulonglong * 8)(); fn(buf, 1); run(); assert fn(buf, 0) == 0
wg = max(buf[7], 1)
names = ["prologue", "tiles (MFMA + select)", "expand (+ wait for loads)", "barrier", "epilogue"]
tot = sum(buf[i] for i in range(5))
print(f"{buf[7]} workgroups, {tot / wg:.0f} clocks each")
for i, nme in enumerate(names): print(f"   {nme:28s} {buf[i] / wg:8.0f} clk  {100.0 * buf[i] / max(tot, 1):5.1f} %")

sp = (C.c_ulonglong * 2048)(); L.orbm_debug_sp_span.argtypes = [C.c_void_p]; assert L.orbm_debug_sp_span(sp) == 0
a = np.frombuffer(sp, dtype=np.uint64).reshape(-1, 2); a = a[(a[:, 0] > 0) & (a[:, 1] > 0)]
t0 = a[:, 0].min(); st = (a[:, 0] - t0) / 100.0; en = (a[:, 1] - t0) / 100.0
print(f"   {len(a)} workgroups: start p50 {np.median(st):.1f} p90 {np.percentile(st, 90):.1f} max {st.max():.1f} us; end p50 {np.median(en):.1f} max {en.max():.1f}; life p50 {np.median(en - st):.1f}")
for t in np.arange(0, en.max(), en.max() / 8): print(f"      t = {t:5.1f} us: {int(((st <= t) & (en > t)).sum())} resident")
