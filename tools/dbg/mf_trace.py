#!/usr/bin/env python3
"""Workgroup timeline of k_best2_mfma (library built with -DMF_TRACE): when workgroups start, when their loads have landed, when they end."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import conftest  # noqa
import my_slam_amd as M
B, n, cap = 64, 1007, 1032
rng = np.random.default_rng(1)
desc = torch.from_numpy(rng.integers(0, 256, (B, cap, 32), dtype=np.uint8)).cuda()
kps = torch.zeros((B, cap, 7), device="cuda")
cnt = torch.full((B,), n, dtype=torch.int32, device="cuda")
m12 = torch.zeros((B, cap), dtype=torch.int32, device="cuda"); nm = torch.zeros(B, dtype=torch.int32, device="cuda")
mt = M.ORBmatcher(0.9, True, max_queries=cap, max_train=cap, max_pairs=1)
for _ in range(5):
    mt.match_batch_device(desc.data_ptr() + cap * 32, kps.data_ptr() + cap * 28, cnt.data_ptr() + 4, desc.data_ptr(), kps.data_ptr(), cnt.data_ptr(),
                          cap, B - 1, m12.data_ptr() + cap * 4, nm.data_ptr() + 4)
torch.cuda.synchronize()
L = M.lib(); L.orbm_debug_mf_trace.argtypes = [C.c_void_p]
buf = np.zeros(4 * 4096, np.uint64)
assert L.orbm_debug_mf_trace(buf.ctypes.data) == 0
t = buf.reshape(4096, 4).astype(np.int64)
t = t[(t[:, 0] > 0) & (t[:, 2] > 0)]
t0 = t[:, 0].min()
st, ld, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t[:, 0]) / 100.0, (t[:, 2] - t[:, 1]) / 100.0
print("workgroups with work: %d; kernel span %.1f us" % (len(t), (t[:, 2].max() - t0) / 100.0))
print("start time  us: p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(st, [10, 50, 90, 100])))
print("load phase  us: p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(ld, [10, 50, 90, 100])))
print("mfma phase  us: p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(en, [10, 50, 90, 100])))
