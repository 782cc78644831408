#!/usr/bin/env python3
"""Dense batched match only (64 x 1007 descriptors against the previous frame), for kernel timing under rocprofv3.
usage: ORBX_LIB=... ab_match.py [nframes n cap reps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import conftest  # noqa
import my_slam_amd as M
a = [int(x) for x in sys.argv[1:5]] + [64, 1007, 1032, 20][len(sys.argv) - 1:]
B, n, cap, reps = a[:4]
rng = np.random.default_rng(1)
desc = torch.from_numpy(rng.integers(0, 256, (B, cap, 32), dtype=np.uint8)).cuda()
kps = torch.zeros((B, cap, 7), device="cuda"); kps[:, :, 3] = torch.rand((B, cap), device="cuda") * 359.0
cnt = torch.full((B,), n, dtype=torch.int32, device="cuda")
m12 = torch.zeros((B, cap), dtype=torch.int32, device="cuda"); nm = torch.zeros(B, dtype=torch.int32, device="cuda")
mt = M.ORBmatcher(0.9, True, max_queries=cap, max_train=cap, max_pairs=1)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
for _ in range(reps):
    mt.match_batch_device(desc.data_ptr() + cap * 32, kps.data_ptr() + cap * 28, cnt.data_ptr() + 4, desc.data_ptr(), kps.data_ptr(), cnt.data_ptr(),
                          cap, B - 1, m12.data_ptr() + cap * 4, nm.data_ptr() + 4, stream=st.cuda_stream)
torch.cuda.synchronize()
print(os.path.basename(M.LIB_PATH), int(nm.sum()))
