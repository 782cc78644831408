#!/usr/bin/env python3
"""Device-API steps of one shape for rocprofv3 --kernel-trace --stats (the other BASELINE shapes beside bench.py's):
usage: prof_shape.py W H nfeatures batch [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import conftest  # noqa
import my_slam_amd as M
import my_slam_amd.synth as synth
W, H, n, B = (int(a) for a in sys.argv[1:5])
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
fr = torch.from_numpy(synth.stream(2, W, H, B)).cuda()
e = M.ORBextractor(n, max_width=W, max_height=H, max_batch=B); cap = e.cap
k = torch.zeros((B, cap, 7), device="cuda"); d = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
c = torch.zeros(B, dtype=torch.int32, device="cuda"); s = torch.zeros(B, dtype=torch.int32, device="cuda")
mt = M.ORBmatcher(0.9, True, max_queries=cap, max_train=cap, max_pairs=1)
m12 = torch.zeros((B, cap), dtype=torch.int32, device="cuda"); nm = torch.zeros(B, dtype=torch.int32, device="cuda")
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
for _ in range(steps):
    e.extract_batch_device(fr.data_ptr(), B, W, H, fr.stride(1), fr.stride(0), k.data_ptr(), d.data_ptr(), c.data_ptr(), s.data_ptr(), st.cuda_stream)
    if B > 1:
        mt.match_batch_device(d.data_ptr() + cap * 32, k.data_ptr() + cap * 28, c.data_ptr() + 4, d.data_ptr(), k.data_ptr(), c.data_ptr(),
                              cap, B - 1, m12.data_ptr() + cap * 4, nm.data_ptr() + 4, stream=st.cuda_stream)
torch.cuda.synchronize()
assert int(s.abs().sum()) == 0
print("ok", int(c.sum()), int(nm.sum()))
