#!/usr/bin/env python3
"""Stage times of one batched extraction for the library ORBX_LIB names (A/B builds: tools/dbg/ab_build.sh).
usage: ORBX_LIB=... ab_run.py [W H nfeatures batch reps]   -> one JSON line; also checks frame 0 against the oracle."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import conftest  # noqa
import my_slam_amd as M
import my_slam_amd.synth as synth
import oracle_lib as O
a = [int(x) for x in sys.argv[1:6]] + [640, 480, 1000, 64, 20][len(sys.argv) - 1:]
W, H, n, B, reps = a[:5]
fr = synth.stream(4, W, H, B)
frames = torch.from_numpy(fr).cuda()
ex = M.ORBextractor(n, 1.2, 8, 20, 7, device=0, max_width=W, max_height=H, max_batch=B)
cap = ex.cap
kps = torch.zeros((B, cap, 7), device="cuda"); desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
cnt = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); s = stream.cuda_stream
def run():
    ex.extract_batch_device(frames.data_ptr(), B, W, H, frames.stride(1), frames.stride(0), kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), st.data_ptr(), s)
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(stream)
for _ in range(reps): run()
e1.record(stream); torch.cuda.synchronize()
whole = e0.elapsed_time(e1) / reps
ex.set_profiling(True)
acc = np.zeros(4)
for _ in range(reps):
    run(); acc += ex.stage_ms()
acc /= reps
ok = None
if os.environ.get("AB_CHECK", "1") == "1":
    ok = True
    for f in (0, B - 1):
        okp, od, _ = O.Extractor(n).extract(fr[f])
        c = int(cnt[f].item())
        k = kps[f, :c].cpu().numpy().view(np.uint8).reshape(c, 28)
        ok = ok and c == len(okp) and k.tobytes() == okp.tobytes() and np.array_equal(desc[f, :c].cpu().numpy(), od)
print(json.dumps({"lib": os.path.basename(M.LIB_PATH), "shape": [W, H, n, B], "extract_ms": round(whole, 4), "status": int(st.abs().sum().item()),
                  "stage_ms": {k: round(float(v), 4) for k, v in zip(("pyramid", "fast", "quadtree", "describe"), acc)}, "oracle_equal": ok}))
