#!/usr/bin/env python3
"""Side measurements recorded in DESIGN.md §7 (never the bench.py `value`):
  * PCIe-inclusive rates of the host-buffer API (orbx_extract / orbx_extract_batch)
  * the other BASELINE shapes through the device API (1920x1080 n=4000, 1241x376 n=2000)
  * N1: windowed search (grid build + GetFeaturesInArea + best2) on the GPU vs the oracle on one core
"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import conftest  # noqa
import oracle_lib as O
import my_slam_amd as M
import my_slam_amd.synth as synth

out = {}
# ---- host API, PCIe inclusive ----
img = synth.texture(1, 640, 480)
ex = M.ORBextractor(1000, max_width=640, max_height=480, max_batch=64)
for _ in range(20):
    ex(img)
ts = []
for _ in range(300):
    t0 = time.perf_counter(); k, d = ex(img); ts.append(time.perf_counter() - t0)
out["host_api_single_640x480"] = {"median_ms": round(float(np.median(ts)) * 1e3, 4), "frames_per_s": round(1 / float(np.median(ts)), 1), "keypoints": len(k)}
ex1 = M.ORBextractor(1000, max_width=640, max_height=480)          # a handle for one frame at a time (the SLAM case): one HIP graph per call
for _ in range(20):
    ex1(img)
ts = []
for _ in range(300):
    t0 = time.perf_counter(); k, d = ex1(img); ts.append(time.perf_counter() - t0)
out["host_api_single_640x480_one_frame_handle"] = {"median_ms": round(float(np.median(ts)) * 1e3, 4), "frames_per_s": round(1 / float(np.median(ts)), 1), "keypoints": len(k)}
frames = synth.stream(4, 640, 480, 64)
for _ in range(3):
    ex.extract_batch(frames)
ts = []
for _ in range(20):
    t0 = time.perf_counter(); r = ex.extract_batch(frames); ts.append(time.perf_counter() - t0)
out["host_api_batch64_640x480"] = {"median_ms": round(float(np.median(ts)) * 1e3, 3), "frames_per_s": round(64 / float(np.median(ts)), 1),
                                   "note": "Python wrapper extract_batch: the C call plus 64 per-frame array slices built in Python"}
ts = []
for _ in range(20):
    t0 = time.perf_counter(); r = ex.extract_batch_raw(frames); ts.append(time.perf_counter() - t0)
out["host_api_batch64_640x480_c_call"] = {"median_ms": round(float(np.median(ts)) * 1e3, 3), "frames_per_s": round(64 / float(np.median(ts)), 1),
                                          "note": "orbx_extract_batch itself (flat output arrays), as bench.py's host_api"}
del ex

# ---- other shapes, device API ----
def dev_bench(W, H, n, B, seed):
    fr = torch.from_numpy(synth.stream(seed, W, H, B)).cuda()
    e = M.ORBextractor(n, max_width=W, max_height=H, max_batch=B); cap = e.cap
    k = torch.zeros((B, cap, 7), device="cuda"); d = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    c = torch.zeros(B, dtype=torch.int32, device="cuda"); s = torch.zeros(B, dtype=torch.int32, device="cuda")
    mt = M.ORBmatcher(0.9, True, max_queries=cap, max_train=cap, max_pairs=1)
    m12 = torch.zeros((B, cap), dtype=torch.int32, device="cuda"); nm = torch.zeros(B, dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    def step():
        e.extract_batch_device(fr.data_ptr(), B, W, H, fr.stride(1), fr.stride(0), k.data_ptr(), d.data_ptr(), c.data_ptr(), s.data_ptr(), st.cuda_stream)
        if B > 1:
            mt.match_batch_device(d.data_ptr() + cap * 32, k.data_ptr() + cap * 28, c.data_ptr() + 4, d.data_ptr(), k.data_ptr(), c.data_ptr(),
                                  cap, B - 1, m12.data_ptr() + cap * 4, nm.data_ptr() + 4, stream=st.cuda_stream)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 20
    assert int(s.abs().sum()) == 0
    e.set_profiling(True); acc = np.zeros(4)
    for _ in range(5):
        e.extract_batch_device(fr.data_ptr(), B, W, H, fr.stride(1), fr.stride(0), k.data_ptr(), d.data_ptr(), c.data_ptr(), s.data_ptr(), st.cuda_stream)
        acc += e.stage_ms()
    return {"ms_per_step": round(el * 1e3, 4), "frames_per_s": round(B / el, 1), "keypoints_per_s": round(float(c.sum()) / el, 1),
            "matches_per_step": int(nm.sum()), "stage_ms": [round(float(v), 4) for v in acc / 5]}
out["device_1920x1080_n4000_b8"] = dev_bench(1920, 1080, 4000, 8, 2)
out["device_1920x1080_n4000_b1"] = dev_bench(1920, 1080, 4000, 1, 2)
out["device_1241x376_n2000_b32"] = dev_bench(1241, 376, 2000, 32, 5)
out["device_640x480_n1000_b1"] = dev_bench(640, 480, 1000, 1, 4)

# ---- a stereo pair (Frame.cc:78-81 extracts left and right on two threads): two handles, blocking one after the other
# vs both in flight through orbx_extract_begin / orbx_extract_end ----
fl, fr = synth.frame_pair(7, 1241, 376, shift=(9, 0))
el = M.ORBextractor(2000, max_width=1241, max_height=376); er = M.ORBextractor(2000, max_width=1241, max_height=376)
for _ in range(10):
    el(fl); er(fr)
ts = []
for _ in range(100):
    t0 = time.perf_counter(); a = el(fl); b = er(fr); ts.append(time.perf_counter() - t0)
seq = float(np.median(ts))
ts = []
for _ in range(100):
    t0 = time.perf_counter(); el.extract_begin(fl); er.extract_begin(fr); a2 = el.extract_end(); b2 = er.extract_end(); ts.append(time.perf_counter() - t0)
ovl = float(np.median(ts))
assert a2[0].tobytes() == a[0].tobytes() and np.array_equal(b2[1], b[1])
out["stereo_pair_1241x376_n2000"] = {"ms_blocking_left_then_right": round(seq * 1e3, 4), "ms_both_in_flight": round(ovl * 1e3, 4)}
del el, er

# ---- N1 windowed search ----
f0, f1 = synth.frame_pair(2, 1241, 376)
e = M.ORBextractor(2000, max_width=1241, max_height=376)
k0, d0 = e(f0); k1, d1 = e(f1)
mt = M.ORBmatcher(0.9, True, max_queries=4096, max_train=4096, max_pairs=1 << 21)
x = k0["x"].copy(); y = k0["y"].copy(); r = (15.0 * np.float32(1.2) ** k0["octave"]).astype(np.float32)
mn = np.maximum(k0["octave"] - 1, -1).astype(np.int32); mx = (k0["octave"] + 1).astype(np.int32)
for _ in range(3):
    mt.grid_build(k1, 0.0, 1241.0, 0.0, 376.0); mt.search_area_best2(d0, x, y, r, mn, mx, d1)
ts = []
for _ in range(30):
    t0 = time.perf_counter(); mt.grid_build(k1, 0.0, 1241.0, 0.0, 376.0); bi, bd, sd = mt.search_area_best2(d0, x, y, r, mn, mx, d1); ts.append(time.perf_counter() - t0)
dq = torch.from_numpy(d0).cuda(); dt = torch.from_numpy(d1).cuda()
dx, dy, dr = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(r).cuda()
dmn, dmx = torch.from_numpy(mn).cuda(), torch.from_numpy(mx).cuda()
o = [torch.zeros(len(x), dtype=torch.int32, device="cuda") for _ in range(3)]
st = torch.cuda.current_stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tk = []
for i in range(30):
    e0.record()
    M._mchk(mt.L.orbm_search_area_best2_device(mt.h, dq.data_ptr(), dx.data_ptr(), dy.data_ptr(), dr.data_ptr(), dmn.data_ptr(), dmx.data_ptr(), len(x),
                                               dt.data_ptr(), None, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), st.cuda_stream))
    e1.record(); torch.cuda.synchronize(); tk.append(e0.elapsed_time(e1))
og = O.FrameGrid(k1, 0.0, 1241.0, 0.0, 376.0)
t0 = time.perf_counter()
for _ in range(5):
    og2 = O.FrameGrid(k1, 0.0, 1241.0, 0.0, 376.0); obi, obd, osd = og2.search_area_best2(d0, x, y, r, mn, mx, d1)
tc = (time.perf_counter() - t0) / 5
assert np.array_equal(bi, obi) and np.array_equal(bd, obd)
out["n1_window_search_1241x376"] = {"queries": len(x), "train": len(k1), "candidates": int(mt.GetFeaturesInArea(x, y, r, mn, mx)[0][-1]),
                                    "gpu_host_api_ms": round(float(np.median(ts)) * 1e3, 4), "gpu_kernel_ms": round(float(np.median(tk)), 4),
                                    "cpu_oracle_ms_1thread": round(tc * 1e3, 3)}
print(json.dumps(out, indent=1))
