#!/usr/bin/env python3
"""BASELINE config 5 shape without the pose solvers: a Tracking-shaped per-frame loop on 1241x376 synthetic frames
(KITTI-odometry-seq-00 shape), nFeatures = 2000:
    extract (GPU) -> Frame grid (GPU) -> ComputeBoW (GPU descent + host maps)
    -> TrackReferenceKeyFrame's ORBmatcher::SearchByBoW (orbm_search_by_bow: TH_LOW, ratio 0.7, rotation filter)
    -> TrackWithMotionModel-style windowed search around the previous positions (GPU, TH_HIGH, rotation filter)
EPnP RANSAC and g2o pose optimisation (src/PnPsolver.cc, src/Optimizer.cc) are host-side dense fp64 solves (include/orbp.h); the
C++ harness tools/track/track_harness.cc runs them in the loop, this script reports the front-end + matching time per frame.
Everything goes through the host-buffer C ABI, i.e. PCIe transfers and Python dispatch are inside the numbers."""
import json, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa
import my_slam_amd as M
import my_slam_amd.synth as synth
from test_vocabulary import make_vocabulary

W, H, NF, K = 1241, 376, 2000, 40
frames = synth.stream(5, W, H, K)
tmp = tempfile.mkdtemp()
make_vocabulary(os.path.join(tmp, "voc.txt"), 10, 4, seed=1)          # 10^4 words, levelsup 2 -> 100 nodes
voc = M.ORBVocabulary(os.path.join(tmp, "voc.txt"))
ex = M.ORBextractor(NF, max_width=W, max_height=H)
mt = M.ORBmatcher(0.7, True, max_queries=4096, max_train=4096, max_pairs=1 << 21)
prev = None
t_stage = np.zeros(5)
nm_bow = nm_proj = 0
t_all = []
for k in range(K):
    t0 = time.perf_counter()
    kp, de = ex(frames[k]); t1 = time.perf_counter()
    mt.grid_build(kp, 0.0, float(W), 0.0, float(H)); t2 = time.perf_counter()
    bow, fv = voc.transform(de, 2); t3 = time.perf_counter()
    if prev is not None:
        pk, pd, pfv = prev
        # TrackReferenceKeyFrame: ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) with the reference's exact semantics
        # (distances of all node-mates on the GPU, the order-dependent selection on the host, inside the library)
        _, n_bow = mt.SearchByBoW(pk, pd, pfv, kp, de, fv)
        nm_bow += n_bow
        at = np.ascontiguousarray(kp["angle"])
        t4 = time.perf_counter()
        # TrackWithMotionModel shape: window 15 * scale around the previous position, octave +-1, TH_HIGH
        r = (15.0 * np.float32(1.2) ** pk["octave"]).astype(np.float32)
        bi, bd, sd = mt.search_area_best2(pd, pk["x"], pk["y"], r, np.maximum(pk["octave"] - 1, -1), pk["octave"] + 1, de)
        m12 = np.where(bd <= 100, bi, -1).astype(np.int32)
        aq = np.ascontiguousarray(pk["angle"])
        nm_proj += mt.L.orbm_rot_filter(aq.ctypes.data, at.ctypes.data, m12.ctypes.data, len(m12))
        t5 = time.perf_counter()
        if k >= 5:
            t_stage += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4]
            t_all.append(t5 - t0)
    prev = (kp, de, fv)
n = len(t_all)
out = {"shape": "%dx%d n=%d" % (W, H, NF), "frames_timed": n, "ms_per_frame_median": round(float(np.median(t_all)) * 1e3, 3),
       "frames_per_s": round(1.0 / float(np.median(t_all)), 1),
       "ms": dict(zip(["extract", "grid", "bow", "search_by_bow", "search_by_projection"], (t_stage / n * 1e3).round(3).tolist())),
       "matches_per_frame": {"bow": round(nm_bow / (K - 1), 1), "projection": round(nm_proj / (K - 1), 1)}}
print(json.dumps(out))
