#!/usr/bin/env python3
"""Randomised GPU-vs-oracle sweep of the six LocalMapping / LoopClosing matchers (src/ORBmatcher.cc:290-403, 522-655, 657-823,
825-975, 977-1100, 1102-1326): random image sizes, feature counts, camera steps, Sim3 scales and small rotations, MapPoint masks,
stereo masks, vocabularies and node granularities, key-frame grids with equal and with different assign / query origins.  Every case
runs the C-ABI pipeline (host projection helpers -> PredictScale -> search on the GPU) against the oracle's single restatement and
requires identical tables and counts.  usage: stress_kf.py [seconds] [seed]"""
import os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import conftest  # noqa
import oracle_lib as O
import my_slam_amd as M
import my_slam_amd.synth as synth
import test_vocabulary as TV

F32 = np.float32
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); n_ok = [0] * 6
tmp = tempfile.mkdtemp(prefix="stress_kf_")
vocs = {}


def rot_y(deg):
    a = np.deg2rad(deg)
    return np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])


while time.time() - t0 < budget:
    W = int(rng.integers(320, 1300)); H = int(rng.integers(200, 720)); nf = int(rng.choice([500, 1000, 2000]))
    sh = int(rng.integers(1, 5))
    shifts = (sh, 2 * sh, 3 * sh)
    frames, layer = synth.stream_layers(int(rng.integers(1, 1 << 20)), W, H, 2, shifts=shifts)
    try:
        ex = M.ORBextractor(nf, max_width=W, max_height=H)
        k = [None, None]; d = [None, None]
        k[0], d[0] = ex(frames[0]); k[1], d[1] = ex(frames[1])
    except M.OrbxError:
        continue
    if len(k[0]) < 16 or len(k[1]) < 16:
        continue
    sf, sig2, isig2 = ex.GetScaleFactors(), ex.GetScaleSigmaSquares(), ex.GetInverseScaleSigmaSquares()
    logsf = float(np.log(F32(1.2)))
    fx = fy = float(rng.uniform(300, 900)); cx, cy = W / 2.0 + float(rng.uniform(-20, 20)), H / 2.0; base = 0.5
    K4 = (fx, fy, cx, cy)
    T = [np.eye(4, dtype=F32), np.eye(4, dtype=F32)]; T[1][0, 3] = -base
    xw, nrm, mfx, mni, mxi = [], [], [], [], []
    for i in (0, 1):
        kk = k[i]
        Z = (fx * base / np.array(shifts, np.float64))[layer[np.clip(np.rint(kk["y"]).astype(int), 0, H - 1), np.clip(np.rint(kk["x"]).astype(int), 0, W - 1)]]
        xc = np.stack([(kk["x"] - cx) * Z / fx, (kk["y"] - cy) * Z / fy, Z], 1)
        xw.append((xc - T[i][:3, 3].astype(np.float64)).astype(F32))
        dr = np.linalg.norm(xc, axis=1)
        n_ = (xc / dr[:, None]).astype(F32)
        flip = rng.random(len(kk)) < 0.08
        n_[flip] = -n_[flip]
        nrm.append(n_)
        mf = (dr * sf[kk["octave"]] * rng.uniform(0.7, 1.4, len(kk))).astype(F32)
        mfx.append(mf); mni.append((F32(0.8) * (mf / sf[-1]).astype(F32)).astype(F32)); mxi.append((F32(1.2) * mf).astype(F32))
    # key-frame grid: equal origins, or Frame's fractional bounds against KeyFrame's truncated ints
    if rng.integers(0, 2):
        grid = (0.0, 0.0, float(F32(64) / F32(W)), float(F32(48) / F32(H)), 0.0, 0.0); bounds = (0.0, float(W), 0.0, float(H))
    else:
        mnx, mxx, mny, mxy = F32(-rng.uniform(0.2, 9)), F32(W + rng.uniform(0.2, 9)), F32(-rng.uniform(0.2, 9)), F32(H + rng.uniform(0.2, 9))
        grid = (float(mnx), float(mny), float(F32(64) / (mxx - mnx)), float(F32(48) / (mxy - mny)), float(int(mnx)), float(int(mny)))
        bounds = (float(int(mnx)), float(int(mxx)), float(int(mny)), float(int(mxy)))
    n0, n1 = len(k[0]), len(k[1])
    ratio = float(rng.choice([0.6, 0.75, 0.9])); ori = bool(rng.integers(0, 2))
    m = M.ORBmatcher(ratio, ori, max_queries=int(rng.choice([64, 4096])), max_train=int(rng.choice([64, 4096])), max_pairs=int(rng.choice([256, 1 << 21])))
    og = [O.KeyFrameGrid(k[0], grid), O.KeyFrameGrid(k[1], grid)]
    s = float(rng.choice([1.0, 1.0, rng.uniform(0.9, 1.1)])); rot = float(rng.choice([0.0, rng.uniform(-0.5, 0.5)]))
    S = np.eye(4); S[:3, :3] = s * rot_y(rot); S[:3, 3] = s * np.array([-base, 0.0, float(rng.choice([0.0, rng.uniform(-3, 3)]))]); S = S.astype(F32)
    # 1. SearchByProjection(KF2, Scw, points of KF1, vpMatched, th)
    usable = (rng.random(n0) < 0.85).astype(np.uint8)
    th_i = int(rng.choice([4, 10]))
    m0 = (rng.random(n1) < 0.2).astype(np.uint8); ma, mb = m0.copy(), m0.copy()
    m.grid_build_kf(k[1], grid)
    Tc, Ow = M.ORBmatcher.Sim3Decompose(S)
    u, v, iz, d3, ok = M.ORBmatcher.ProjectPointsKF(Tc, K4, bounds, xw[0], nrm[0], Ow)
    use = usable.astype(bool) & ok.astype(bool) & ~(d3 < mni[0]) & ~(d3 > mxi[0])
    lv = M.ORBmatcher.PredictScale(mfx[0], d3, logsf, len(sf))
    a, na = m.SearchByProjectionSim3(use.astype(np.uint8), u, v, lv, d[0], sf, k[1], d[1], ma, th_i)
    b, nb = O.search_by_projection_sim3(usable, xw[0], nrm[0], mni[0], mxi[0], mfx[0], d[0], S, K4, bounds, sf, logsf, og[1], d[1], mb, th_i)
    assert na == nb and np.array_equal(a, b) and np.array_equal(ma, mb), ("proj_sim3", W, H, nf, s, rot, th_i)
    n_ok[0] += 1
    # 2. Fuse(KF2, Scw, points of KF1, th)
    th_f = float(rng.choice([2.5, 4.0]))
    a, na = m.FuseSim3(use.astype(np.uint8), u, v, lv, d[0], sf, k[1], d[1], th_f)
    b, nb = O.fuse_sim3(usable, xw[0], nrm[0], mni[0], mxi[0], mfx[0], d[0], S, K4, bounds, sf, logsf, og[1], d[1], th_f)
    assert na == nb and np.array_equal(a, b), ("fuse_sim3", W, H, nf, s, rot, th_f)
    n_ok[1] += 1
    # 3. Fuse(KF2, points of KF1, th) with stereo / mono key-frame features
    Tk = T[1].copy(); Tk[2, 3] = float(rng.choice([0.0, rng.uniform(-4, 4)]))
    Owk = O.camera_center(Tk); bf = F32(fx * base)
    urk = np.full(n1, -1, F32); st = rng.random(n1) < 0.5
    Zc = xw[1][:, 2] + Tk[2, 3]
    urk[st] = (k[1]["x"] - bf / Zc)[st].astype(F32); urk[st & (rng.random(n1) < 0.2)] += 5.0
    th_k = float(rng.choice([2.0, 3.0, 5.0]))
    u, v, iz, d3, ok = M.ORBmatcher.ProjectPointsKF(Tk, K4, bounds, xw[0], nrm[0], Owk)
    use = usable.astype(bool) & ok.astype(bool) & ~(d3 < mni[0]) & ~(d3 > mxi[0])
    lv = M.ORBmatcher.PredictScale(mfx[0], d3, logsf, len(sf))
    ur = (u - bf * iz).astype(F32)
    a, na = m.Fuse(use.astype(np.uint8), u, v, ur, lv, d[0], sf, isig2, k[1], urk, d[1], th_k)
    b, nb = O.fuse(usable, xw[0], nrm[0], mni[0], mxi[0], mfx[0], d[0], Tk, Owk, K4, float(bf), bounds, sf, isig2, logsf, og[1], urk, d[1], th_k)
    assert na == nb and np.array_equal(a, b), ("fuse", W, H, nf, th_k, float(Tk[2, 3]))
    n_ok[2] += 1
    # 4. SearchBySim3(KF1, KF2, ...)
    s12 = float(rng.choice([1.0, rng.uniform(0.95, 1.05)])); R12 = rot_y(float(rng.choice([0.0, rng.uniform(-0.3, 0.3)]))).astype(F32)
    t12 = np.array([base, 0.0, float(rng.choice([0.0, rng.uniform(-2, 2)]))], F32)
    us1 = (rng.random(n0) < 0.85).astype(np.uint8); us2 = (rng.random(n1) < 0.85).astype(np.uint8)
    th_s = float(rng.choice([4.0, 7.5]))
    sR12, sR21, t21 = M.ORBmatcher.Sim3Relative(s12, R12, t12)
    u1, v1, d1_, ok1 = M.ORBmatcher.ProjectPointsSim3(T[0], sR21, t21, K4, bounds, xw[0])
    u2, v2, d2_, ok2 = M.ORBmatcher.ProjectPointsSim3(T[1], sR12, t12, K4, bounds, xw[1])
    use1 = us1.astype(bool) & ok1.astype(bool) & ~(d1_ < mni[0]) & ~(d1_ > mxi[0])
    use2 = us2.astype(bool) & ok2.astype(bool) & ~(d2_ < mni[1]) & ~(d2_ > mxi[1])
    l1 = M.ORBmatcher.PredictScale(mfx[0], d1_, logsf, len(sf)); l2 = M.ORBmatcher.PredictScale(mfx[1], d2_, logsf, len(sf))
    a, na = m.SearchBySim3(use1, u1, v1, l1, d[0], use2, u2, v2, l2, d[1], k[0], d[0], grid, sf, k[1], d[1], grid, sf, th_s)
    side = lambda i, us: dict(usable=us, xw=xw[i], min_inv=mni[i], max_inv=mxi[i], mf_max=mfx[i], mp_desc=d[i], Tw=T[i], bounds=bounds, sf=sf,
                              log_sf=logsf, grid=og[i], desc=d[i])
    b, nb = O.search_by_sim3(side(0, us1), side(1, us2), K4, s12, R12, t12, th_s)
    assert na == nb and np.array_equal(a, b), ("sim3", W, H, nf, s12, th_s)
    n_ok[3] += 1
    # 5 + 6. the vocabulary-guided pair: SearchByBoW(KF1, KF2) and SearchForTriangulation
    kd = (int(rng.choice([6, 10])), int(rng.choice([2, 3])))
    if kd not in vocs:
        path = os.path.join(tmp, "voc_%d_%d.txt" % kd)
        TV.make_vocabulary(path, kd[0], kd[1], seed=kd[0] + kd[1])
        vocs[kd] = M.ORBVocabulary(path)
    lu = int(rng.integers(0, kd[1] + 1))
    fv = [vocs[kd].transform(d[i], lu)[1] for i in (0, 1)]
    v1 = (rng.random(n0) < 0.75).astype(np.uint8); v2 = (rng.random(n1) < 0.75).astype(np.uint8)
    a, na = m.SearchByBoWKF(k[0], d[0], fv[0], v1, k[1], d[1], fv[1], v2)
    b, nb = O.search_by_bow_kf(d[0], k[0]["angle"], v1, fv[0], d[1], k[1]["angle"], v2, fv[1], ratio, ori)
    assert na == nb and np.array_equal(a, b), ("bow_kf", W, H, nf, kd, lu, ratio, ori)
    n_ok[4] += 1
    h1 = (rng.random(n0) < 0.3).astype(np.uint8); h2 = (rng.random(n1) < 0.3).astype(np.uint8)
    ur1 = np.where(rng.random(n0) < 0.5, k[0]["x"] - 20.0, -1.0).astype(F32); ur2 = np.where(rng.random(n1) < 0.5, k[1]["x"] - 20.0, -1.0).astype(F32)
    Kinv = np.linalg.inv(np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]]))
    F12 = (Kinv.T @ np.array([[0, 0, 0], [0, 0, -base], [0, base, 0]]) @ Kinv).astype(F32)
    Cw = np.array([0.0, 0.0, float(rng.choice([0.0, 20.0, 60.0]))], F32)          # some epipoles inside image 2
    only_st = bool(rng.integers(0, 3) == 0)
    a, na = m.SearchForTriangulation(k[0], d[0], h1, ur1, fv[0], k[1], d[1], h2, ur2, fv[1], Cw, T[1], K4, F12, sf, sig2, only_st)
    b, nb = O.search_for_triangulation(k[0], d[0], h1, ur1, fv[0], k[1], d[1], h2, ur2, fv[1], Cw, T[1], K4, F12, sf, sig2, only_st, ori)
    assert na == nb and np.array_equal(a, b), ("triangulation", W, H, nf, kd, lu, only_st, ori)
    n_ok[5] += 1
print("stress_kf: %d / %d / %d / %d / %d / %d random cases (SearchByProjection(KF, Scw) / Fuse(KF, Scw) / Fuse(KF, points) / SearchBySim3 / "
      "SearchByBoW(KF, KF) / SearchForTriangulation) identical to the oracle in %.0f s" % (*n_ok, time.time() - t0))
