#!/usr/bin/env python3
"""Randomised GPU-vs-oracle sweep of the three window matchers built on the Frame grid: SearchForInitialization,
SearchByProjection(CurrentFrame, LastFrame), SearchByProjection(F, vpMapPoints), SearchByProjection(CurrentFrame, pKF, ...) and the
fused window search.  usage: stress_proj.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import conftest  # noqa
import oracle_lib as O
import my_slam_amd as M
import my_slam_amd.synth as synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); n_ok = [0, 0, 0, 0, 0]
while time.time() - t0 < budget:
    W = int(rng.integers(240, 1300)); H = int(rng.integers(180, 720)); nf = int(rng.choice([300, 1000, 2000, 4000]))
    shift = (int(rng.integers(0, 30)), int(rng.integers(0, 12)))
    f0, f1 = synth.frame_pair(int(rng.integers(1, 1 << 30)), W, H, shift=shift)
    try:
        ex = M.ORBextractor(nf, max_width=W, max_height=H)
        k0, d0 = ex(f0); k1, d1 = ex(f1)
    except M.OrbxError:
        continue
    if len(k0) < 8 or len(k1) < 8:
        continue
    sf = ex.GetScaleFactors()
    ratio = float(rng.choice([0.6, 0.8, 0.9, 1.0])); ori = bool(rng.integers(0, 2))
    m = M.ORBmatcher(ratio, ori, max_queries=8192, max_train=8192, max_pairs=1 << 22)
    m.grid_build(k1, 0.0, float(W), 0.0, float(H))
    og = O.FrameGrid(k1, 0.0, float(W), 0.0, float(H))
    # 1. SearchForInitialization
    win = int(rng.choice([10, 30, 100]))
    pa = np.ascontiguousarray(np.stack([k0["x"], k0["y"]], 1), np.float32); pb = pa.copy()
    a, na = m.SearchForInitialization(k0, d0, k1, d1, pa, win)
    b, nb = O.search_for_initialization(k0, d0, og, d1, pb, win, ratio, ori)
    assert na == nb and np.array_equal(a, b) and np.array_equal(pa, pb), ("init", W, H, nf, shift, win, ratio, ori)
    n_ok[0] += 1
    # 2. SearchByProjection(CurrentFrame, LastFrame): points on a plane at depth Z seen by a camera stepping (tx, 0, tz)
    fx = fy = float(rng.uniform(300, 900)); cx, cy = W / 2.0, H / 2.0
    Z = float(rng.uniform(5, 50)); tz = float(rng.choice([0.0, 0.0, 1.0, -1.0])); mono = bool(rng.integers(0, 2))
    xw = np.stack([(k0["x"] - cx) * Z / fx, (k0["y"] - cy) * Z / fy, np.full(len(k0), Z)], 1).astype(np.float32)
    Tcw = np.eye(4, dtype=np.float32); Tcw[0, 3] = -shift[0] * Z / fx; Tcw[1, 3] = -shift[1] * Z / fy; Tcw[2, 3] = -tz
    has = (rng.random(len(k0)) < 0.9).astype(np.uint8); obs = rng.integers(0, 3, len(k0)).astype(np.int32)
    ur = None
    if not mono:
        ur = (k1["x"] - 40.0 / Z * fx / 100 - rng.uniform(0, 3, len(k1))).astype(np.float32); ur[::3] = -1
    th = float(rng.choice([7.0, 15.0]))
    c0 = np.full(len(k1), -1, np.int32); c0[::7] = rng.integers(0, 3, len(c0[::7])); ca, cb = c0.copy(), c0.copy()
    a, na = m.SearchByProjectionLast(has, xw, d0, obs, k0, Tcw, np.eye(4, dtype=np.float32), (fx, fy, cx, cy), 0.5, 40.0 * fx / 100, (0.0, W, 0.0, H), sf, k1, d1, ca, th, mono, ur)
    b, nb = O.search_by_projection_last(has, xw, d0, obs, k0, Tcw, np.eye(4, dtype=np.float32), (fx, fy, cx, cy), 0.5, 40.0 * fx / 100, (0.0, W, 0.0, H), sf, og, d1, cb, th, mono, ori, ur)
    assert na == nb and np.array_equal(a, b) and np.array_equal(ca, cb), ("last", W, H, nf, shift, th, mono, tz)
    n_ok[1] += 1
    # 3. SearchByProjection(F, vpMapPoints, th)
    n = len(k0)
    px = (k0["x"] - shift[0] + rng.normal(0, 1, n)).astype(np.float32); py = (k0["y"] - shift[1] + rng.normal(0, 1, n)).astype(np.float32)
    lv = np.clip(k0["octave"] + rng.integers(-1, 2, n), 0, len(sf) - 1).astype(np.int32)
    vc = rng.choice([0.9985, 0.998, 0.99], n).astype(np.float32)
    inv = (rng.random(n) < 0.8).astype(np.uint8); obs = rng.integers(1, 4, n).astype(np.int32)
    pxr = (px - rng.uniform(0, 20, n)).astype(np.float32) if ur is not None else None
    thm = float(rng.choice([1.0, 3.0, 5.0]))
    ca, cb = c0.copy(), c0.copy()
    a, na = m.SearchByProjectionMap(inv, px, py, lv, vc, d0, obs, sf, k1, d1, ca, thm, pxr, ur)
    b, nb = O.search_by_projection_map(inv, px, py, lv, vc, d0, obs, sf, og, d1, cb, thm, ratio, pxr, ur)
    assert na == nb and np.array_equal(a, b) and np.array_equal(ca, cb), ("map", W, H, nf, shift, thm, ratio)
    n_ok[2] += 1
    # 4. the fused window search (orbm_search_area_best2): random windows, octave ranges and skip masks
    nq = int(rng.integers(1, len(k0) + 1))
    qx = rng.uniform(-20, W + 20, nq).astype(np.float32); qy = rng.uniform(-20, H + 20, nq).astype(np.float32)
    qr = rng.uniform(1, 80, nq).astype(np.float32)
    qmn = rng.integers(-1, 4, nq).astype(np.int32); qmx = np.where(rng.random(nq) < 0.3, -1, qmn + rng.integers(0, 4, nq)).astype(np.int32)
    skip = (rng.random(len(k1)) < 0.2).astype(np.uint8) if rng.integers(0, 2) else None
    a = m.search_area_best2(d0[:nq], qx, qy, qr, qmn, qmx, d1, skip)
    b = og.search_area_best2(d0[:nq], qx, qy, qr, qmn, qmx, d1, skip)
    assert all(np.array_equal(u, v) for u, v in zip(a, b)), ("area", W, H, nf, nq)
    n_ok[3] += 1
    # 5. SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist): the same plane seen from a pose that is off along z
    usable = (rng.random(n) < 0.8).astype(np.uint8)
    d_ref = np.linalg.norm(xw.astype(np.float64), axis=1)
    mf_max = (d_ref * sf[k0["octave"]]).astype(np.float32); mf_min = (mf_max / sf[-1]).astype(np.float32)
    mn_inv, mx_inv = (np.float32(0.8) * mf_min).astype(np.float32), (np.float32(1.2) * mf_max).astype(np.float32)
    Tk = Tcw.copy(); Tk[2, 3] = -float(rng.choice([0.0, 0.3 * Z, -0.2 * Z, 2.0 * Z]))
    logsf = float(np.log(np.float32(sf[1])))
    thk, od = (10.0, 100) if rng.integers(0, 2) else (3.0, 64)
    h0 = (rng.random(len(k1)) < 0.2).astype(np.uint8); ha, hb = h0.copy(), h0.copy()
    K4 = (fx, fy, cx, cy); bnd = (0.0, float(W), 0.0, float(H))
    u, v, _, d3, ins = m.ProjectPoints(Tk, K4, bnd, xw)
    lvk = m.PredictScale(mf_max, d3, logsf, len(sf))
    use = (usable.astype(bool) & ins.astype(bool) & ~(d3 < mn_inv) & ~(d3 > mx_inv)).astype(np.uint8)
    a, na = m.SearchByProjectionKF(use, u, v, lvk, d0, k0["angle"], sf, k1, d1, ha, thk, od)
    b, nb = O.search_by_projection_kf(usable, xw, mn_inv, mx_inv, mf_max, d0, k0["angle"], Tk, K4, bnd, sf, logsf, og, d1, hb, thk, od, ori)
    assert na == nb and np.array_equal(a, b) and np.array_equal(ha, hb), ("kf", W, H, nf, shift, thk, od, float(Tk[2, 3]))
    n_ok[4] += 1
print("stress_proj: %d / %d / %d / %d / %d random cases (SearchForInitialization / SearchByProjection last frame / local map / fused window search / "
      "SearchByProjection key frame) identical to the oracle in %.0f s" % (n_ok[0], n_ok[1], n_ok[2], n_ok[3], n_ok[4], time.time() - t0))
