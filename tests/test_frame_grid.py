"""N1: Frame grid + GetFeaturesInArea + windowed best/second-best search.
CPU: hand-checkable cases on the oracle restatement of src/Frame.cc:230-245, 327-392.
GPU: HIP (through the C ABI) == oracle exactly, candidate order included."""
import numpy as np
import pytest

import oracle_lib as O


def _kps(pts):
    k = np.zeros(len(pts), O.KP_DTYPE)
    for i, (x, y, o) in enumerate(pts):
        k[i] = (x, y, 31.0, 0.0, 50.0, o, -1)
    return k


def test_oracle_grid_by_hand():
    # 640x480 image: cells are 10 x 10 px; PosInGrid uses round(), GetFeaturesInArea floor/ceil
    k = _kps([(5, 5, 0), (14.9, 5, 0), (15.1, 5, 1), (100, 100, 2), (639, 479, 0), (634, 474, 0)])
    g = O.FrameGrid(k, 0.0, 640.0, 0.0, 480.0)
    cs = np.ctypeslib.as_array(g.g.cell_start)
    cell = lambda i: int(np.searchsorted(cs, np.nonzero(g.items[:cs[-1]] == i)[0][0], side="right") - 1)
    # PosInGrid rounds: 5 * 0.1 = 0.5 -> cell 1 (half away from zero), not floor
    assert cell(0) == 1 * 48 + 1
    assert cell(1) == 1 * 48 + 1 and cell(2) == 2 * 48 + 1
    assert cs[-1] == 5                       # (639,479) -> posX = round(63.9) = 64 -> outside the grid: dropped (:388)
    assert list(g.features_in_area(10.0, 5.0, 6.0)) == [0, 1, 2]          # |dx| < r strict
    assert list(g.features_in_area(10.0, 5.0, 5.0)) == [1]                # 5 - 10 = -5: not < 5
    assert list(g.features_in_area(10.0, 5.0, 6.0, 1, -1)) == [2]         # minLevel only
    assert list(g.features_in_area(10.0, 5.0, 6.0, 0, 0)) == [0, 1]       # level window
    assert list(g.features_in_area(-50.0, 5.0, 10.0)) == []               # window misses the grid
    assert list(g.features_in_area(1000.0, 5.0, 10.0)) == []


def test_oracle_area_order_is_column_major_then_index():
    rng = np.random.default_rng(0)
    pts = [(float(x), float(y), int(o)) for x, y, o in zip(rng.uniform(0, 640, 600), rng.uniform(0, 480, 600), rng.integers(0, 8, 600))]
    k = _kps(pts)
    g = O.FrameGrid(k, 0.0, 640.0, 0.0, 480.0)
    idx = g.features_in_area(320.0, 240.0, 90.0)
    assert len(idx) > 20
    px = np.rint((k["x"][idx] - 0) * np.float32(64 / 640)).astype(int)
    py = np.rint((k["y"][idx] - 0) * np.float32(48 / 480)).astype(int)
    key = list(zip(px, py, idx))
    assert key == sorted(key)
    brute = [i for i in range(600) if abs(k["x"][i] - 320) < 90 and abs(k["y"][i] - 240) < 90
             and 0 <= round(float(k["x"][i]) * 0.1) < 64 and 0 <= round(float(k["y"][i]) * 0.1) < 48]
    assert sorted(idx.tolist()) == sorted(brute)


@pytest.mark.gpu
def test_hip_grid_and_area_equal_oracle(orbx, synth):
    img0, img1 = synth.frame_pair(2, 960, 540)
    ex = orbx.ORBextractor(2000, max_width=960, max_height=540)
    k0, d0 = ex(img0)
    k1, d1 = ex(img1)
    m = orbx.ORBmatcher(0.9, True, max_queries=4096, max_train=4096, max_pairs=1 << 21)
    m.grid_build(k1, 0.0, 960.0, 0.0, 540.0)
    og = O.FrameGrid(k1, 0.0, 960.0, 0.0, 540.0)
    rng = np.random.default_rng(1)
    # windows around the previous frame's keypoints (the SearchForInitialization / SearchByProjection shape)
    x = k0["x"] + rng.uniform(-8, 8, len(k0)).astype(np.float32)
    y = k0["y"] + rng.uniform(-8, 8, len(k0)).astype(np.float32)
    r = (15.0 * np.float32(1.2) ** k0["octave"]).astype(np.float32)
    mn = np.maximum(k0["octave"] - 1, -1).astype(np.int32)
    mx = (k0["octave"] + 1).astype(np.int32)
    mn[::7] = -1; mx[::7] = -1                       # no level check
    x[::50] = -500.0                                  # windows that miss the image
    off, idx = m.GetFeaturesInArea(x, y, r, mn, mx)
    for i in range(len(x)):
        ref = og.features_in_area(float(x[i]), float(y[i]), float(r[i]), int(mn[i]), int(mx[i]))
        assert np.array_equal(idx[off[i]:off[i + 1]], ref), i
    assert off[-1] > 3000
    # fused search == oracle, with and without a skip mask
    skip = (rng.uniform(size=len(k1)) < 0.2).astype(np.uint8)
    for sk in (None, skip):
        bi, bd, sd = m.search_area_best2(d0, x, y, r, mn, mx, d1, sk)
        obi, obd, osd = og.search_area_best2(d0, x, y, r, mn, mx, d1, sk)
        assert np.array_equal(bd, obd) and np.array_equal(sd, osd) and np.array_equal(bi, obi)
    # and it equals the two-step path GetFeaturesInArea -> orbm_best2 on the CSR lists
    bi2, bd2, sd2 = m.best2(d0, d1, off, idx)
    bi, bd, sd = m.search_area_best2(d0, x, y, r, mn, mx, d1, None)
    assert np.array_equal(bi, bi2) and np.array_equal(bd, bd2) and np.array_equal(sd, sd2)


@pytest.mark.gpu
def test_hip_grid_edge_cases(orbx):
    m = orbx.ORBmatcher(max_queries=64, max_train=64, max_pairs=4096)
    with pytest.raises(orbx.OrbxError):
        m.GetFeaturesInArea([1.0], [1.0], 5.0)                 # no grid yet
    k = _kps([(5, 5, 0), (14.9, 5, 0), (15.1, 5, 1), (100, 100, 2), (639, 479, 0), (634, 474, 0)])
    m.grid_build(k, 0.0, 640.0, 0.0, 480.0)
    off, idx = m.GetFeaturesInArea([10.0, 10.0, -50.0], [5.0, 5.0, 5.0], [6.0, 5.0, 10.0])
    assert list(off) == [0, 3, 4, 4] and list(idx) == [0, 1, 2, 1]
    m.grid_build(k[:0], 0.0, 640.0, 0.0, 480.0)                # empty frame
    off, idx = m.GetFeaturesInArea([10.0], [5.0], 6.0)
    assert list(off) == [0, 0]
    bi, bd, sd = m.search_area_best2(np.zeros((1, 32), np.uint8), [10.0], [5.0], 6.0, -1, -1, np.zeros((0, 32), np.uint8))
    assert (bi[0], bd[0], sd[0]) == (-1, 256, 256)


@pytest.mark.gpu
@pytest.mark.parametrize("shift,window,ratio,ori", [((7, 3), 100, 0.9, True), ((25, 10), 30, 0.9, True), ((3, 2), 100, 0.6, False),
                                                      ((40, 0), 50, 1.0, True)])
def test_search_for_initialization_equals_oracle(orbx, synth, shift, window, ratio, ori):
    """ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:405-520) as Tracking::MonocularInitialization calls it
    (Tracking.cc:608-609: ORBmatcher(0.9, true), vbPrevMatched = the initial frame's keypoints, window 100), and a second
    call that starts from the updated vbPrevMatched like the next frame does."""
    W, H = 640, 480
    f0, f1 = synth.frame_pair(11, W, H, shift=shift)
    ex = orbx.ORBextractor(2000, max_width=W, max_height=H)          # mpIniORBextractor: 2 * nFeatures
    k0, d0 = ex(f0); k1, d1 = ex(f1)
    m = orbx.ORBmatcher(ratio, ori, max_queries=4096, max_train=4096, max_pairs=1 << 21)
    m.grid_build(k1, 0.0, float(W), 0.0, float(H))
    og = O.FrameGrid(k1, 0.0, float(W), 0.0, float(H))
    prev = np.ascontiguousarray(np.stack([k0["x"], k0["y"]], 1), np.float32)
    oprev = prev.copy()
    m12, nm = m.SearchForInitialization(k0, d0, k1, d1, prev, window)
    om12, onm = O.search_for_initialization(k0, d0, og, d1, oprev, window, ratio, ori)
    assert nm == onm and np.array_equal(m12, om12) and np.array_equal(prev, oprev)
    assert nm == int((m12 >= 0).sum()) and (k0["octave"][m12 >= 0] == 0).all()
    if window >= 50 and ratio >= 0.9:
        assert nm > 50
    sel = m12 >= 0
    assert len(np.unique(m12[sel])) == sel.sum()                         # vnMatches21 keeps the matching one-to-one
    # second round from the updated vbPrevMatched
    m12b, nmb = m.SearchForInitialization(k0, d0, k1, d1, prev, window)
    om12b, onmb = O.search_for_initialization(k0, d0, og, d1, oprev, window, ratio, ori)
    assert nmb == onmb and np.array_equal(m12b, om12b) and np.array_equal(prev, oprev)
    # degenerate: no keypoints in frame 1
    e12, en = m.SearchForInitialization(k0[:0], d0[:0], k1, d1, prev[:0].copy(), window)
    assert en == 0 and len(e12) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("mono,th,with_right,dz", [(True, 15.0, False, 0.0), (True, 7.0, False, 0.0), (False, 15.0, True, 0.0),
                                                    (False, 15.0, True, 0.9), (False, 15.0, True, -0.9)])
def test_search_by_projection_last_equals_oracle(orbx, synth, mono, th, with_right, dz):
    """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (src/ORBmatcher.cc:1328-1470) as TrackWithMotionModel
    calls it (Tracking.cc:879-883: ORBmatcher(0.9, true), th 15 or 7), on the three-depth scene: the last frame's MapPoints are
    its keypoints back-projected to their layer's depth, the current pose is the true one (a camera step along x, for the
    stereo cases also along z to exercise the forward / backward octave windows and the mvuRight gate)."""
    W, H = 1241, 376
    fx, fy, cx, cy, base, shifts = 718.856, 718.856, 607.1928, 185.2157, 0.5, (2, 4, 6)
    frames, layer = synth.stream_layers(5, W, H, 2, shifts=shifts)
    ex = orbx.ORBextractor(2000, max_width=W, max_height=H)
    k0, d0 = ex(frames[0]); k1, d1 = ex(frames[1])
    sf = ex.GetScaleFactors()
    Z = (fx * base / np.array(shifts, np.float64))[layer[np.clip(np.rint(k0["y"]).astype(int), 0, H - 1), np.clip(np.rint(k0["x"]).astype(int), 0, W - 1)]]
    xw = np.stack([(k0["x"] - cx) * Z / fx, (k0["y"] - cy) * Z / fy, Z], 1).astype(np.float32)
    rng = np.random.default_rng(3)
    has = (rng.random(len(k0)) < 0.85).astype(np.uint8)                 # some features without a MapPoint / flagged outlier
    obs = rng.integers(0, 4, len(k0)).astype(np.int32)                  # Observations(): 0 = a temporal point that may be overwritten
    Tlw = np.eye(4, dtype=np.float32)
    Tcw = np.eye(4, dtype=np.float32); Tcw[0, 3] = -base; Tcw[2, 3] = -dz
    mb, mbf = 0.54, 386.1448
    ur = None
    if with_right:
        Zc = (fx * base / np.array(shifts, np.float64))[layer[np.clip(np.rint(k1["y"]).astype(int), 0, H - 1), np.clip(np.rint(k1["x"]).astype(int), 0, W - 1)]]
        ur = (k1["x"] - mbf / Zc).astype(np.float32)
        ur[::5] = -1.0                                                  # monocular points
        ur[1::7] += 40.0                                                # stereo matches that contradict the projection: gated out
    m = orbx.ORBmatcher(0.9, True, max_queries=4096, max_train=4096, max_pairs=1 << 21)
    m.grid_build(k1, 0.0, float(W), 0.0, float(H))
    og = O.FrameGrid(k1, 0.0, float(W), 0.0, float(H))
    cur0 = np.full(len(k1), -1, np.int32)
    cur0[::11] = rng.integers(0, 3, len(cur0[::11]))                    # a few entries already hold a point (0 = replaceable)
    ca, cb = cur0.copy(), cur0.copy()
    cm, nm = m.SearchByProjectionLast(has, xw, d0, obs, k0, Tcw, Tlw, (fx, fy, cx, cy), mb, mbf, (0.0, W, 0.0, H), sf, k1, d1, ca, th, mono, ur)
    ocm, onm = O.search_by_projection_last(has, xw, d0, obs, k0, Tcw, Tlw, (fx, fy, cx, cy), mb, mbf, (0.0, W, 0.0, H), sf, og, d1, cb, th, mono, True, ur)
    assert nm == onm and np.array_equal(cm, ocm) and np.array_equal(ca, cb)
    assert (has[cm[cm >= 0]] == 1).all()
    if dz == 0.0:
        assert nm > 300                                                 # the true pose puts most points on their match
        good = cm >= 0
        assert np.median(np.abs(k1["x"][good] - (k0["x"][cm[good]] - np.array(shifts)[layer[np.clip(np.rint(k0["y"][cm[good]]).astype(int), 0, H - 1),
                                                                                                np.clip(np.rint(k0["x"][cm[good]]).astype(int), 0, W - 1)]]))) < 1.5


@pytest.mark.gpu
@pytest.mark.parametrize("th,orb_dist,ori,dz", [(10.0, 100, True, 0.0), (3.0, 64, True, 0.0), (10.0, 100, False, 0.0), (10.0, 100, True, 6.0), (3.0, 64, True, -4.0)])
def test_search_by_projection_kf_equals_oracle(orbx, synth, th, orb_dist, ori, dz):
    """ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (src/ORBmatcher.cc:1472-1599) with the call
    pairs of Tracking::Relocalization (Tracking.cc:1459: th 10, ORBdist 100; :1473: th 3, ORBdist 64; ORBmatcher(0.9, true)) on the
    three-depth scene: the key frame's MapPoints are its keypoints back-projected to their layer's depth, some of them bad, NULL or
    already found; the current frame already holds some points; a pose step along z moves points across pyramid levels and out of
    their distance-invariance range.  projection -> PredictScale -> search through the C ABI == the oracle's single restatement."""
    W, H = 1241, 376
    fx, fy, cx, cy, base, shifts = 718.856, 718.856, 607.1928, 185.2157, 0.5, (2, 4, 6)
    frames, layer = synth.stream_layers(5, W, H, 2, shifts=shifts)
    ex = orbx.ORBextractor(2000, max_width=W, max_height=H)
    k0, d0 = ex(frames[0]); k1, d1 = ex(frames[1])
    sf = ex.GetScaleFactors()
    logsf = float(np.log(np.float32(1.2)))
    Z = (fx * base / np.array(shifts, np.float64))[layer[np.clip(np.rint(k0["y"]).astype(int), 0, H - 1), np.clip(np.rint(k0["x"]).astype(int), 0, W - 1)]]
    xw = np.stack([(k0["x"] - cx) * Z / fx, (k0["y"] - cy) * Z / fy, Z], 1).astype(np.float32)
    rng = np.random.default_rng(int(th) + orb_dist)
    usable = (rng.random(len(k0)) < 0.8).astype(np.uint8)                # pMP && !isBad() && !sAlreadyFound.count(pMP)
    # MapPoint::UpdateNormalAndDepth: mfMaxDistance = dist * scaleFactor^level, mfMinDistance = mfMaxDistance / scaleFactor^(nlevels-1)
    d_ref = np.linalg.norm(xw.astype(np.float64), axis=1)
    mf_max = (d_ref * sf[k0["octave"]]).astype(np.float32)
    mf_min = (mf_max / sf[7]).astype(np.float32)
    min_inv, max_inv = (np.float32(0.8) * mf_min).astype(np.float32), (np.float32(1.2) * mf_max).astype(np.float32)
    Tcw = np.eye(4, dtype=np.float32); Tcw[0, 3] = -base; Tcw[2, 3] = -dz
    K, bounds = (fx, fy, cx, cy), (0.0, float(W), 0.0, float(H))
    m = orbx.ORBmatcher(0.9, ori, max_queries=4096, max_train=4096, max_pairs=1 << 21)
    m.grid_build(k1, *bounds)
    og = O.FrameGrid(k1, *bounds)
    has0 = (rng.random(len(k1)) < 0.15).astype(np.uint8)                 # slots of the current frame that already hold a MapPoint
    ha, hb = has0.copy(), has0.copy()
    u, v, iz, d3, inside = m.ProjectPoints(Tcw, K, bounds, xw)
    lv = m.PredictScale(mf_max, d3, logsf, 8)
    use = (usable.astype(bool) & inside.astype(bool) & ~(d3 < min_inv) & ~(d3 > max_inv)).astype(np.uint8)
    cm, nm = m.SearchByProjectionKF(use, u, v, lv, d0, k0["angle"], sf, k1, d1, ha, th, orb_dist)
    ocm, onm = O.search_by_projection_kf(usable, xw, min_inv, max_inv, mf_max, d0, k0["angle"], Tcw, K, bounds, sf, logsf, og, d1, hb, th, orb_dist, ori)
    assert nm == onm and np.array_equal(cm, ocm) and np.array_equal(ha, hb)
    assert nm == int((cm >= 0).sum()) and (has0[cm >= 0] == 0).all() and (usable[cm[cm >= 0]] == 1).all()
    assert np.array_equal(ha, has0 | (cm >= 0))
    if dz == 0.0:
        assert nm > (250 if th >= 10 else 150)                           # the true pose puts most free points on their match
    elif dz > 0:
        assert (use == 0).sum() > (usable == 0).sum()                    # the step towards the scene pushed some points out of view / range
    # degenerate inputs
    e_cm, e_nm = m.SearchByProjectionKF(use[:0], u[:0], v[:0], lv[:0], d0[:0], k0["angle"][:0], sf, k1, d1, ha.copy(), th, orb_dist)
    assert e_nm == 0 and (e_cm == -1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("th,ratio,with_right", [(1.0, 0.8, False), (3.0, 0.8, False), (5.0, 0.6, True), (1.0, 1.0, True)])
def test_search_by_projection_map_equals_oracle(orbx, synth, th, ratio, with_right):
    """ORBmatcher::SearchByProjection(F, vpMapPoints, th) (src/ORBmatcher.cc:45-125) as Tracking::SearchLocalPoints calls it
    (Tracking.cc:1191-1199: ORBmatcher(0.8), th 1, 3 or 5): the local map = the previous frame's keypoints predicted at their
    shifted positions and octaves (+-1 for some), viewing cosines on both sides of 0.998."""
    W, H = 1241, 376
    f0, f1 = synth.frame_pair(21, W, H, shift=(6, 2))
    ex = orbx.ORBextractor(2000, max_width=W, max_height=H)
    k0, d0 = ex(f0); k1, d1 = ex(f1)
    sf = ex.GetScaleFactors()
    rng = np.random.default_rng(int(th * 10) + int(with_right))
    n = len(k0)
    in_view = (rng.random(n) < 0.8).astype(np.uint8)
    px = (k0["x"] - 6 + rng.normal(0, 0.7, n)).astype(np.float32); py = (k0["y"] - 2 + rng.normal(0, 0.7, n)).astype(np.float32)
    lv = np.clip(k0["octave"] + rng.integers(0, 2, n), 0, 7).astype(np.int32)       # GetFeaturesInArea(level - 1, level)
    vc = np.where(rng.random(n) < 0.5, 0.9985, 0.99).astype(np.float32)
    obs = rng.integers(1, 5, n).astype(np.int32)
    ur = pxr = None
    if with_right:
        ur = (k1["x"] - rng.uniform(2, 30, len(k1))).astype(np.float32); ur[::4] = -1.0
        pxr = (px - rng.uniform(2, 30, n)).astype(np.float32)
    m = orbx.ORBmatcher(ratio, True, max_queries=4096, max_train=4096, max_pairs=1 << 21)
    m.grid_build(k1, 0.0, float(W), 0.0, float(H))
    og = O.FrameGrid(k1, 0.0, float(W), 0.0, float(H))
    cur0 = np.full(len(k1), -1, np.int32); cur0[::9] = rng.integers(0, 3, len(cur0[::9]))
    ca, cb = cur0.copy(), cur0.copy()
    cm, nm = m.SearchByProjectionMap(in_view, px, py, lv, vc, d0, obs, sf, k1, d1, ca, th, pxr, ur)
    ocm, onm = O.search_by_projection_map(in_view, px, py, lv, vc, d0, obs, sf, og, d1, cb, th, ratio, pxr, ur)
    assert nm == onm and np.array_equal(cm, ocm) and np.array_equal(ca, cb)
    assert nm == int((cm >= 0).sum()) or nm >= int((cm >= 0).sum())        # nmatches counts assignments (overwrites included)
    if not with_right and th >= 3.0:
        assert nm > 300
