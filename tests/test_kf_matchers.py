"""The six ORBmatcher methods of the LocalMapping / LoopClosing threads:
SearchByProjection(KeyFrame*, Scw, ...) src/ORBmatcher.cc:290-403, SearchByBoW(KeyFrame*, KeyFrame*, ...) :522-655,
SearchForTriangulation :657-823, Fuse(KeyFrame*, vpMapPoints, th) :825-975, Fuse(KeyFrame*, Scw, ...) :977-1100,
SearchBySim3 :1102-1326.
CPU: hand-checkable cases on the oracle restatement (oracle/orb_oracle_kf.c).
GPU: the C-ABI pipeline (host projection helper -> PredictScale -> search on the GPU) == the oracle's single restatement of each
method, bit for bit, on the three-depth scene: two key frames a baseline apart, their MapPoints = their keypoints back-projected
to the depth of their layer.  Parity is unpinned to the reference (it holds no fixtures)."""
import numpy as np
import pytest

import oracle_lib as O

W, H = 1241, 376
FX, FY, CX, CY, BASE, SHIFTS = 718.856, 718.856, 607.1928, 185.2157, 0.5, (2, 4, 6)
K = (FX, FY, CX, CY)
F32 = np.float32


def _kf_grid(distorted):
    """(assign_min_x, assign_min_y, inv_w, inv_h, query_min_x, query_min_y), int bounds, as KeyFrame holds them.  `distorted`:
    Frame's undistorted bounds are fractional, KeyFrame truncates its copies to int (include/KeyFrame.h:190-193)."""
    if not distorted:
        return (0.0, 0.0, float(F32(64) / F32(W)), float(F32(48) / F32(H)), 0.0, 0.0), (0.0, float(W), 0.0, float(H))
    mnx, mxx, mny, mxy = F32(-3.7), F32(W + 4.2), F32(-2.6), F32(H + 1.9)
    grid = (float(mnx), float(mny), float(F32(64) / (mxx - mnx)), float(F32(48) / (mxy - mny)), float(int(mnx)), float(int(mny)))
    return grid, (float(int(mnx)), float(int(mxx)), float(int(mny)), float(int(mxy)))


class Scene:
    """Two key frames of the three-depth scene.  KF1 = frame 0 at the origin, KF2 = frame 1 after a step of BASE along +x."""

    def __init__(self, orbx, synth, seed=5, nfeatures=2000):
        frames, layer = synth.stream_layers(seed, W, H, 2, shifts=SHIFTS)
        ex = orbx.ORBextractor(nfeatures, max_width=W, max_height=H)
        self.k = [None, None]; self.d = [None, None]
        self.k[0], self.d[0] = ex(frames[0]); self.k[1], self.d[1] = ex(frames[1])
        self.sf = ex.GetScaleFactors(); self.sigma2 = ex.GetScaleSigmaSquares(); self.inv_sigma2 = ex.GetInverseScaleSigmaSquares()
        self.logsf = float(np.log(F32(1.2)))
        self.T = [np.eye(4, dtype=F32), np.eye(4, dtype=F32)]
        self.T[1][0, 3] = -BASE
        self.xw, self.normal, self.mf_max, self.min_inv, self.max_inv = [], [], [], [], []
        for i in (0, 1):
            k = self.k[i]
            Z = (FX * BASE / np.array(SHIFTS, np.float64))[layer[np.clip(np.rint(k["y"]).astype(int), 0, H - 1), np.clip(np.rint(k["x"]).astype(int), 0, W - 1)]]
            xc = np.stack([(k["x"] - CX) * Z / FX, (k["y"] - CY) * Z / FY, Z], 1)
            xw = (xc - self.T[i][:3, 3].astype(np.float64)).astype(F32)              # R = I
            d_ref = np.linalg.norm(xc, axis=1)
            self.xw.append(xw)
            self.normal.append((xc / d_ref[:, None]).astype(F32))                      # MapPoint::UpdateNormalAndDepth: mean viewing direction
            mf_max = (d_ref * self.sf[k["octave"]]).astype(F32)
            self.mf_max.append(mf_max)
            self.min_inv.append((F32(0.8) * (mf_max / self.sf[7]).astype(F32)).astype(F32))
            self.max_inv.append((F32(1.2) * mf_max).astype(F32))

    def featvecs(self, orbx, tmp_path, levelsup=2, k=10, depth=3, seed=9):
        import test_vocabulary as TV
        path = str(tmp_path / "voc.txt")
        TV.make_vocabulary(path, k, depth, seed=seed)
        v = orbx.ORBVocabulary(path)
        return [v.transform(self.d[i], levelsup)[1] for i in (0, 1)]


def _sim3(s, rot_deg=0.0, t=(-BASE, 0.0, 0.0)):
    a = np.deg2rad(rot_deg)
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    S = np.eye(4)
    S[:3, :3] = s * R; S[:3, 3] = s * np.array(t)
    return S.astype(F32)


# ---------------------------------------------------------------------------------------------------------------------------------
# CPU: the oracle by hand
# ---------------------------------------------------------------------------------------------------------------------------------
def test_oracle_small_matrix_algebra():
    """cv::gemm's small-matrix path: three float products summed in float, left to right; -R^T t likewise."""
    T = np.eye(4, dtype=F32)
    T[0, :3] = [F32(1e8), F32(1.0), F32(-1e8)]; T[0, 3] = F32(0.25)
    x = np.array([1, 1, 1], F32)
    L = O.lib()
    got = L.oro_gemm_row(O._p(T.reshape(16)), 0, O._p(x))
    assert got == 0.25                       # (1e8 + 1) rounds to 1e8 in float, then - 1e8 = 0; a double accumulation would give 1.25
    S = _sim3(2.0)
    Tc, Ow = O.sim3_decompose(S)
    assert np.allclose(Tc[:3, :3], np.eye(3)) and np.allclose(Tc[:3, 3], [-BASE, 0, 0]) and np.allclose(Ow, [BASE, 0, 0])
    assert np.array_equal(O.camera_center(Tc), Ow)


def test_oracle_kf_grid_origins():
    """A key frame's cells come from Frame's float origin, its queries subtract the key frame's int origin."""
    k = np.zeros(3, O.KP_DTYPE)
    k["x"] = [0.4, 10.0, 20.0]; k["y"] = [5.0, 5.0, 5.0]
    inv = 0.1
    g = O.KeyFrameGrid(k, (-3.7, 0.0, inv, inv, -3.0, 0.0))
    cs = np.ctypeslib.as_array(g.g.cell_start)
    assert cs[-1] == 3
    # assignment: round((0.4 + 3.7) * 0.1) = 0, round(1.37) = 1, round(2.37) = 2
    first = [int(np.searchsorted(cs, np.nonzero(g.items[:3] == i)[0][0], side="right") - 1) // 48 for i in range(3)]
    assert first == [0, 1, 2]
    # query at x = 16, r = 5 with the int origin: floor((16 + 3 - 5) * 0.1) = 1 .. ceil((16 + 3 + 5) * 0.1) = 3 -> keypoints 1 and 2 by
    # cell, then |dx| < r keeps only keypoint 2 (|20 - 16| = 4)
    assert list(g.features_in_area(16.0, 5.0, 5.0)) == [2]
    assert list(g.features_in_area(16.0, 5.0, 7.0)) == [1, 2]


def test_oracle_triangulation_tie_takes_the_last():
    """`dist > bestDist` is the skip test (:738), so among equal distances the LAST candidate that passes the epipolar tests wins."""
    kp1 = np.zeros(1, O.KP_DTYPE); kp1["x"], kp1["y"] = 100.0, 50.0
    kp2 = np.zeros(3, O.KP_DTYPE); kp2["x"] = [90.0, 95.0, 300.0]; kp2["y"] = [50.0, 50.0, 50.0]
    d1 = np.zeros((1, 32), np.uint8)
    d2 = np.zeros((3, 32), np.uint8); d2[:, 0] = 0x0F              # all three at distance 4
    fv1 = (np.array([7], np.int32), np.array([0, 1], np.int32), np.array([0], np.int32))
    fv2 = (np.array([7], np.int32), np.array([0, 3], np.int32), np.array([0, 1, 2], np.int32))
    # F12 for a pure x translation: epipolar lines are the rows y2 = y1
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], F32)
    sf = np.array([1.0, 1.2], F32); s2 = sf * sf
    T2w = np.eye(4, dtype=F32); T2w[0, 3] = -1.0
    Cw = np.array([0, 0, -10], F32)                                   # epipole (679, 185): far from the candidates
    args = dict(Cw=Cw, T2w=T2w, K2=K, F12=F12, sf2=sf, sigma2_2=s2)
    no = np.zeros(1, np.uint8); no3 = np.zeros(3, np.uint8)
    mono1 = np.full(1, -1, F32); mono3 = np.full(3, -1, F32)
    m, n = O.search_for_triangulation(kp1, d1, no, mono1, fv1, kp2, d2, no3, mono3, fv2, only_stereo=False, check_ori=False, **args)
    assert n == 1 and m[0] == 2
    # a MapPoint on the last one: the second takes it
    has = np.array([0, 0, 1], np.uint8)
    m, n = O.search_for_triangulation(kp1, d1, no, mono1, fv1, kp2, d2, has, mono3, fv2, only_stereo=False, check_ori=False, **args)
    assert n == 1 and m[0] == 1
    # off the epipolar line by more than sqrt(3.84): rejected
    kp2b = kp2.copy(); kp2b["y"] = [50.0, 53.0, 55.0]
    m, n = O.search_for_triangulation(kp1, d1, no, mono1, fv1, kp2b, d2, no3, mono3, fv2, only_stereo=False, check_ori=False, **args)
    assert n == 1 and m[0] == 0
    # bOnlyStereo with monocular points: nothing
    m, n = O.search_for_triangulation(kp1, d1, no, mono1, fv1, kp2, d2, no3, mono3, fv2, only_stereo=True, check_ori=False, **args)
    assert n == 0 and m[0] == -1


def test_oracle_bow_kf_is_strict_below_th_low():
    """SearchByBoW(KF, KF) accepts `bestDist1 < TH_LOW` (:598), its key-frame / frame sibling `<= TH_LOW` (:228)."""
    d1 = np.zeros((1, 32), np.uint8)
    d2 = np.zeros((1, 32), np.uint8); d2[0, :6] = 0xFF; d2[0, 6] = 0x03      # distance 50
    fv = (np.array([3], np.int32), np.array([0, 1], np.int32), np.array([0], np.int32))
    one = np.ones(1, np.uint8); a = np.zeros(1, F32)
    m, n = O.search_by_bow_kf(d1, a, one, fv, d2, a, one, fv, 0.9, False)
    assert n == 0 and m[0] == -1
    d2[0, 6] = 0x01                                                          # distance 49
    m, n = O.search_by_bow_kf(d1, a, one, fv, d2, a, one, fv, 0.9, False)
    assert n == 1 and m[0] == 0
    m, n = O.search_by_bow_kf(d1, a, one, fv, d2, a, np.zeros(1, np.uint8), fv, 0.9, False)      # KF2's feature has no MapPoint
    assert n == 0


def test_host_algebra_of_the_library_equals_oracle(orbx):
    """orbm_sim3_decompose / orbm_sim3_relative / orbm_project_points_kf are host code: checked here without a GPU."""
    rng = np.random.default_rng(4)
    for _ in range(200):
        a, b, c = rng.uniform(-0.5, 0.5, 3)
        Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
        Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
        Rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
        S = np.eye(4); s = rng.uniform(0.5, 2.0)
        S[:3, :3] = s * (Rx @ Ry @ Rz); S[:3, 3] = s * rng.uniform(-5, 5, 3)
        S = S.astype(F32)
        T, Ow = orbx.ORBmatcher.Sim3Decompose(S)
        oT, oOw = O.sim3_decompose(S)
        assert np.array_equal(T.view(np.uint32), oT.view(np.uint32)) and np.array_equal(Ow.view(np.uint32), oOw.view(np.uint32))
        # the projection helper against the oracle's row products
        X = rng.uniform(-20, 20, (16, 3)).astype(F32)
        u, v, iz, d3, ok = orbx.ORBmatcher.ProjectPointsKF(T, K, (0.0, float(W), 0.0, float(H)), X, None, None)
        L = O.lib()
        for i in range(len(X)):
            pc = [L.oro_gemm_row(O._p(oT.reshape(16)), r, O._p(X[i])) for r in range(3)]
            izr = F32(1) / F32(pc[2])
            assert iz[i] == izr or (np.isnan(iz[i]) and np.isnan(izr))
            uu = F32(FX) * (F32(pc[0]) * izr) + F32(CX)
            assert u[i] == uu or (np.isnan(u[i]) and np.isnan(uu))
            PO = X[i] - oOw
            assert d3[i] == F32(np.sqrt(np.sum(PO.astype(np.float64) ** 2)))
    sR12, sR21, t21 = orbx.ORBmatcher.Sim3Relative(1.25, np.eye(3, dtype=F32), np.array([1, 2, 3], F32))
    assert np.array_equal(sR12, 1.25 * np.eye(3, dtype=F32)) and np.array_equal(sR21, F32(0.8) * np.eye(3, dtype=F32))
    assert np.array_equal(t21, -(F32(0.8) * np.array([1, 2, 3], F32)))


# ---------------------------------------------------------------------------------------------------------------------------------
# GPU: C ABI == oracle
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def scene(orbx, synth):
    return Scene(orbx, synth)


def _project_kf(orbx, sc, src, T, Ow, bounds, normal=True):
    """host half of a KeyFrame search through the C ABI: projection, distance gate, PredictScale"""
    u, v, iz, d3, ok = orbx.ORBmatcher.ProjectPointsKF(T, K, bounds, sc.xw[src], sc.normal[src] if normal is not False else None, Ow)
    use = ok.astype(bool) & ~(d3 < sc.min_inv[src]) & ~(d3 > sc.max_inv[src])
    lv = orbx.ORBmatcher.PredictScale(sc.mf_max[src], d3, sc.logsf, 8)
    return u, v, iz, d3, use, lv


@pytest.mark.gpu
@pytest.mark.parametrize("s,rot,th,distorted", [(1.0, 0.0, 10, False), (1.04, 0.4, 10, False), (0.97, -0.3, 4, True), (1.0, 0.0, 10, True)])
def test_search_by_projection_sim3_equals_oracle(orbx, scene, s, rot, th, distorted):
    """LoopClosing.cc:376: the loop MapPoints (here KF1's) projected with Scw into the current key frame (KF2); some slots of
    vpMatched are taken already, some points are bad / already found, scale and a small rotation move points across levels."""
    sc = scene
    grid, bounds = _kf_grid(distorted)
    Scw = _sim3(s, rot)
    rng = np.random.default_rng(int(s * 100) + th)
    n = len(sc.k[0])
    usable = (rng.random(n) < 0.85).astype(np.uint8)
    normal = sc.normal[0].copy()
    flip = rng.random(n) < 0.1
    normal[flip] = -normal[flip]                                             # viewed from behind: the 60-degree test drops them
    matched0 = (rng.random(len(sc.k[1])) < 0.2).astype(np.uint8)
    ma, mb = matched0.copy(), matched0.copy()
    m = orbx.ORBmatcher(0.75, True, max_queries=4096, max_train=4096, max_pairs=1 << 21)
    m.grid_build_kf(sc.k[1], grid)
    T, Ow = orbx.ORBmatcher.Sim3Decompose(Scw)
    u, v, iz, d3, ok = orbx.ORBmatcher.ProjectPointsKF(T, K, bounds, sc.xw[0], normal, Ow)
    use = usable.astype(bool) & ok.astype(bool) & ~(d3 < sc.min_inv[0]) & ~(d3 > sc.max_inv[0])
    lv = orbx.ORBmatcher.PredictScale(sc.mf_max[0], d3, sc.logsf, 8)
    km, nm = m.SearchByProjectionSim3(use.astype(np.uint8), u, v, lv, sc.d[0], sc.sf, sc.k[1], sc.d[1], ma, th)
    og = O.KeyFrameGrid(sc.k[1], grid)
    okm, onm = O.search_by_projection_sim3(usable, sc.xw[0], normal, sc.min_inv[0], sc.max_inv[0], sc.mf_max[0], sc.d[0], Scw, K, bounds, sc.sf,
                                           sc.logsf, og, sc.d[1], mb, th)
    assert nm == onm and np.array_equal(km, okm) and np.array_equal(ma, mb)
    assert nm == int((km >= 0).sum()) and (matched0[km >= 0] == 0).all() and (usable[km[km >= 0]] == 1).all() and not flip[km[km >= 0]].any()
    if s == 1.0 and th == 10:
        assert nm > 300


@pytest.mark.gpu
@pytest.mark.parametrize("th,distorted,dz", [(3.0, False, 0.0), (3.0, True, 0.0), (2.0, False, 3.0), (5.0, False, -2.0)])
def test_fuse_equals_oracle(orbx, scene, th, distorted, dz):
    """LocalMapping.cc:491, 516 (th = 3.0 default): KF1's MapPoints fused into KF2, stereo and monocular key-frame features, a pose
    step along z for the distance-invariance gate and the level prediction."""
    sc = scene
    grid, bounds = _kf_grid(distorted)
    rng = np.random.default_rng(int(th * 10) + int(distorted))
    n, n2 = len(sc.k[0]), len(sc.k[1])
    T = sc.T[1].copy(); T[2, 3] = -dz
    Ow = O.camera_center(T)
    bf = F32(FX * BASE)
    usable = (rng.random(n) < 0.85).astype(np.uint8)                         # pMP && !isBad() && !IsInKeyFrame(pKF)
    ur_kf = np.full(n2, -1, F32)
    st = rng.random(n2) < 0.6
    Zc = sc.xw[1][:, 2] + T[2, 3]
    ur_kf[st] = (sc.k[1]["x"] - bf / Zc)[st].astype(F32)
    ur_kf[st & (rng.random(n2) < 0.15)] += 6.0                               # stereo observations that contradict the projection: gated out
    m = orbx.ORBmatcher(0.6, True, max_queries=4096, max_train=4096, max_pairs=1 << 21)
    m.grid_build_kf(sc.k[1], grid)
    u, v, iz, d3, ok = orbx.ORBmatcher.ProjectPointsKF(T, K, bounds, sc.xw[0], sc.normal[0], Ow)
    use = usable.astype(bool) & ok.astype(bool) & ~(d3 < sc.min_inv[0]) & ~(d3 > sc.max_inv[0])
    lv = orbx.ORBmatcher.PredictScale(sc.mf_max[0], d3, sc.logsf, 8)
    ur = (u - bf * iz).astype(F32)                                           # :870
    bi, nf = m.Fuse(use.astype(np.uint8), u, v, ur, lv, sc.d[0], sc.sf, sc.inv_sigma2, sc.k[1], ur_kf, sc.d[1], th)
    og = O.KeyFrameGrid(sc.k[1], grid)
    obi, onf = O.fuse(usable, sc.xw[0], sc.normal[0], sc.min_inv[0], sc.max_inv[0], sc.mf_max[0], sc.d[0], T, Ow, K, float(bf), bounds, sc.sf,
                      sc.inv_sigma2, sc.logsf, og, ur_kf, sc.d[1], th)
    assert nf == onf and np.array_equal(bi, obi)
    assert nf == int((bi >= 0).sum()) and (usable[bi >= 0] == 1).all()
    if dz == 0.0 and th >= 3.0:
        assert nf > 250
    # degenerate inputs
    ebi, enf = m.Fuse(use[:0], u[:0], v[:0], ur[:0], lv[:0], sc.d[0][:0], sc.sf, sc.inv_sigma2, sc.k[1], ur_kf, sc.d[1], th)
    assert enf == 0 and len(ebi) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("s,rot,th,distorted", [(1.0, 0.0, 4.0, False), (1.03, 0.3, 4.0, True), (0.98, -0.2, 2.5, False)])
def test_fuse_sim3_equals_oracle(orbx, scene, s, rot, th, distorted):
    """LoopClosing.cc:600 (th = 4): the loop MapPoints projected with the corrected Scw into a key frame."""
    sc = scene
    grid, bounds = _kf_grid(distorted)
    Scw = _sim3(s, rot)
    rng = np.random.default_rng(int(s * 1000))
    n = len(sc.k[0])
    usable = (rng.random(n) < 0.9).astype(np.uint8)
    m = orbx.ORBmatcher(0.8, True, max_queries=4096, max_train=4096, max_pairs=1 << 21)
    m.grid_build_kf(sc.k[1], grid)
    T, Ow = orbx.ORBmatcher.Sim3Decompose(Scw)
    u, v, iz, d3, ok = orbx.ORBmatcher.ProjectPointsKF(T, K, bounds, sc.xw[0], sc.normal[0], Ow)
    use = usable.astype(bool) & ok.astype(bool) & ~(d3 < sc.min_inv[0]) & ~(d3 > sc.max_inv[0])
    lv = orbx.ORBmatcher.PredictScale(sc.mf_max[0], d3, sc.logsf, 8)
    bi, nf = m.FuseSim3(use.astype(np.uint8), u, v, lv, sc.d[0], sc.sf, sc.k[1], sc.d[1], th)
    og = O.KeyFrameGrid(sc.k[1], grid)
    obi, onf = O.fuse_sim3(usable, sc.xw[0], sc.normal[0], sc.min_inv[0], sc.max_inv[0], sc.mf_max[0], sc.d[0], Scw, K, bounds, sc.sf, sc.logsf,
                           og, sc.d[1], th)
    assert nf == onf and np.array_equal(bi, obi)
    if s == 1.0:
        assert nf > 300
    oT, oOw = O.sim3_decompose(Scw)
    assert np.array_equal(T.view(np.uint32), oT.view(np.uint32)) and np.array_equal(Ow.view(np.uint32), oOw.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("s12,rot,th,distorted", [(1.0, 0.0, 7.5, False), (1.02, 0.3, 7.5, True), (0.99, -0.2, 4.0, False)])
def test_search_by_sim3_equals_oracle(orbx, scene, s12, rot, th, distorted):
    """LoopClosing.cc:324 (th = 7.5): the MapPoints of each key frame searched in the other under the relative Sim3, kept where both
    directions agree.  Here the true relative motion is (R12, t12) = (I, (BASE, 0, 0)), s12 = 1."""
    sc = scene
    grid, bounds = _kf_grid(distorted)
    rng = np.random.default_rng(int(s12 * 1000) + int(th))
    n1, n2 = len(sc.k[0]), len(sc.k[1])
    a = np.deg2rad(rot)
    R12 = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]).astype(F32)
    t12 = np.array([BASE, 0.0, 0.0], F32)                                    # x1 = s12 R12 x2 + t12
    usable1 = (rng.random(n1) < 0.85).astype(np.uint8); usable2 = (rng.random(n2) < 0.85).astype(np.uint8)
    m = orbx.ORBmatcher(0.75, True, max_queries=2048, max_train=2048, max_pairs=1 << 16)      # small on purpose: the call grows the handle
    sR12, sR21, t21 = orbx.ORBmatcher.Sim3Relative(s12, R12, t12)
    u1, v1, d1, ok1 = orbx.ORBmatcher.ProjectPointsSim3(sc.T[0], sR21, t21, K, bounds, sc.xw[0])         # KF1's points into KF2
    u2, v2, d2, ok2 = orbx.ORBmatcher.ProjectPointsSim3(sc.T[1], sR12, t12, K, bounds, sc.xw[1])         # KF2's points into KF1
    use1 = usable1.astype(bool) & ok1.astype(bool) & ~(d1 < sc.min_inv[0]) & ~(d1 > sc.max_inv[0])
    use2 = usable2.astype(bool) & ok2.astype(bool) & ~(d2 < sc.min_inv[1]) & ~(d2 > sc.max_inv[1])
    lv1 = orbx.ORBmatcher.PredictScale(sc.mf_max[0], d1, sc.logsf, 8); lv2 = orbx.ORBmatcher.PredictScale(sc.mf_max[1], d2, sc.logsf, 8)
    m12, nf = m.SearchBySim3(use1, u1, v1, lv1, sc.d[0], use2, u2, v2, lv2, sc.d[1], sc.k[0], sc.d[0], grid, sc.sf, sc.k[1], sc.d[1], grid, sc.sf, th)
    g1, g2 = O.KeyFrameGrid(sc.k[0], grid), O.KeyFrameGrid(sc.k[1], grid)
    side = lambda i, us, g: dict(usable=us, xw=sc.xw[i], min_inv=sc.min_inv[i], max_inv=sc.max_inv[i], mf_max=sc.mf_max[i], mp_desc=sc.d[i],
                                 Tw=sc.T[i], bounds=bounds, sf=sc.sf, log_sf=sc.logsf, grid=g, desc=sc.d[i])
    om12, onf = O.search_by_sim3(side(0, usable1, g1), side(1, usable2, g2), K, s12, R12, t12, th)
    assert nf == onf and np.array_equal(m12, om12)
    assert nf == int((m12 >= 0).sum()) and (usable1[m12 >= 0] == 1).all() and (usable2[m12[m12 >= 0]] == 1).all()
    if s12 == 1.0:
        assert nf > 200
        good = m12 >= 0
        dx = sc.k[0]["x"][good] - sc.k[1]["x"][m12[good]]
        assert (np.abs(dx[:, None] - np.array(SHIFTS, F32)).min(1) < 1.5).mean() > 0.8          # matched across the true disparity (2, 4 or 6 px)
    # the handle's first grid slot holds key frame 1 afterwards: a KF1 search needs no rebuild
    T = sc.T[0]
    u, v, iz, d3, ok = orbx.ORBmatcher.ProjectPointsKF(T, K, bounds, sc.xw[1], sc.normal[1], None)
    use = ok.astype(bool) & ~(d3 < sc.min_inv[1]) & ~(d3 > sc.max_inv[1])
    lv = orbx.ORBmatcher.PredictScale(sc.mf_max[1], d3, sc.logsf, 8)
    bi, nfu = m.FuseSim3(use.astype(np.uint8), u, v, lv, sc.d[1], sc.sf, sc.k[0], sc.d[0], 4.0)
    S = np.eye(4, dtype=F32)
    obi, onfu = O.fuse_sim3(np.ones(n2, np.uint8), sc.xw[1], sc.normal[1], sc.min_inv[1], sc.max_inv[1], sc.mf_max[1], sc.d[1], S, K, bounds, sc.sf,
                            sc.logsf, g1, sc.d[0], 4.0)
    assert nfu == onfu and np.array_equal(bi, obi)


@pytest.mark.gpu
@pytest.mark.parametrize("nnratio,check_ori,levelsup", [(0.75, True, 2), (0.9, False, 2), (0.75, True, 3), (0.6, True, 1)])
def test_search_by_bow_kf_equals_oracle(orbx, scene, tmp_path, nnratio, check_ori, levelsup):
    """LoopClosing.cc:266 (ORBmatcher(0.75, true)): both key frames' features need a good MapPoint."""
    sc = scene
    fv = sc.featvecs(orbx, tmp_path, levelsup)
    rng = np.random.default_rng(levelsup + int(nnratio * 100))
    valid1 = (rng.random(len(sc.k[0])) < 0.75).astype(np.uint8); valid2 = (rng.random(len(sc.k[1])) < 0.75).astype(np.uint8)
    m = orbx.ORBmatcher(nnratio, check_ori, max_queries=4096, max_train=4096, max_pairs=1 << 20)
    m12, nm = m.SearchByBoWKF(sc.k[0], sc.d[0], fv[0], valid1, sc.k[1], sc.d[1], fv[1], valid2)
    om12, onm = O.search_by_bow_kf(sc.d[0], sc.k[0]["angle"], valid1, fv[0], sc.d[1], sc.k[1]["angle"], valid2, fv[1], nnratio, check_ori)
    assert nm == onm and np.array_equal(m12, om12)
    assert nm == int((m12 >= 0).sum()) and nm > 30
    assert valid1[m12 >= 0].all() and valid2[m12[m12 >= 0]].all()
    assert len(set(m12[m12 >= 0].tolist())) == nm                            # vbMatched2: one-to-one
    # nobody valid on one side
    z12, zn = m.SearchByBoWKF(sc.k[0], sc.d[0], fv[0], valid1, sc.k[1], sc.d[1], fv[1], np.zeros_like(valid2))
    assert zn == 0 and (z12 == -1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("only_stereo,check_ori,levelsup", [(False, True, 2), (True, True, 2), (False, False, 3), (False, True, 0)])
def test_search_for_triangulation_equals_oracle(orbx, scene, tmp_path, only_stereo, check_ori, levelsup):
    """LocalMapping.cc:270 (ORBmatcher(0.6, false) there; both orientation settings here): unmatched features of two key frames a
    baseline apart, F12 of the true motion, stereo and monocular features, the epipole of camera 1 in image 2."""
    sc = scene
    fv = sc.featvecs(orbx, tmp_path, levelsup)
    rng = np.random.default_rng(levelsup * 2 + int(only_stereo))
    n1, n2 = len(sc.k[0]), len(sc.k[1])
    has1 = (rng.random(n1) < 0.3).astype(np.uint8); has2 = (rng.random(n2) < 0.3).astype(np.uint8)
    ur1 = np.where(rng.random(n1) < 0.5, sc.k[0]["x"] - 20.0, -1.0).astype(F32)
    ur2 = np.where(rng.random(n2) < 0.5, sc.k[1]["x"] - 20.0, -1.0).astype(F32)
    # F12 = K^-T [t12]x R12 K^-1 with R12 = I, t12 = (BASE, 0, 0) (LocalMapping::ComputeF12), in float
    Kinv = np.linalg.inv(np.array([[FX, 0, CX], [0, FY, CY], [0, 0, 1]]))
    tx = np.array([[0, 0, 0], [0, 0, -BASE], [0, BASE, 0]])
    F12 = (Kinv.T @ tx @ Kinv).astype(F32)
    Cw = O.camera_center(sc.T[0])
    Cw = (Cw + np.array([0, 0, 30.0], F32)).astype(F32) if levelsup == 0 else Cw      # an epipole inside image 2 for one case
    m = orbx.ORBmatcher(0.6, check_ori, max_queries=4096, max_train=4096, max_pairs=1 << 20)
    m12, nm = m.SearchForTriangulation(sc.k[0], sc.d[0], has1, ur1, fv[0], sc.k[1], sc.d[1], has2, ur2, fv[1], Cw, sc.T[1], K, F12, sc.sf, sc.sigma2,
                                       only_stereo)
    om12, onm = O.search_for_triangulation(sc.k[0], sc.d[0], has1, ur1, fv[0], sc.k[1], sc.d[1], has2, ur2, fv[1], Cw, sc.T[1], K, F12, sc.sf, sc.sigma2,
                                           only_stereo, check_ori)
    assert nm == onm and np.array_equal(m12, om12)
    assert nm == int((m12 >= 0).sum()) and nm > (20 if only_stereo else 60)
    assert not has1[m12 >= 0].any() and not has2[m12[m12 >= 0]].any()
    if only_stereo:
        assert (ur1[m12 >= 0] >= 0).all() and (ur2[m12[m12 >= 0]] >= 0).all()


@pytest.mark.gpu
def test_matcher_grows_instead_of_refusing(orbx, scene):
    """The reference's matcher has no size limit: a handle created small grows when a call needs more (queries, train descriptors,
    candidate pairs); results equal a handle that was large from the start."""
    sc = scene
    small = orbx.ORBmatcher(0.9, True, max_queries=64, max_train=64, max_pairs=256)
    big = orbx.ORBmatcher(0.9, True, max_queries=8192, max_train=8192, max_pairs=1 << 22)
    for m in (small, big):
        m.grid_build(sc.k[1], 0.0, float(W), 0.0, float(H))
    n = len(sc.k[0])
    x, y = sc.k[0]["x"].copy(), sc.k[0]["y"].copy()
    r = np.full(n, 40.0, F32); lv = np.full(n, -1, np.int32)
    a = small.search_area_best2(sc.d[0], x, y, r, lv, lv, sc.d[1])
    b = big.search_area_best2(sc.d[0], x, y, r, lv, lv, sc.d[1])
    assert all(np.array_equal(p, q) for p, q in zip(a, b))
    pm = np.stack([x, y], 1).astype(F32)
    ra = small.SearchForInitialization(sc.k[0], sc.d[0], sc.k[1], sc.d[1], pm.copy(), 100)
    rb = big.SearchForInitialization(sc.k[0], sc.d[0], sc.k[1], sc.d[1], pm.copy(), 100)
    assert ra[1] == rb[1] and np.array_equal(ra[0], rb[0]) and ra[1] > 100
    da = small.best2(sc.d[0], sc.d[1]); db = big.best2(sc.d[0], sc.d[1])
    assert all(np.array_equal(p, q) for p, q in zip(da, db))
    # more descriptors than the reference-sized default of the adapters' pool (8192)
    rng = np.random.default_rng(0)
    q = rng.integers(0, 256, (9000, 32), dtype=np.uint8); t = rng.integers(0, 256, (8300, 32), dtype=np.uint8)
    bi, bd, sd = small.best2(q, t)
    obi, obd, osd = O.best2(q[:300], t)
    assert np.array_equal(bi[:300], obi) and np.array_equal(bd[:300], obd) and np.array_equal(sd[:300], osd)
    small.reserve(20000, 20000, 1 << 20)
