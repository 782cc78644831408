"""Host-side pose solvers of liborbx.so (include/orbp.h) against oracle/pose_oracle.py and against
ground truth.  They are plain host C++ (SURVEY keeps EPnP / pose optimisation on the host), so these tests need no GPU.

PARITY UNPINNED: the reference holds no fixture for PnPsolver / PoseOptimization and cannot be built here; the checker
is an independent numpy/LAPACK restatement of the same reference lines.  Tolerances (fp64 algorithms, fp32 outputs):
  EPnP, n >= 6 generic points:  |R, t| difference <= 1e-8 (both sides fp64; Jacobi vs LAPACK eigenvectors)
  PoseOptimization:             pose difference <= 2e-6 (output is fp32), outlier flags identical
  RANSAC (min_set 6):           identical draw sequence, iteration count and inlier mask; pose <= 2e-6
EPnP on 4 or 5 points uses the exactly-null singular vectors of a rank-8 / rank-10 M^T M, whose basis is the SVD
routine's own choice (OpenCV's in the reference): there only properties are asserted.
"""
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pose_oracle as po  # noqa: E402

FX, FY, CX, CY = 718.856, 718.856, 607.1928, 185.2157      # KITTI 00-02 (Examples/Monocular/KITTI00-02.yaml)
SIGMA2 = (1.2 ** np.arange(8)) ** 2


@pytest.fixture(scope="module")
def ms(orbx):
    orbx.build()
    orbx.lib()
    return orbx


def scene(rng, n, noise=0.0, n_wrong=0):
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    ang = rng.uniform(0, 0.5)
    K = po._skew(ax)
    R = np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * K @ K
    t = rng.uniform(-1, 1, 3)
    pc = np.stack([rng.uniform(-4, 4, n), rng.uniform(-2, 2, n), rng.uniform(4, 20, n)], 1)
    pw = (pc - t) @ R
    u = np.stack([FX * pc[:, 0] / pc[:, 2] + CX, FY * pc[:, 1] / pc[:, 2] + CY], 1) + rng.normal(size=(n, 2)) * noise
    if n_wrong:
        u[:n_wrong] += rng.uniform(30, 80, (n_wrong, 2)) * rng.choice([-1, 1], (n_wrong, 2))
    return R, t, pw, u


def T_of(R, t):
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


class Lcg:
    """deterministic stand-in for libc rand() (31-bit), the same stream on both sides"""
    MAX = 2147483647

    def __init__(self, seed):
        self.s = seed

    def __call__(self):
        self.s = (self.s * 1103515245 + 12345) & 0x7FFFFFFF
        return self.s


# --------------------------------------------------------------------------------------------- EPnP
@pytest.mark.parametrize("n", [6, 7, 8, 12, 50, 300])
def test_epnp_equals_oracle(ms, n):
    rng = np.random.default_rng(100 + n)
    for trial in range(12):
        R, t, pw, u = scene(rng, n, noise=0.6 if trial % 2 else 0.0)
        R1, t1, e1 = ms.epnp(pw, u, FX, FY, CX, CY)
        R2, t2, e2 = po.epnp(pw, u, FX, FY, CX, CY)
        assert np.abs(R1 - R2).max() <= 1e-8 and np.abs(t1 - t2).max() <= 1e-8 and abs(e1 - e2) <= 1e-7
        assert abs(np.linalg.det(R1) - 1) < 1e-9 and np.abs(R1 @ R1.T - np.eye(3)).max() < 1e-9
        if trial % 2 == 0:      # exact data: EPnP is exact
            assert np.abs(R1 - R).max() < 1e-7 and np.abs(t1 - t).max() < 1e-6 and e1 < 1e-6


def test_epnp_minimal_sets_properties(ms):
    """4 points: rank-8 M^T M, basis-dependent.  The result is always a rigid motion, and exact data is solved exactly
    in a good share of the draws (that share is what RANSAC lives on)."""
    rng = np.random.default_rng(7)
    hits = 0
    for trial in range(60):
        R, t, pw, u = scene(rng, 4)
        R1, t1, e1 = ms.epnp(pw, u, FX, FY, CX, CY)
        assert np.isfinite(R1).all() and np.isfinite(t1).all()
        assert np.abs(R1 @ R1.T - np.eye(3)).max() < 1e-8
        hits += np.abs(R1 - R).max() < 1e-5 and np.abs(t1 - t).max() < 1e-4
    assert hits >= 8


def test_epnp_rejects_bad_arguments(ms):
    with pytest.raises(ms.OrbxError):
        ms.epnp(np.zeros((3, 3)), np.zeros((3, 2)), FX, FY, CX, CY)


# --------------------------------------------------------------------------------------------- RANSAC
def test_ransac_parameter_adjustment_equals_oracle(ms):
    rng = np.random.default_rng(3)
    for n in (1, 5, 8, 9, 15, 20, 40, 100, 500, 2000):
        R, t, pw, u = scene(rng, n)
        s2 = SIGMA2[rng.integers(0, 8, n)]
        a = ms.PnPsolver(u, s2, pw, FX, FY, CX, CY)
        b = po.PnpRansac(u, s2, pw, FX, FY, CX, CY, Lcg(1), Lcg.MAX)
        for args in ((), (0.99, 10, 300, 4, 0.5, 5.991), (0.99, 10, 300, 6, 0.5, 5.991), (0.9, 3, 50, 4, 0.05, 2.0)):
            a.SetRansacParameters(*args)
            b.set_parameters(*args)
            st = a.ransac_state()
            assert (st["min_inliers"], st["max_its"]) == (b.min_inliers, b.max_its), (n, args)
            assert np.float32(st["epsilon"]) == np.float32(b.eps)


@pytest.mark.parametrize("n,n_wrong,min_set", [(40, 0, 6), (60, 20, 6), (120, 50, 6), (200, 90, 8), (25, 14, 6)])
def test_ransac_iterate_equals_oracle(ms, n, n_wrong, min_set):
    """Relocalization's use (Tracking.cc:1392-1394,1421-1427): SetRansacParameters(0.99,10,300,minSet,0.5,5.991), then
    iterate(5) until a pose or bNoMore.  min_set 6 / 8 keeps EPnP's null space one-dimensional so both sides agree."""
    rng = np.random.default_rng(n * 7 + n_wrong)
    R, t, pw, u = scene(rng, n, noise=0.5, n_wrong=n_wrong)
    s2 = SIGMA2[rng.integers(0, 8, n)]
    a = ms.PnPsolver(u, s2, pw, FX, FY, CX, CY, rand=Lcg(42), rand_max=Lcg.MAX)
    b = po.PnpRansac(u, s2, pw, FX, FY, CX, CY, Lcg(42), Lcg.MAX)
    a.SetRansacParameters(0.99, 10, 300, min_set, 0.5, 5.991)
    b.set_parameters(0.99, 10, 300, min_set, 0.5, 5.991)
    for call in range(80):
        Ta, nma, ia, na = a.iterate(5)
        Tb, nmb, ib, nb = b.iterate(5)
        assert a.ransac_state()["iterations"] == b.n_iter
        assert (Ta is None) == (Tb is None) and nma == nmb and na == nb, call
        if Ta is not None:
            assert np.array_equal(ia, ib)
            assert np.abs(Ta - Tb).max() <= 2e-6
        if Ta is not None or nma:
            break
    else:
        pytest.fail("RANSAC neither converged nor gave up")


def test_ransac_default_min_set_recovers_pose(ms):
    """The shipped configuration (4-point draws, libc rand()): with 40 % wrong matches the pose comes back."""
    rng = np.random.default_rng(11)
    ok = 0
    for trial in range(6):
        n = 150
        R, t, pw, u = scene(rng, n, noise=0.5, n_wrong=60)
        s2 = SIGMA2[rng.integers(0, 8, n)]
        s = ms.PnPsolver(u, s2, pw, FX, FY, CX, CY)
        s.SetRansacParameters(0.99, 10, 300, 4, 0.5, 5.991)
        T = None
        for call in range(100):
            T, no_more, inl, ninl = s.iterate(5)
            if T is not None or no_more:
                break
        if T is not None and np.abs(T - T_of(R, t)).max() < 0.05 and not inl[:60].any() and ninl >= 80:
            ok += 1
    assert ok >= 5


def test_ransac_too_few_correspondences(ms):
    rng = np.random.default_rng(5)
    R, t, pw, u = scene(rng, 6)
    s = ms.PnPsolver(u, np.ones(6), pw, FX, FY, CX, CY)       # default minInliers 8 > N
    T, no_more, inl, n = s.iterate(5)
    assert T is None and no_more and n == 0
    s = ms.PnPsolver(np.zeros((0, 2)), np.zeros(0), np.zeros((0, 3)), FX, FY, CX, CY)
    T, no_more, inl, n = s.iterate(5)
    assert T is None and no_more


# --------------------------------------------------------------------------------------------- PoseOptimization
@pytest.mark.parametrize("n", [0, 2, 3, 5, 9, 10, 30, 200, 1500])
def test_pose_optimization_equals_oracle(ms, n):
    rng = np.random.default_rng(500 + n)
    for trial in range(6):
        R, t, pw, u = scene(rng, n, noise=0.7, n_wrong=(n // 5 if trial % 2 else 0))
        inv_s2 = (1.0 / SIGMA2[rng.integers(0, 8, n)]).astype(np.float32)
        dR, dt = po._se3_exp(np.concatenate([rng.normal(size=3) * 0.02, rng.normal(size=3) * 0.1]))
        T0 = T_of(dR @ R, dR @ t + dt)
        ur, bf = None, 0.0
        if trial >= 4 and n:                                   # stereo edges mixed in (mvuRight >= 0)
            bf = 386.1448
            pc = pw @ R.T + t
            ur = (u[:, 0] - bf / pc[:, 2] + rng.normal(size=n) * 0.5).astype(np.float32)
            ur[::3] = -1
        T1, o1, n1 = ms.PoseOptimization(u, inv_s2, pw, FX, FY, CX, CY, T0, ur, bf)
        T2, o2, n2 = po.pose_optimization(u, ur, inv_s2, pw, FX, FY, CX, CY, bf, T0)
        assert np.abs(T1 - T2).max() <= 2e-6, (n, trial)
        assert np.array_equal(o1, o2.astype(bool)) and n1 == n2
        if n < 3:
            assert n1 == 0 and np.array_equal(T1, T0)
        if n >= 30:
            assert np.abs(T1 - T_of(R, t)).max() < 0.02        # converged next to the truth
            if trial % 2:
                assert o1[:n // 5].all() and o1.sum() <= n // 5 + max(2, n // 50)


def test_pose_optimization_is_a_fixed_point_at_the_truth(ms):
    rng = np.random.default_rng(9)
    R, t, pw, u = scene(rng, 100)
    T0 = T_of(R, t)
    T1, o, n = ms.PoseOptimization(u, np.ones(100, np.float32), pw, FX, FY, CX, CY, T0)
    assert n == 100 and not o.any() and np.abs(T1 - T0).max() < 1e-5
