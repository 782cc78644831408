"""No-GPU checks of the C-ABI library: it loads, exports every symbol include/*.h declares, its
host-only helpers agree with the oracle, and with no HIP device it fails loudly (no CPU fallback)."""
import ctypes as C
import glob
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(orb[xmvp]_[a-z0-9_]+)\s*\(", text))
    return names


@pytest.fixture(scope="module")
def built(orbx):
    orbx.build()
    return orbx


def test_library_exports_every_declared_symbol(built):
    lib = C.CDLL(built.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 40
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, "declared in include/*.h but not exported: %s" % missing


def test_library_contains_gfx950_code_object(built):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-S", built.LIB_PATH], capture_output=True, text=True).stdout
    assert ".hip_fatbin" in out
    raw = open(built.LIB_PATH, "rb").read()
    assert b"gfx950" in raw


def test_no_gpu_fails_loudly(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(built.OrbxError) as ei:
        built.ORBextractor(1000)
    assert ei.value.code == built.ORBX_E_HIP and "no CPU path" in str(ei.value)
    with pytest.raises(built.OrbxError) as ei:
        built.ORBmatcher()
    assert ei.value.code == built.ORBX_E_HIP


def test_host_helpers_match_oracle(built):
    L = built.lib()
    OL = O.lib()
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    for i in range(200):
        assert L.orbm_distance(a[i].ctypes.data, b[i].ctypes.data) == OL.oro_descriptor_distance(a[i].ctypes.data, b[i].ctypes.data)
    for trial in range(50):
        hist = rng.integers(0, 40, 30).astype(np.int32)
        ind = np.zeros(3, np.int32)
        assert L.orbm_three_maxima(hist.ctypes.data, 30, ind.ctypes.data) == 0
        i1, i2, i3 = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        OL.oro_three_maxima(hist.ctypes.data_as(C.POINTER(C.c_int)), 30, C.byref(i1), C.byref(i2), C.byref(i3))
        assert list(ind) == [i1.value, i2.value, i3.value]
    nq, nt = 500, 400
    aq = rng.uniform(0, 360, nq).astype(np.float32)
    at = rng.uniform(0, 360, nt).astype(np.float32)
    m = rng.integers(-1, nt, nq).astype(np.int32)
    at[m[m >= 0][:200]] = aq[np.nonzero(m >= 0)[0][:200]] - np.float32(12.0)    # a dominant rotation
    m1, m2 = m.copy(), m.copy()
    n1 = L.orbm_rot_filter(aq.ctypes.data, at.ctypes.data, m1.ctypes.data, nq)
    n2 = OL.oro_rot_filter(aq.ctypes.data, at.ctypes.data, m2.ctypes.data, nq)
    assert n1 == n2 and np.array_equal(m1, m2) and 0 < n1 < int((m >= 0).sum())


def test_null_and_bad_arguments_return_status(built):
    L = built.lib()
    assert L.orbx_create(None, 1000, 1.2, 8, 20, 7, 0, 640, 480, 1) == built.ORBX_E_INVALID
    h = C.c_void_p()
    assert L.orbx_create(C.byref(h), 1000, 1.2, 99, 20, 7, 0, 640, 480, 1) == built.ORBX_E_INVALID
    assert b"bad constructor" in L.orbx_last_error()
    assert L.orbx_capacity(None) == 0 and L.orbx_get_levels(None) == 0
    L.orbx_destroy(None)
    L.orbm_destroy(None)


# Distorted calibrations: the TUM1 / TUM2 values of upstream ORB-SLAM2 and a strong synthetic one (this fork ships a single
# configuration, Examples/Monocular/slam_cfg/config.yaml, with k1 = 0, for which UndistortKeyPoints is the identity)
_CALIBS = [(517.306408, 516.469215, 318.643040, 255.313989, [0.262383, -0.953104, -0.005358, 0.002628, 1.163314], 640, 480),
           (520.908620, 521.007327, 325.141442, 249.701764, [0.231222, -0.784899, -0.003257, -0.000105, 0.917205], 640, 480),
           (700.0, 705.0, 620.0, 190.0, [-0.35, 0.15, 0.001, -0.002, 0.0], 1241, 376)]


@pytest.mark.parametrize("fx,fy,cx,cy,dist,W,H", _CALIBS)
def test_undistort_keypoints_and_bounds_match_oracle(built, fx, fy, cx, cy, dist, W, H):
    """Frame::UndistortKeyPoints / ComputeImageBounds (src/Frame.cc:404-463), host code: bit-identical to the oracle
    restatement of cv::undistortPoints, inverse of the Brown model to 1e-3 px, identity for k1 == 0."""
    rng = np.random.default_rng(int(fx))
    n = 2000
    kps = np.zeros(n, built.KP_DTYPE)
    kps["x"] = rng.uniform(0, W, n).astype(np.float32); kps["y"] = rng.uniform(0, H, n).astype(np.float32)
    kps["octave"] = rng.integers(0, 8, n); kps["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    for nd in (5, 4):
        d = np.array(dist[:nd], np.float32)
        d5 = np.zeros(5, np.float32); d5[:nd] = d
        un = built.UndistortKeyPoints(kps, fx, fy, cx, cy, d)
        oxy = O.undistort_points(np.stack([kps["x"], kps["y"]], 1), fx, fy, cx, cy, d5)
        assert np.array_equal(un["x"].view(np.uint32), oxy[:, 0].view(np.uint32))
        assert np.array_equal(un["y"].view(np.uint32), oxy[:, 1].view(np.uint32))
        for f in ("size", "angle", "response", "octave", "class_id"):
            assert np.array_equal(un[f], kps[f])
        # forward Brown model on the undistorted points gives the measured pixels back (where the iteration converges)
        x = (un["x"].astype(np.float64) - np.float32(cx)) / np.float32(fx); y = (un["y"].astype(np.float64) - np.float32(cy)) / np.float32(fy)
        r2 = x * x + y * y
        k1, k2, p1, p2, k3 = (float(v) for v in d5)
        rad = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
        xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x); yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        err = np.hypot(xd * np.float32(fx) + np.float32(cx) - kps["x"], yd * np.float32(fy) + np.float32(cy) - kps["y"])
        rin2 = ((kps["x"] - cx) / fx) ** 2 + ((kps["y"] - cy) / fy) ** 2        # five iterations converge near the centre only
        central = rin2 < 0.16
        assert central.sum() > 400 and err[central].max() < 2e-2
        b = built.ComputeImageBounds(W, H, fx, fy, cx, cy, d)
        assert b == O.image_bounds(W, H, fx, fy, cx, cy, d5)
        assert b[0] < b[1] and b[2] < b[3]
    # k1 == 0: identity and the plain image rectangle, whatever the other coefficients say (:406-410, :455-461)
    d0 = np.array([0.0, 0.3, 0.01, 0.02, 0.1], np.float32)
    un = built.UndistortKeyPoints(kps, fx, fy, cx, cy, d0)
    assert un.tobytes() == kps.tobytes()
    assert built.ComputeImageBounds(W, H, fx, fy, cx, cy, d0) == (0.0, float(W), 0.0, float(H))
    with pytest.raises(built.OrbxError):
        built.UndistortKeyPoints(kps, fx, fy, cx, cy, np.zeros(3, np.float32))
