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
