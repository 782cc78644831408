"""Oracle vs the committed golden vectors (tests/golden/, made by tests/golden/make_golden.py).
PARITY UNPINNED: the vectors come from the oracle itself (the reference has none); they pin the
oracle against drift and give the GPU box a fixture that does not need the oracle to be rebuilt."""
import glob
import hashlib
import os

import numpy as np
import pytest

import oracle_lib as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(p for p in glob.glob(os.path.join(GOLDEN, "*.npz")) if not os.path.basename(p).startswith("match"))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_oracle_reproduces_golden(path, synth):
    g = np.load(path)
    img = synth.texture(int(g["seed"]), int(g["W"]), int(g["H"]))
    assert hashlib.sha256(img.tobytes()).hexdigest() == str(g["image_sha256"])     # generator is platform independent
    ex = O.Extractor(int(g["nfeatures"]), blur_mode=int(g["blur_mode"]))
    kps, desc, npl, levels = ex.extract(img, want_levels=True)
    assert [hashlib.sha256(l.tobytes()).hexdigest() for l in levels] == list(g["level_sha256"])
    assert npl == list(g["n_per_level"])
    assert kps.tobytes() == g["keypoints"].tobytes()
    assert np.array_equal(desc, g["descriptors"])


def test_oracle_matcher_golden():
    g = np.load(os.path.join(GOLDEN, "match_960x540_n2000.npz"))
    bi, bd, sd = O.best2(g["q"], g["t"])
    assert np.array_equal(bi, g["best_idx"]) and np.array_equal(bd, g["best_d"]) and np.array_equal(sd, g["second_d"])
    n, m = O.match_dense(g["q"], g["angle_q"], g["t"], g["angle_t"], 50, 0.9, True)
    assert n == int(g["nmatches"]) and np.array_equal(m, g["match12"])


def test_native_build_of_oracle_is_bit_identical(synth):
    """The -O3 -march=native build timed as cpu_baseline must compute the same thing."""
    img = synth.texture(1, 640, 480)
    a = O.Extractor(1000).extract(img)
    b = O.Extractor(1000, native=True).extract(img)
    assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1])
