"""N3: Frame::ComputeStereoMatches on the GPU == oracle restatement (src/Frame.cc:466-640),
mvuRight / mvDepth compared as bit patterns."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _stereo_pair(synth, W, H, seed, disp):
    """Rectified pair: the right image sees the scene shifted left by `disp(x)` px (depth varies along x)."""
    canvas = synth.texture(seed, W + 128, H)
    left = canvas[:, 32:32 + W].copy()
    right = np.empty_like(left)
    for x0 in range(0, W, 160):          # piecewise-constant disparity: 6, 11, 16, ... px (capped at 36)
        d = min(disp + 5 * (x0 // 160), 36)
        wseg = min(160, W - x0)
        right[:, x0:x0 + wseg] = canvas[:, 32 + x0 + d:32 + x0 + d + wseg]
    noise = (synth.splitmix64(seed + 7, W * H) % np.uint64(5)).astype(np.int16).reshape(H, W) - 2
    right = np.clip(right.astype(np.int16) + noise, 0, 255).astype(np.uint8)
    return left, right


@pytest.mark.parametrize("W,H,n", [(640, 480, 1000), (1241, 376, 2000)])
def test_stereo_matches_equal_oracle(orbx, synth, W, H, n):
    left, right = _stereo_pair(synth, W, H, 31, 6)
    exL = orbx.ORBextractor(n, max_width=W, max_height=H)
    exR = orbx.ORBextractor(n, max_width=W, max_height=H)
    kl, dl = exL(left)
    kr, dr = exR(right)
    fx, bf = 500.0, 40.0 * 500.0 / 100.0      # mbf = baseline * fx; mb = mbf / fx
    mb = np.float32(bf) / np.float32(fx)
    u, d = orbx.ComputeStereoMatches(exL, exR, kl, dl, kr, dr, float(mb), float(bf))
    oex = O.Extractor(n)
    ou, od = O.stereo_matches(oex, kl, dl, kr, dr, oex.pyramid(left), oex.pyramid(right), float(mb), float(bf))
    assert np.array_equal(u.view(np.uint32), ou.view(np.uint32))
    assert np.array_equal(d.view(np.uint32), od.view(np.uint32))
    matched = u >= 0
    assert matched.sum() > 0.3 * len(kl)                       # the rig really matches
    disp = kl["x"][matched] - u[matched]
    assert (disp > 2).mean() > 0.9 and (disp < 40).all()


def test_stereo_no_right_keypoints(orbx, synth):
    left, right = _stereo_pair(synth, 320, 240, 5, 6)
    exL = orbx.ORBextractor(300, max_width=320, max_height=240)
    exR = orbx.ORBextractor(300, max_width=320, max_height=240)
    kl, dl = exL(left)
    exR(np.full((240, 320), 77, np.uint8))
    u, d = orbx.ComputeStereoMatches(exL, exR, kl, dl, kl[:0], dl[:0], 0.08, 40.0)
    assert (u == -1).all() and (d == -1).all()
