"""GPU parity: HIP extractor (through the C ABI) == oracle, byte for byte."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _compare(kps, desc, okps, odesc, tag=""):
    assert len(kps) == len(okps), "%s count %d vs oracle %d" % (tag, len(kps), len(okps))
    for name in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        a, b = kps[name], okps[name]
        bad = np.nonzero(a.view(np.uint32 if a.dtype.kind == "f" else a.dtype) !=
                         b.view(np.uint32 if b.dtype.kind == "f" else b.dtype))[0]
        assert bad.size == 0, "%s field %s differs at %s: %s vs %s" % (tag, name, bad[:5], a[bad[:5]], b[bad[:5]])
    assert np.array_equal(desc, odesc), "%s descriptors differ in %d rows" % (
        tag, int((desc != odesc).any(axis=1).sum()))


@pytest.mark.parametrize("W,H,n", [(640, 480, 1000), (1241, 376, 2000), (322, 241, 500)])
def test_extract_matches_oracle(orbx, synth, W, H, n):
    img = synth.texture(1, W, H)
    ex = orbx.ORBextractor(n, 1.2, 8, 20, 7, max_width=W, max_height=H)
    kps, desc = ex(img)
    okps, odesc, npl = O.Extractor(n).extract(img)
    _compare(kps, desc, okps, odesc, "%dx%d" % (W, H))


def test_pyramid_and_candidates_match_oracle(orbx, synth):
    W, H = 640, 480
    img = synth.texture(7, W, H)
    ex = orbx.ORBextractor(1000, max_width=W, max_height=H)
    ex(img)
    oex = O.Extractor(1000)
    opyr = oex.pyramid(img)
    pyr = ex.image_pyramid()
    for l in range(8):
        assert pyr[l].shape == opyr[l].shape
        assert np.array_equal(pyr[l], opyr[l]), "level %d differs" % l
        oc = oex.detect_level(opyr[l])
        gc = ex.candidates(0, l)
        oset = sorted((int(c["x"]) + 16, int(c["y"]) + 16, int(c["response"])) for c in oc)
        gset = sorted(map(tuple, gc.tolist()))
        assert gset == oset, "level %d candidates differ (%d vs %d)" % (l, len(gset), len(oset))


def test_batch_equals_single(orbx, synth):
    frames = synth.stream(4, 640, 480, 4)
    ex = orbx.ORBextractor(1000, max_width=640, max_height=480, max_batch=4)
    res = ex.extract_batch(frames)
    oex = O.Extractor(1000)
    for k in range(4):
        okps, odesc, _ = oex.extract(frames[k])
        _compare(res[k][0], res[k][1], okps, odesc, "frame %d" % k)
