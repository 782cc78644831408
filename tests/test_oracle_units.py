"""CPU tests of the oracle (the CPU restatement): hand-checkable cases per stage and
the tables SURVEY.md §8 lists.  The reference ships no tests for this path; these pin the oracle's
own behaviour and the decisions DESIGN.md records at the OpenCV 3.1.0 boundary."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import oracle_lib as O

L = O.lib()


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- A1 constructor tables ----
def test_quotas_levels_umax_match_survey_table():
    ex = O.Extractor(1000)
    assert list(ex.e.quota)[:8] == [217, 181, 151, 126, 105, 87, 73, 60]
    assert list(ex.e.umax) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert sum(2 * u + 1 for u in list(ex.e.umax)[1:]) * 2 + 31 == 749        # the 749-px disc
    assert ex.level_sizes(640, 480) == [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
    assert ex.level_sizes(1920, 1080) == [(1920, 1080), (1600, 900), (1333, 750), (1111, 625), (926, 521), (772, 434), (643, 362), (536, 301)]
    assert ex.level_sizes(1241, 376) == [(1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)]
    assert list(O.Extractor(4000).e.quota)[:8] == [869, 724, 603, 503, 419, 349, 291, 242]
    assert list(O.Extractor(2000).e.quota)[:8] == [434, 362, 302, 251, 209, 175, 145, 122]
    sizes = [int(31 * s) for s in list(ex.e.scale)[:8]]
    assert sizes == [31, 37, 44, 53, 64, 77, 92, 111]


def test_pattern_sha256_and_radius():
    pat = np.ctypeslib.as_array(L.oro_pattern(), shape=(1024,)).astype(np.int8)
    assert hashlib.sha256(pat.tobytes()).hexdigest() == "2164181aea6ff9ac426ca512d5130d15e1f6e3cd47b1cbdd568bbe1e55d49023"
    assert int(pat.sum()) == -406 and int(pat.min()) == -13 and int(pat.max()) == 12
    r = np.sqrt((pat.astype(np.float64).reshape(-1, 2) ** 2).sum(1)).max()
    assert 18.38 < r < 18.39          # the rotated samples reach 18.38 px, not 15


def test_gaussian_kernel_fixed_point():
    assert list(O.Extractor().e.gauss_k) == [18, 34, 49, 55, 49, 34, 18]      # sum 257


# ---- OpenCV scalar helpers ----
def test_cv_round_half_to_even():
    for v, r in [(0.5, 0), (1.5, 2), (2.5, 2), (-0.5, 0), (-1.5, -2), (2.4999, 2), (2.5001, 3), (-2.5, -2)]:
        assert L.oro_cv_round(v) == r


def test_reflect101():
    assert [L.oro_reflect101(p, 5) for p in range(-3, 8)] == [3, 2, 1, 0, 1, 2, 3, 4, 3, 2, 1]


def test_fast_atan2_accuracy_and_quadrants():
    rng = np.random.default_rng(0)
    for _ in range(2000):
        y, x = rng.integers(-3_000_000, 3_000_000, 2)
        a = L.oro_fast_atan2(float(y), float(x))
        ref = np.degrees(np.arctan2(float(y), float(x))) % 360.0
        d = abs(a - ref)
        assert min(d, 360 - d) < 0.3
    assert L.oro_fast_atan2(0.0, 0.0) == 0.0
    assert L.oro_fast_atan2(0.0, 5.0) == 0.0
    assert abs(L.oro_fast_atan2(5.0, 0.0) - 90.0) < 1e-3
    assert abs(L.oro_fast_atan2(0.0, -5.0) - 180.0) < 1e-3
    assert abs(L.oro_fast_atan2(-5.0, 0.0) - 270.0) < 1e-3


def test_sincos_is_correctly_rounded():
    rng = np.random.default_rng(1)
    ang = np.concatenate([rng.uniform(0, 360, 4000), [0, 90, 180, 270, 360, 45, 1e-3]]).astype(np.float32)
    a, b = C.c_float(), C.c_float()
    fpi = np.float32(np.float64(np.pi) / np.float32(180.0))
    for v in ang:
        L.oro_sincos_deg(float(v), C.byref(a), C.byref(b))
        th = np.longdouble(np.float32(v) * fpi)
        assert np.float32(np.cos(th)) == np.float32(a.value)
        assert np.float32(np.sin(th)) == np.float32(b.value)


# ---- resize ----
def _resize(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    L.oro_resize_linear(_p(src), src.shape[1], src.shape[0], src.shape[1], _p(dst), dw, dh, dw)
    return dst


def test_resize_constant_ramp_identity():
    assert (_resize(np.full((48, 60), 77, np.uint8), 50, 40) == 77).all()
    ramp = np.tile(np.arange(0, 240, 2, dtype=np.uint8), (30, 1))           # 120 wide, slope 2
    out = _resize(ramp, 100, 25)
    x = (np.arange(100) + 0.5) * 1.2 - 0.5                                   # source coordinate
    ref = np.clip(x, 0, 119) * 2
    assert np.abs(out[5].astype(np.float64) - ref).max() <= 1.0
    assert np.array_equal(out[0], out[24])
    img = np.random.default_rng(2).integers(0, 256, (33, 47), dtype=np.uint8)
    assert np.array_equal(_resize(img, 47, 33), img)


def test_resize_exact_half_uses_area_average():
    img = np.random.default_rng(3).integers(0, 256, (40, 64), dtype=np.uint8).astype(np.int32)
    ref = (img[0::2, 0::2] + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2
    assert np.array_equal(_resize(img.astype(np.uint8), 32, 20), ref.astype(np.uint8))


def test_resize_fixed_point_formula_spot_check():
    rng = np.random.default_rng(4)
    src = rng.integers(0, 256, (97, 131), dtype=np.uint8)
    dw, dh = 109, 81
    out = _resize(src, dw, dh)
    sx_scale, sy_scale = 1.0 / (dw / 131.0), 1.0 / (dh / 97.0)
    for (dx, dy) in [(0, 0), (108, 80), (50, 40), (1, 79), (107, 3)]:
        fx = np.float32((dx + 0.5) * sx_scale - 0.5); sx = int(np.floor(fx)); fx = np.float32(fx - sx)
        fy = np.float32((dy + 0.5) * sy_scale - 0.5); sy = int(np.floor(fy)); fy = np.float32(fy - sy)
        if sx < 0: sx, fx = 0, np.float32(0)
        if sx >= 130: sx, fx = 130, np.float32(0)
        a0, a1 = int(np.rint((np.float32(1) - fx) * 2048)), int(np.rint(fx * 2048))
        b0, b1 = int(np.rint((np.float32(1) - fy) * 2048)), int(np.rint(fy * 2048))
        r0, r1 = min(max(sy, 0), 96), min(max(sy + 1, 0), 96)
        s1 = min(sx + 1, 130)
        h0 = int(src[r0, sx]) * a0 + int(src[r0, s1]) * a1
        h1 = int(src[r1, sx]) * a0 + int(src[r1, s1]) * a1
        v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2
        assert out[dy, dx] == v


def test_copy_make_border_reflect101():
    img = np.arange(20, dtype=np.uint8).reshape(4, 5)
    out = np.zeros((8, 9), np.uint8)
    L.oro_copy_make_border101(_p(img), 5, 4, 5, _p(out), 9, 2)
    assert np.array_equal(out, np.pad(img, 2, mode="reflect"))


# ---- Gaussian blur ----
def test_blur_constant_and_impulse():
    k = (C.c_int * 7)(18, 34, 49, 55, 49, 34, 18)
    for val in (0, 1, 100, 254, 255):
        src = np.full((20, 24), val, np.uint8)
        dst = np.zeros_like(src)
        L.oro_gaussian_blur7(_p(src), 24, 20, 24, _p(dst), 24, k, 0)
        ref = min((val * 257 * 257 + 32768) >> 16, 255)       # kernel sums to 257 per pass; saturates
        assert (dst == ref).all()
    src = np.zeros((21, 21), np.uint8)
    src[10, 10] = 200
    dst = np.zeros_like(src)
    L.oro_gaussian_blur7(_p(src), 21, 21, 21, _p(dst), 21, k, 0)
    kk = np.array([18, 34, 49, 55, 49, 34, 18])
    ref = (np.outer(kk, kk) * 200 + 32768) >> 16
    assert np.array_equal(dst[7:14, 7:14], ref)
    assert dst.sum() == ref.sum()


def test_blur_rounding_modes_differ_only_on_exact_ties():
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, (64, 70), dtype=np.uint8)
    k = (C.c_int * 7)(18, 34, 49, 55, 49, 34, 18)
    a = np.zeros_like(src); b = np.zeros_like(src)
    L.oro_gaussian_blur7(_p(src), 70, 64, 70, _p(a), 70, k, 0)
    L.oro_gaussian_blur7(_p(src), 70, 64, 70, _p(b), 70, k, 1)
    diff = a.astype(int) - b.astype(int)
    assert set(np.unique(diff)) <= {0, 1}                      # half-up vs half-even
    assert (diff[:, 68:] == 0).all()                           # columns >= (w & ~3) use the scalar path


# ---- FAST ----
def _fast(img, th, nonmax=1):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros(4096, O.CAND_DTYPE)
    n = L.oro_fast9_16(_p(img), img.shape[1], img.shape[1], img.shape[0], th, nonmax, _p(out), 4096)
    return out[:n]


CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def _arc_image(arc_len, delta, center=100, size=15):
    img = np.full((size, size), center, np.uint8)
    c = size // 2
    for k in range(arc_len):
        dx, dy = CIRCLE[k]
        img[c + dy, c + dx] = center + delta
    return img


def _has_center(kp, c=7):
    return any((int(k["x"]), int(k["y"])) == (c, c) for k in kp)


def test_fast_nine_arc_passes_eight_arc_fails():
    # (the arc pixels themselves are corners too -- isolated dots -- so only the centre is checked)
    for delta in (30, -30):
        assert _has_center(_fast(_arc_image(9, delta), 20, nonmax=0))
        assert not _has_center(_fast(_arc_image(8, delta), 20, nonmax=0))
    assert not _has_center(_fast(_arc_image(9, 20), 20, nonmax=0))      # strict: |diff| must exceed t
    assert _has_center(_fast(_arc_image(9, 21), 20, nonmax=0))
    assert _has_center(_fast(_arc_image(9, 8), 7, nonmax=0))
    assert _has_center(_fast(_arc_image(16, 30), 20, nonmax=0))          # full ring


def test_fast_score_is_max_min_minus_one():
    img = _arc_image(9, 0)
    c = 7
    vals = [40, 35, 50, 33, 60, 45, 38, 52, 41]
    for k, v in enumerate(vals):
        dx, dy = CIRCLE[k]
        img[c + dy, c + dx] = 100 + v
    kp = [k for k in _fast(img, 20, nonmax=0) if (int(k["x"]), int(k["y"])) == (c, c)]
    assert len(kp) == 1
    img = np.ascontiguousarray(img)
    assert L.oro_fast_score_pixel(img.ctypes.data + c * 15 + c, 15) == min(vals) - 1
    kp = [k for k in _fast(img, 20, nonmax=1)]
    for k in kp:                                      # with NMS the response is that score
        x, y = int(k["x"]), int(k["y"])
        assert int(k["response"]) == L.oro_fast_score_pixel(img.ctypes.data + y * 15 + x, 15)


def test_fast_score_equals_threshold_free_definition():
    rng = np.random.default_rng(6)
    img = rng.integers(0, 256, (40, 40), dtype=np.uint8)
    for th in (7, 20, 40):
        for k in _fast(img, th, nonmax=1):
            x, y = int(k["x"]), int(k["y"])
            s = L.oro_fast_score_pixel(img.ctypes.data + y * 40 + x, 40)
            assert s == int(k["response"]) and s >= th


def test_fast_nms_is_strict_equal_neighbours_suppress_each_other():
    img = np.full((20, 24), 100, np.uint8)
    for cx in (8, 9):                                   # two adjacent, identical corners
        pass
    a = _arc_image(9, 40, size=20)
    img = np.hstack([a, a])                              # two well separated identical corners: both kept
    assert len(_fast(img, 20)) == 2
    # build two horizontally adjacent pixels with identical scores: constant rows make every pixel of a row equal
    img = np.full((16, 30), 100, np.uint8)
    img[8:, :] = 160                                     # horizontal step edge: score constant along x
    kp = _fast(img, 20)
    assert len(kp) == 0 or len(set(int(k["x"]) for k in kp)) == len(kp)
    nk = _fast(img, 20, nonmax=0)
    if len(nk):                                          # all equal scores in a row => all suppressed
        assert len(kp) == 0


def test_fast_border_three_pixels_never_detected():
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (30, 33), dtype=np.uint8)
    kp = _fast(img, 10)
    assert len(kp) > 0
    assert kp["x"].min() >= 3 and kp["x"].max() <= 33 - 4 and kp["y"].min() >= 3 and kp["y"].max() <= 30 - 4


# ---- cell loop ----
def test_detect_level_cells_partition_and_fallback(synth):
    ex = O.Extractor(1000)
    img = synth.texture(1, 640, 480)
    c = ex.detect_level(img)
    xy = set((int(a["x"]), int(a["y"])) for a in c)
    assert len(xy) == len(c)                                        # every pixel tested in exactly one cell
    assert c["x"].min() >= 3 and c["x"].max() <= 640 - 32 - 4       # coordinates relative to (16,16), inside [19, w-19)
    assert c["y"].min() >= 3 and c["y"].max() <= 480 - 32 - 4
    assert (c["response"] < 20).any() and (c["response"] >= 20).any()   # the 20 -> 7 fallback fires somewhere
    # cells are emitted row-major, rows inside a cell ascending
    w_cell, h_cell = int(np.ceil(608 / 20)), int(np.ceil(448 / 14))
    cell = ((c["y"] - 3) // h_cell) * 20 + (c["x"] - 3) // w_cell
    assert (np.diff(cell) >= 0).all()
    # a cell whose strongest corner is below 20 only emits responses in [7, 20)
    for cid in np.unique(cell):
        r = c["response"][cell == cid]
        assert (r >= 20).all() or (r < 20).all()
        assert r.min() >= 7


def test_detect_level_tiny_level_has_no_cells():
    ex = O.Extractor(100)
    assert len(ex.detect_level(np.random.default_rng(8).integers(0, 256, (50, 200), dtype=np.uint8))) == 0


# ---- quadtree ----
def _cands(pts):
    a = np.zeros(len(pts), O.CAND_DTYPE)
    for i, (x, y, r) in enumerate(pts):
        a[i] = (x, y, r)
    return a


def test_octree_by_hand():
    # one root (100x100), four points in four quadrants, N=4: one split, children pushed front n1..n4
    pts = [(10, 10, 5), (90, 10, 6), (10, 90, 7), (90, 90, 8)]
    sel = O.distribute_octree(_cands(pts), 0, 100, 0, 100, 4)
    assert list(sel) == [3, 2, 1, 0]                    # list order: n4, n3, n2, n1
    # N=1: the loop still runs once (do-while) and splits the root
    assert sorted(O.distribute_octree(_cands(pts), 0, 100, 0, 100, 1)) == [0, 1, 2, 3]
    # two points in the same quadrant, N=2: the split yields ONE child, so lNodes.size()==prevSize and the
    # reference stops (src/ORBextractor.cc:671) although N is not reached -- a quirk the restatement keeps
    pts = [(10, 10, 5), (30, 30, 9)]
    sel = O.distribute_octree(_cands(pts), 0, 100, 0, 100, 2)
    assert list(sel) == [1]
    # the same two points straddling the first split are separated
    sel = O.distribute_octree(_cands([(10, 10, 5), (60, 30, 9)]), 0, 100, 0, 100, 2)
    assert list(sel) == [1, 0]
    # N=1 with a cluster: root splits once into a single child holding both; best response kept
    sel = O.distribute_octree(_cands(pts), 0, 100, 0, 100, 1)
    assert list(sel) == [1]
    # ties in response: the first key in candidate order wins
    pts = [(10, 10, 9), (12, 30, 9), (30, 12, 9)]
    assert list(O.distribute_octree(_cands(pts), 0, 100, 0, 100, 1)) == [0]


def test_octree_two_roots_and_empty():
    pts = [(10, 10, 5), (150, 10, 6)]
    sel = O.distribute_octree(_cands(pts), 0, 200, 0, 100, 2)       # nIni = 2
    assert sorted(sel) == [0, 1]
    assert len(O.distribute_octree(_cands([]), 0, 200, 0, 100, 5)) == 0


def test_octree_never_exceeds_n_plus_3_and_keeps_unique_points(synth):
    ex = O.Extractor(1000)
    c = ex.detect_level(synth.texture(3, 640, 480))
    for N in (1, 17, 217, 1000, 5000):
        sel = O.distribute_octree(c, 16, 640 - 16, 16, 480 - 16, N)
        assert len(set(sel.tolist())) == len(sel)
        assert len(sel) <= max(N + 3, 4)
        if N >= len(c):
            assert len(sel) == len(c)                                # every candidate ends alone in a node


# ---- orientation, descriptor ----
def test_ic_angle_half_planes():
    ex = O.Extractor()
    umax = (C.c_int * 16)(*list(ex.e.umax))
    img = np.zeros((64, 64), np.uint8)
    img[:, 32:] = 200                        # bright to the right -> centroid at +x -> 0 deg
    a = L.oro_ic_angle(_p(img), 64, 31, 32, umax)
    assert a < 1.0 or a > 359.0
    img = np.zeros((64, 64), np.uint8)
    img[32:, :] = 200                        # bright below (+y) -> 90 deg
    assert abs(L.oro_ic_angle(_p(img), 64, 32, 31, umax) - 90.0) < 1.0
    img = np.full((64, 64), 50, np.uint8)
    assert L.oro_ic_angle(_p(img), 64, 32, 32, umax) == 0.0      # m01 = m10 = 0


def test_descriptor_constant_patch_is_zero_and_bit_order():
    img = np.full((64, 64), 123, np.uint8)
    d = np.full(32, 255, np.uint8)
    L.oro_descriptor(_p(img), 64, 32, 32, C.c_float(37.0), _p(d))
    assert (d == 0).all()                    # strict '<' at src/ORBextractor.cc:129
    # one bright pixel at pattern point 1 of pair 0 (angle 0: no rotation) sets bit 0 of byte 0 only...
    pat = np.ctypeslib.as_array(L.oro_pattern(), shape=(1024,)).astype(int)
    x1, y1 = pat[2], pat[3]
    img2 = img.copy()
    img2[32 + y1, 32 + x1] = 200
    L.oro_descriptor(_p(img2), 64, 32, 32, C.c_float(0.0), _p(d))
    bits = np.unpackbits(d, bitorder="little")
    assert bits[0] == 1
    # every set bit must be a pair whose second point is that pixel
    for k in np.nonzero(bits)[0]:
        assert (pat[4 * k + 2], pat[4 * k + 3]) == (x1, y1)


# ---- matcher ----
def test_descriptor_distance_vs_naive_popcount():
    rng = np.random.default_rng(9)
    a = rng.integers(0, 256, (100, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (100, 32), dtype=np.uint8)
    for i in range(100):
        assert L.oro_descriptor_distance(_p(a[i]), _p(b[i])) == int(np.unpackbits(a[i] ^ b[i]).sum())


def test_best2_tie_rules():
    t = np.zeros((4, 32), np.uint8)
    t[0, 0] = 0b111         # d=3
    t[1, 0] = 0b1           # d=1
    t[2, 1] = 0b1           # d=1  (tie with the best: becomes second best, index stays 1)
    t[3, 0] = 0b11          # d=2
    q = np.zeros((1, 32), np.uint8)
    bi, bd, sd = O.best2(q, t)
    assert (bi[0], bd[0], sd[0]) == (1, 1, 1)
    bi, bd, sd = O.best2(q, t[:1])
    assert (bi[0], bd[0], sd[0]) == (0, 3, 256)
    bi, bd, sd = O.best2(q, t[:0])
    assert (bi[0], bd[0], sd[0]) == (-1, 256, 256)
    off = np.array([0, 3], np.int32); idx = np.array([3, 0, 2], np.int32)
    bi, bd, sd = O.best2(q, t, off, idx)
    assert (bi[0], bd[0], sd[0]) == (2, 1, 2)


def test_three_maxima_and_rot_bins():
    h = (C.c_int * 30)(*([0] * 30))
    h[3], h[7], h[20] = 100, 50, 9
    i1, i2, i3 = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    L.oro_three_maxima(h, 30, C.byref(i1), C.byref(i2), C.byref(i3))
    assert (i1.value, i2.value, i3.value) == (3, 7, -1)          # 9 < 0.1*100
    h[20] = 10
    i1, i2, i3 = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    L.oro_three_maxima(h, 30, C.byref(i1), C.byref(i2), C.byref(i3))
    assert (i1.value, i2.value, i3.value) == (3, 7, 20)
    assert L.oro_rot_bin(10.0, 350.0) == 1 and L.oro_rot_bin(350.0, 10.0) == 11
    assert L.oro_rot_bin(359.9, 0.0) == 12 and L.oro_rot_bin(0.0, 0.0) == 0
    assert L.oro_rot_bin(0.0, 0.1) == 12                          # 359.9/30 = 11.997 -> 12
