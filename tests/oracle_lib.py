"""ctypes binding to oracle/liborb_oracle.so (the CPU restatement).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
MAX_LEVELS = 16

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
CAND_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("response", "<i4")])
assert KP_DTYPE.itemsize == 28 and CAND_DTYPE.itemsize == 12


class OroExtractor(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("ini_th_fast", C.c_int), ("min_th_fast", C.c_int),
                ("scale", C.c_float * MAX_LEVELS), ("inv_scale", C.c_float * MAX_LEVELS),
                ("sigma2", C.c_float * MAX_LEVELS), ("inv_sigma2", C.c_float * MAX_LEVELS),
                ("quota", C.c_int * MAX_LEVELS), ("umax", C.c_int * 16), ("gauss_k", C.c_int * 7),
                ("blur_mode", C.c_int)]


_lib = None


def build(native=False):
    target = "liborb_oracle_native.so" if native else "liborb_oracle.so"
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, target])
    return os.path.join(ORACLE_DIR, target)


def lib(native=False):
    global _lib
    if _lib is not None and not native:
        return _lib
    path = build(native)
    L = C.CDLL(path)
    u8p, i32p, f32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_float)
    vp = C.c_void_p
    L.oro_extractor_init.argtypes = [C.POINTER(OroExtractor), C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
    L.oro_extractor_init.restype = C.c_int
    L.oro_level_size.argtypes = [C.POINTER(OroExtractor), C.c_int, C.c_int, C.c_int,
                                 C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.oro_pattern.restype = C.POINTER(C.c_int8)
    L.oro_cv_round.argtypes = [C.c_double]
    L.oro_cv_round.restype = C.c_int
    L.oro_fast_atan2.argtypes = [C.c_float, C.c_float]
    L.oro_fast_atan2.restype = C.c_float
    L.oro_sincos_deg.argtypes = [C.c_float, f32p, f32p]
    L.oro_reflect101.argtypes = [C.c_int, C.c_int]
    L.oro_reflect101.restype = C.c_int
    L.oro_resize_linear.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int]
    L.oro_copy_make_border101.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int]
    L.oro_gaussian_blur7.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.POINTER(C.c_int), C.c_int]
    L.oro_fast9_16.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
    L.oro_fast9_16.restype = C.c_int
    L.oro_fast_score_pixel.argtypes = [vp, C.c_int]
    L.oro_fast_score_pixel.restype = C.c_int
    L.oro_detect_level.argtypes = [C.POINTER(OroExtractor), vp, C.c_int, C.c_int, C.c_int, vp, C.c_int]
    L.oro_detect_level.restype = C.c_int
    L.oro_distribute_octree.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
    L.oro_distribute_octree.restype = C.c_int
    L.oro_ic_angle.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.oro_ic_angle.restype = C.c_float
    L.oro_descriptor.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_float, vp]
    L.oro_extract.argtypes = [C.POINTER(OroExtractor), vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int,
                              C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    L.oro_extract.restype = C.c_int
    L.oro_descriptor_distance.argtypes = [vp, vp]
    L.oro_descriptor_distance.restype = C.c_int
    L.oro_best2.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp]
    L.oro_three_maxima.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.oro_rot_bin.argtypes = [C.c_float, C.c_float]
    L.oro_rot_bin.restype = C.c_int
    L.oro_rot_filter.argtypes = [vp, vp, vp, C.c_int]
    L.oro_rot_filter.restype = C.c_int
    L.oro_search_for_initialization.argtypes = [vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, C.c_int, C.c_float, C.c_int, vp]
    L.oro_search_for_initialization.restype = C.c_int
    L.oro_search_by_projection_last.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                                vp, vp, vp, vp, vp, vp, C.c_int, C.c_float, C.c_int, C.c_int, vp, vp]
    L.oro_search_by_projection_last.restype = C.c_int
    L.oro_search_by_projection_map.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_float, C.c_float, vp, vp]
    L.oro_search_by_projection_map.restype = C.c_int
    L.oro_search_by_projection_kf.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, vp, vp, C.c_int,
                                              C.c_float, vp, vp, vp, C.c_int, C.c_float, C.c_int, C.c_int, vp, vp]
    L.oro_search_by_projection_kf.restype = C.c_int
    L.oro_predict_scale.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int]
    L.oro_predict_scale.restype = C.c_int
    L.oro_undistort_points.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, vp]
    L.oro_undistort_points.restype = None
    L.oro_image_bounds.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, vp, vp]
    L.oro_image_bounds.restype = None
    L.oro_grid_build.argtypes = [vp, vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, vp]
    L.oro_grid_build.restype = None
    L.oro_features_in_area.argtypes = [vp, vp, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, vp, C.c_int]
    L.oro_features_in_area.restype = C.c_int
    L.oro_search_area_best2.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp]
    L.oro_search_area_best2.restype = None
    L.oro_stereo_matches.argtypes = [C.POINTER(OroExtractor), vp, vp, C.c_int, vp, vp, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_float, C.c_float, vp, vp]
    L.oro_stereo_matches.restype = None
    L.oro_match_dense.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_float, C.c_int, vp]
    L.oro_search_by_bow.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, C.c_int,
                                    C.c_float, C.c_int, vp]
    L.oro_search_by_bow.restype = C.c_int
    L.oro_match_dense.restype = C.c_int
    f = C.c_float
    L.oro_gemm_row.argtypes = [vp, C.c_int, vp]
    L.oro_gemm_row.restype = C.c_float
    L.oro_camera_center.argtypes = [vp, vp]
    L.oro_camera_center.restype = None
    L.oro_grid_build_kf.argtypes = [vp, vp, C.c_int, f, f, f, f, f, f, vp]
    L.oro_grid_build_kf.restype = None
    L.oro_sim3_decompose.argtypes = [vp, vp, vp]
    L.oro_sim3_decompose.restype = None
    L.oro_search_by_projection_sim3.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, f, f, f, f, vp, vp, C.c_int, f, vp, vp, vp, C.c_int,
                                                C.c_int, vp, vp]
    L.oro_search_by_projection_sim3.restype = C.c_int
    L.oro_search_by_bow_kf.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, f, C.c_int, vp]
    L.oro_search_by_bow_kf.restype = C.c_int
    L.oro_search_for_triangulation.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int,
                                               vp, vp, f, f, f, f, vp, vp, vp, C.c_int, C.c_int, vp]
    L.oro_search_for_triangulation.restype = C.c_int
    L.oro_fuse.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, f, f, f, f, f, vp, vp, vp, C.c_int, f, vp, vp, vp, vp, C.c_int, f, vp]
    L.oro_fuse.restype = C.c_int
    L.oro_fuse_sim3.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, f, f, f, f, vp, vp, C.c_int, f, vp, vp, vp, C.c_int, f, vp]
    L.oro_fuse_sim3.restype = C.c_int
    side = [C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, f, vp, vp, vp]
    L.oro_search_by_sim3.argtypes = side + side + [f, f, f, f, f, vp, vp, f, vp]
    L.oro_search_by_sim3.restype = C.c_int
    if not native:
        _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Extractor:
    """Mirror of ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) on the oracle."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7,
                 blur_mode=0, native=False):
        self.L = lib(native)
        self.e = OroExtractor()
        rc = self.L.oro_extractor_init(C.byref(self.e), nfeatures, scale_factor, nlevels, ini_th, min_th)
        if rc != 0:
            raise ValueError("oro_extractor_init rc=%d" % rc)
        self.e.blur_mode = blur_mode
        self.nlevels = nlevels
        self.nfeatures = nfeatures

    def level_size(self, W, H, level):
        w, h = C.c_int(), C.c_int()
        self.L.oro_level_size(C.byref(self.e), W, H, level, C.byref(w), C.byref(h))
        return w.value, h.value

    def level_sizes(self, W, H):
        return [self.level_size(W, H, l) for l in range(self.nlevels)]

    def pyramid(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        H, W = img.shape
        out = [np.empty((h, w), dtype=np.uint8) for (w, h) in self.level_sizes(W, H)]
        src = img
        out[0][:] = img
        for l in range(1, self.nlevels):
            h, w = out[l].shape
            self.L.oro_resize_linear(_p(out[l - 1]), out[l - 1].shape[1], out[l - 1].shape[0],
                                     out[l - 1].shape[1], _p(out[l]), w, h, w)
        return out

    def detect_level(self, lvl, cap=1 << 20):
        lvl = np.ascontiguousarray(lvl, dtype=np.uint8)
        h, w = lvl.shape
        out = np.empty(cap, dtype=CAND_DTYPE)
        n = self.L.oro_detect_level(C.byref(self.e), _p(lvl), w, h, w, _p(out), cap)
        if n < 0:
            raise RuntimeError("oro_detect_level overflow")
        return out[:n].copy()

    def extract(self, img, cap=None, want_levels=False):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        H, W = img.shape
        if cap is None:
            cap = self.nfeatures + 4 * self.nlevels + 4096
        kps = np.zeros(cap, dtype=KP_DTYPE)
        desc = np.zeros((cap, 32), dtype=np.uint8)
        n = C.c_int()
        npl = (C.c_int * MAX_LEVELS)()
        levels = None
        lv_ptrs = None
        if want_levels:
            levels = [np.empty((h, w), dtype=np.uint8) for (w, h) in self.level_sizes(W, H)]
            lv_ptrs = (C.c_void_p * self.nlevels)(*[a.ctypes.data for a in levels])
        rc = self.L.oro_extract(C.byref(self.e), _p(img), W, H, img.strides[0], _p(kps), _p(desc), cap,
                                C.byref(n), lv_ptrs, npl)
        if rc != 0:
            raise RuntimeError("oro_extract rc=%d" % rc)
        res = (kps[:n.value].copy(), desc[:n.value].copy(), list(npl)[:self.nlevels])
        return res + (levels,) if want_levels else res


def distribute_octree(cands, minX, maxX, minY, maxY, N):
    L = lib()
    cands = np.ascontiguousarray(cands, dtype=CAND_DTYPE)
    out = np.empty(max(len(cands), 1), dtype=np.int32)
    n = L.oro_distribute_octree(_p(cands), len(cands), minX, maxX, minY, maxY, N, _p(out), len(out))
    if n < 0:
        raise RuntimeError("oro_distribute_octree rc=%d" % n)
    return out[:n].copy()


def best2(q, t, cand_off=None, cand_idx=None):
    L = lib()
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.ascontiguousarray(t, dtype=np.uint8)
    nq, nt = len(q), len(t)
    bi = np.empty(nq, np.int32); bd = np.empty(nq, np.int32); sd = np.empty(nq, np.int32)
    if cand_off is not None:
        cand_off = np.ascontiguousarray(cand_off, dtype=np.int32)
        cand_idx = np.ascontiguousarray(cand_idx, dtype=np.int32)
        L.oro_best2(_p(q), nq, _p(t), nt, _p(cand_off), _p(cand_idx), _p(bi), _p(bd), _p(sd))
    else:
        L.oro_best2(_p(q), nq, _p(t), nt, None, None, _p(bi), _p(bd), _p(sd))
    return bi, bd, sd


def search_by_bow(desc_kf, angle_kf, featvec_kf, desc_f, angle_f, featvec_f, nnratio=0.7, check_ori=True, valid_kf=None):
    """ORBmatcher::SearchByBoW on feature indices (oracle)."""
    L = lib()
    desc_kf = np.ascontiguousarray(desc_kf, np.uint8); desc_f = np.ascontiguousarray(desc_f, np.uint8)
    angle_kf = np.ascontiguousarray(angle_kf, np.float32); angle_f = np.ascontiguousarray(angle_f, np.float32)
    nk, ok, ik = [np.ascontiguousarray(a, np.int32) for a in featvec_kf]
    nf, of, i_f = [np.ascontiguousarray(a, np.int32) for a in featvec_f]
    v = None if valid_kf is None else np.ascontiguousarray(valid_kf, np.uint8)
    match_f = np.full(len(desc_f), -1, np.int32)
    p = lambda a: None if a is None else a.ctypes.data
    nm = L.oro_search_by_bow(p(desc_kf), p(angle_kf), len(desc_kf), p(v), p(nk), p(ok), p(ik), len(nk),
                             p(desc_f), p(angle_f), len(desc_f), p(nf), p(of), p(i_f), len(nf),
                             C.c_float(nnratio), 1 if check_ori else 0, p(match_f))
    return match_f, nm


def match_dense(q, aq, t, at, th=50, nnratio=0.9, check_ori=True):
    L = lib()
    q = np.ascontiguousarray(q, dtype=np.uint8); t = np.ascontiguousarray(t, dtype=np.uint8)
    aq = np.ascontiguousarray(aq, dtype=np.float32); at = np.ascontiguousarray(at, dtype=np.float32)
    m = np.empty(len(q), np.int32)
    n = L.oro_match_dense(_p(q), _p(aq), len(q), _p(t), _p(at), len(t), th, nnratio, int(check_ori), _p(m))
    return n, m


class OroGrid(C.Structure):
    _fields_ = [("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float),
                ("inv_w", C.c_float), ("inv_h", C.c_float), ("n", C.c_int),
                ("cell_start", C.c_int * (64 * 48 + 1)), ("items", C.POINTER(C.c_int))]


class FrameGrid:
    """Frame::AssignFeaturesToGrid / GetFeaturesInArea on the oracle (src/Frame.cc:230-245, 327-380)."""

    def __init__(self, kps_un, min_x, max_x, min_y, max_y):
        self.L = lib()
        self.kps = np.ascontiguousarray(kps_un, KP_DTYPE)
        self.items = np.zeros(max(len(self.kps), 1), np.int32)
        self.g = OroGrid()
        self.L.oro_grid_build(C.byref(self.g), _p(self.kps), len(self.kps), min_x, max_x, min_y, max_y, _p(self.items))

    def features_in_area(self, x, y, r, min_level=-1, max_level=-1):
        out = np.zeros(max(len(self.kps), 1), np.int32)
        n = self.L.oro_features_in_area(C.byref(self.g), _p(self.kps), x, y, r, min_level, max_level, _p(out), len(out))
        assert n >= 0
        return out[:n].copy()

    def search_area_best2(self, qdesc, x, y, r, mn, mx, train_desc, skip=None):
        qdesc = np.ascontiguousarray(qdesc, np.uint8); train_desc = np.ascontiguousarray(train_desc, np.uint8)
        x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y, np.float32); r = np.ascontiguousarray(r, np.float32)
        mn = np.ascontiguousarray(mn, np.int32); mx = np.ascontiguousarray(mx, np.int32)
        nq = len(x)
        bi = np.empty(nq, np.int32); bd = np.empty(nq, np.int32); sd = np.empty(nq, np.int32)
        sk = np.ascontiguousarray(skip, np.uint8) if skip is not None else None
        self.L.oro_search_area_best2(C.byref(self.g), _p(self.kps), _p(train_desc), _p(sk) if sk is not None else None,
                                     _p(qdesc), _p(x), _p(y), _p(r), _p(mn), _p(mx), nq, _p(bi), _p(bd), _p(sd))
        return bi, bd, sd


def search_for_initialization(kps1, desc1, grid2, desc2, prev_matched, window_size, nnratio, check_ori):
    """ORBmatcher::SearchForInitialization on the oracle; grid2 = FrameGrid of frame 2; prev_matched (n1, 2) float32 is
    updated in place.  -> (vnMatches12, nmatches)"""
    kps1 = np.ascontiguousarray(kps1, KP_DTYPE)
    desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
    assert prev_matched.dtype == np.float32 and prev_matched.flags["C_CONTIGUOUS"]
    m12 = np.full(len(kps1), -1, np.int32)
    n = lib().oro_search_for_initialization(_p(kps1), _p(desc1), len(kps1), C.byref(grid2.g), _p(grid2.kps), _p(desc2), len(grid2.kps),
                                            _p(prev_matched), int(window_size), nnratio, int(check_ori), _p(m12))
    return m12, n


def search_by_projection_last(has_point, xw, mp_desc, mp_obs, kps_last, Tcw, Tlw, K, mb, mbf, bounds, scale_factors, grid_cur, desc_cur,
                              cur_obs, th, mono, check_ori, u_right=None):
    """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) on the oracle -> (cur_match, nmatches); cur_obs in/out"""
    has_point = np.ascontiguousarray(has_point, np.uint8); xw = np.ascontiguousarray(xw, np.float32)
    mp_desc = np.ascontiguousarray(mp_desc, np.uint8); mp_obs = np.ascontiguousarray(mp_obs, np.int32)
    kps_last = np.ascontiguousarray(kps_last, KP_DTYPE); desc_cur = np.ascontiguousarray(desc_cur, np.uint8)
    Tcw = np.ascontiguousarray(Tcw, np.float32).reshape(16); Tlw = np.ascontiguousarray(Tlw, np.float32).reshape(16)
    b = np.ascontiguousarray(bounds, np.float32); sf = np.ascontiguousarray(scale_factors, np.float32)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    cm = np.full(len(grid_cur.kps), -1, np.int32)
    fx, fy, cx, cy = K
    n = lib().oro_search_by_projection_last(len(kps_last), _p(has_point), _p(xw), _p(mp_desc), _p(mp_obs), _p(kps_last), _p(Tcw), _p(Tlw),
                                            fx, fy, cx, cy, mb, mbf, _p(b), _p(sf), C.byref(grid_cur.g), _p(grid_cur.kps), _p(desc_cur),
                                            _p(ur) if ur is not None else None, len(grid_cur.kps), th, int(mono), int(check_ori), _p(cur_obs), _p(cm))
    return cm, n


def search_by_projection_map(in_view, proj_x, proj_y, pred_level, view_cos, mp_desc, mp_obs, scale_factors, grid_cur, desc_cur, cur_obs,
                             th, nnratio, proj_xr=None, u_right=None):
    """ORBmatcher::SearchByProjection(F, vpMapPoints, th) on the oracle -> (cur_match, nmatches); cur_obs in/out"""
    f32 = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
    in_view = np.ascontiguousarray(in_view, np.uint8); px, py, pxr, vc = f32(proj_x), f32(proj_y), f32(proj_xr), f32(view_cos)
    lv = np.ascontiguousarray(pred_level, np.int32); mp_desc = np.ascontiguousarray(mp_desc, np.uint8)
    mp_obs = np.ascontiguousarray(mp_obs, np.int32); sf = f32(scale_factors); ur = f32(u_right)
    desc_cur = np.ascontiguousarray(desc_cur, np.uint8)
    cm = np.full(len(grid_cur.kps), -1, np.int32)
    pp = lambda a: _p(a) if a is not None else None
    n = lib().oro_search_by_projection_map(len(in_view), _p(in_view), _p(px), _p(py), pp(pxr), _p(lv), _p(vc), _p(mp_desc), _p(mp_obs), _p(sf),
                                           C.byref(grid_cur.g), _p(grid_cur.kps), _p(desc_cur), pp(ur), len(grid_cur.kps), th, nnratio, _p(cur_obs), _p(cm))
    return cm, n


def search_by_projection_kf(usable, xw, min_dist_inv, max_dist_inv, mf_max_distance, mp_desc, kf_angle, Tcw, K, bounds, scale_factors,
                            log_scale_factor, grid_cur, desc_cur, cur_has_point, th, orb_dist, check_ori):
    """ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) on the oracle -> (cur_match, nmatches);
    cur_has_point (uint8, in/out) = mvpMapPoints[i2] != NULL"""
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    usable = np.ascontiguousarray(usable, np.uint8); xw = f32(xw); mn, mx, mf = f32(min_dist_inv), f32(max_dist_inv), f32(mf_max_distance)
    mp_desc = np.ascontiguousarray(mp_desc, np.uint8); ka = f32(kf_angle); Tcw = f32(Tcw).reshape(16)
    b = f32(bounds); sf = f32(scale_factors); desc_cur = np.ascontiguousarray(desc_cur, np.uint8)
    cm = np.full(len(grid_cur.kps), -1, np.int32)
    fx, fy, cx, cy = K
    n = lib().oro_search_by_projection_kf(len(usable), _p(usable), _p(xw), _p(mn), _p(mx), _p(mf), _p(mp_desc), _p(ka), _p(Tcw), fx, fy, cx, cy,
                                          _p(b), _p(sf), len(sf), log_scale_factor, C.byref(grid_cur.g), _p(grid_cur.kps), _p(desc_cur),
                                          len(grid_cur.kps), th, int(orb_dist), int(check_ori), _p(cur_has_point), _p(cm))
    return cm, n


def stereo_matches(ex, kl, dl, kr, dr, pyrL, pyrR, mb, mbf):
    """Frame::ComputeStereoMatches on the oracle; pyrL/pyrR = lists of contiguous level images."""
    n = len(pyrL)
    PL = (C.c_void_p * n)(*[a.ctypes.data for a in pyrL]); PR = (C.c_void_p * n)(*[a.ctypes.data for a in pyrR])
    lw = (C.c_int * n)(*[a.shape[1] for a in pyrL]); lh = (C.c_int * n)(*[a.shape[0] for a in pyrL])
    kl = np.ascontiguousarray(kl, KP_DTYPE); kr = np.ascontiguousarray(kr, KP_DTYPE)
    dl = np.ascontiguousarray(dl, np.uint8); dr = np.ascontiguousarray(dr, np.uint8)
    u = np.zeros(len(kl), np.float32); d = np.zeros(len(kl), np.float32)
    ex.L.oro_stereo_matches(C.byref(ex.e), _p(kl), _p(dl), len(kl), _p(kr), _p(dr), len(kr), PL, PR, lw, lh, mb, mbf, _p(u), _p(d))
    return u, d


def undistort_points(xy, fx, fy, cx, cy, dist5):
    """oro_undistort_points on an (n, 2) float32 array -> new array"""
    xy = np.ascontiguousarray(xy, np.float32).copy()
    d = np.ascontiguousarray(dist5, np.float32)
    lib().oro_undistort_points(xy.ctypes.data, len(xy), fx, fy, cx, cy, d.ctypes.data)
    return xy


def image_bounds(w, h, fx, fy, cx, cy, dist5):
    d = np.ascontiguousarray(dist5, np.float32)
    b = np.zeros(4, np.float32)
    lib().oro_image_bounds(w, h, fx, fy, cx, cy, d.ctypes.data, b.ctypes.data)
    return tuple(float(v) for v in b)


# ---- the LocalMapping / LoopClosing matchers (oracle/orb_oracle_kf.c) ----
class KeyFrameGrid(FrameGrid):
    """KeyFrame's grid: cells assigned with Frame's float origin, queried with the key frame's int origin (orb_oracle_kf.c)."""

    def __init__(self, kps_un, grid):
        self.L = lib()
        self.kps = np.ascontiguousarray(kps_un, KP_DTYPE)
        self.items = np.zeros(max(len(self.kps), 1), np.int32)
        self.g = OroGrid()
        self.L.oro_grid_build_kf(C.byref(self.g), _p(self.kps), len(self.kps), *[float(v) for v in grid], _p(self.items))


def camera_center(T):
    T = np.ascontiguousarray(T, np.float32).reshape(16)
    Ow = np.zeros(3, np.float32)
    lib().oro_camera_center(_p(T), _p(Ow))
    return Ow


def sim3_decompose(Scw):
    Scw = np.ascontiguousarray(Scw, np.float32).reshape(16)
    T, Ow = np.zeros(16, np.float32), np.zeros(3, np.float32)
    lib().oro_sim3_decompose(_p(Scw), _p(T), _p(Ow))
    return T.reshape(4, 4), Ow


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def _u8(a):
    return np.ascontiguousarray(a, np.uint8)


def _i32(a):
    return np.ascontiguousarray(a, np.int32)


def search_by_projection_sim3(usable, xw, normal, min_inv, max_inv, mf_max, mp_desc, Scw, K, bounds, scale_factors, log_sf, grid, desc_kf,
                              kf_matched, th):
    """SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) on the oracle -> (kf_match, nmatches); kf_matched uint8 in/out"""
    usable, xw, normal, mn, mx, mf, mp_desc = _u8(usable), _f32(xw), _f32(normal), _f32(min_inv), _f32(max_inv), _f32(mf_max), _u8(mp_desc)
    Scw, b, sf, desc_kf = _f32(Scw).reshape(16), _f32(bounds), _f32(scale_factors), _u8(desc_kf)
    km = np.full(len(grid.kps), -1, np.int32)
    fx, fy, cx, cy = K
    n = lib().oro_search_by_projection_sim3(len(usable), _p(usable), _p(xw), _p(normal), _p(mn), _p(mx), _p(mf), _p(mp_desc), _p(Scw), fx, fy, cx, cy,
                                            _p(b), _p(sf), len(sf), log_sf, C.byref(grid.g), _p(grid.kps), _p(desc_kf), len(grid.kps), int(th),
                                            _p(kf_matched), _p(km))
    return km, n


def search_by_bow_kf(desc1, angle1, valid1, fv1, desc2, angle2, valid2, fv2, nnratio, check_ori):
    desc1, desc2, a1, a2, v1, v2 = _u8(desc1), _u8(desc2), _f32(angle1), _f32(angle2), _u8(valid1), _u8(valid2)
    n1, o1, i1 = [_i32(a) for a in fv1]; n2, o2, i2 = [_i32(a) for a in fv2]
    m12 = np.full(len(desc1), -1, np.int32)
    n = lib().oro_search_by_bow_kf(_p(desc1), _p(a1), len(desc1), _p(v1), _p(n1), _p(o1), _p(i1), len(n1),
                                   _p(desc2), _p(a2), len(desc2), _p(v2), _p(n2), _p(o2), _p(i2), len(n2), nnratio, int(check_ori), _p(m12))
    return m12, n


def search_for_triangulation(kps1, desc1, has_mp1, ur1, fv1, kps2, desc2, has_mp2, ur2, fv2, Cw, T2w, K2, F12, sf2, sigma2_2, only_stereo, check_ori):
    kps1 = np.ascontiguousarray(kps1, KP_DTYPE); kps2 = np.ascontiguousarray(kps2, KP_DTYPE)
    desc1, desc2, h1, h2, ur1, ur2 = _u8(desc1), _u8(desc2), _u8(has_mp1), _u8(has_mp2), _f32(ur1), _f32(ur2)
    n1, o1, i1 = [_i32(a) for a in fv1]; n2, o2, i2 = [_i32(a) for a in fv2]
    Cw, T2w, F12, sf2, s2 = _f32(Cw).reshape(3), _f32(T2w).reshape(16), _f32(F12).reshape(9), _f32(sf2), _f32(sigma2_2)
    m12 = np.full(len(desc1), -1, np.int32)
    fx, fy, cx, cy = K2
    n = lib().oro_search_for_triangulation(_p(kps1), _p(desc1), len(desc1), _p(h1), _p(ur1), _p(n1), _p(o1), _p(i1), len(n1),
                                           _p(kps2), _p(desc2), len(desc2), _p(h2), _p(ur2), _p(n2), _p(o2), _p(i2), len(n2),
                                           _p(Cw), _p(T2w), fx, fy, cx, cy, _p(F12), _p(sf2), _p(s2), int(only_stereo), int(check_ori), _p(m12))
    return m12, n


def fuse(usable, xw, normal, min_inv, max_inv, mf_max, mp_desc, Tcw, Ow, K, bf, bounds, scale_factors, inv_sigma2, log_sf, grid, ur_kf, desc_kf, th):
    usable, xw, normal, mn, mx, mf, mp_desc = _u8(usable), _f32(xw), _f32(normal), _f32(min_inv), _f32(max_inv), _f32(mf_max), _u8(mp_desc)
    Tcw, Ow, b, sf, inv, ur_kf, desc_kf = _f32(Tcw).reshape(16), _f32(Ow).reshape(3), _f32(bounds), _f32(scale_factors), _f32(inv_sigma2), _f32(ur_kf), _u8(desc_kf)
    bi = np.full(len(usable), -1, np.int32)
    fx, fy, cx, cy = K
    n = lib().oro_fuse(len(usable), _p(usable), _p(xw), _p(normal), _p(mn), _p(mx), _p(mf), _p(mp_desc), _p(Tcw), _p(Ow), fx, fy, cx, cy, bf, _p(b),
                       _p(sf), _p(inv), len(sf), log_sf, C.byref(grid.g), _p(grid.kps), _p(ur_kf), _p(desc_kf), len(grid.kps), th, _p(bi))
    return bi, n


def fuse_sim3(usable, xw, normal, min_inv, max_inv, mf_max, mp_desc, Scw, K, bounds, scale_factors, log_sf, grid, desc_kf, th):
    usable, xw, normal, mn, mx, mf, mp_desc = _u8(usable), _f32(xw), _f32(normal), _f32(min_inv), _f32(max_inv), _f32(mf_max), _u8(mp_desc)
    Scw, b, sf, desc_kf = _f32(Scw).reshape(16), _f32(bounds), _f32(scale_factors), _u8(desc_kf)
    bi = np.full(len(usable), -1, np.int32)
    fx, fy, cx, cy = K
    n = lib().oro_fuse_sim3(len(usable), _p(usable), _p(xw), _p(normal), _p(mn), _p(mx), _p(mf), _p(mp_desc), _p(Scw), fx, fy, cx, cy, _p(b),
                            _p(sf), len(sf), log_sf, C.byref(grid.g), _p(grid.kps), _p(desc_kf), len(grid.kps), th, _p(bi))
    return bi, n


def search_by_sim3(side1, side2, K, s12, R12, t12, th):
    """side = dict(usable, xw, min_inv, max_inv, mf_max, mp_desc, Tw, bounds, sf, log_sf, grid, desc) -> (match12, nfound)"""
    keep = []

    def args(sd):
        a = [_u8(sd["usable"]), _f32(sd["xw"]), _f32(sd["min_inv"]), _f32(sd["max_inv"]), _f32(sd["mf_max"]), _u8(sd["mp_desc"]),
             _f32(sd["Tw"]).reshape(16), _f32(sd["bounds"]), _f32(sd["sf"]), _u8(sd["desc"])]
        keep.extend(a)
        g = sd["grid"]
        return [len(a[0]), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), _p(a[5]), _p(a[6]), _p(a[7]), _p(a[8]), len(a[8]), sd["log_sf"],
                C.byref(g.g), _p(g.kps), _p(a[9])]
    R12, t12 = _f32(R12).reshape(9), _f32(t12).reshape(3)
    m12 = np.full(len(side1["usable"]), -1, np.int32)
    fx, fy, cx, cy = K
    n = lib().oro_search_by_sim3(*(args(side1) + args(side2) + [fx, fy, cx, cy, s12, _p(R12), _p(t12), th, _p(m12)]))
    return m12, n
