"""my-slam_amd: MI355X-native ORB front-end (extractor + Hamming matcher) behind a C ABI.

The product is `lib/liborbx.so` (hand-written HIP for gfx950, see csrc/).  This module is only the
ctypes plumbing the tests and bench.py use to reach that C ABI from Python; there is no CPU path:
if the library is missing or no HIP device is present every entry point raises.

The directory name carries a hyphen (the framework's name); load it with
    importlib.util.spec_from_file_location("my_slam_amd", "<repo>/my-slam_amd/__init__.py")
or via tests/conftest.py / the loader at the top of bench.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# ORBX_LIB: developer switch for A/B builds of the same library (tools/dbg/ab_variants.sh); never a different back end
LIB_PATH = os.environ.get("ORBX_LIB") or os.path.join(PKG_DIR, "lib", "liborbx.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

ORBX_OK = 0
ORBX_E_INVALID, ORBX_E_CAPACITY, ORBX_E_SHAPE, ORBX_E_HIP, ORBX_E_CAND_OVERFLOW, ORBX_E_TREE_OVERFLOW = -1, -2, -3, -4, -5, -6
ORBX_OPT_BLUR_ROUNDING = 1
ORBX_OPT_SUBBATCHES = 2
ORBX_OPT_OVERLAP_PYRAMID = 3
ORBX_OPT_BATCH_CHUNK = 4


class OrbxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("orbx status %d: %s" % (code, msg))
        self.code = code


def build(verbose=False):
    """Compile lib/liborbx.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", PKG_DIR] + ([] if verbose else ["-s"]))
    return LIB_PATH


_lib = None


def lib():
    """The loaded C-ABI library.  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OrbxError(ORBX_E_HIP, "%s is missing: run __graft_entry__.build() / make -C my-slam_amd" % LIB_PATH)
    try:
        # PyTorch-ROCm wheels carry their own libamdhip64; whichever copy is mapped first serves the whole
        # process.  Loading torch first keeps `torch.cuda` usable next to liborbx (bench.py, tests).
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    vp, ip = C.c_void_p, C.POINTER(C.c_int)
    L.orbx_create.argtypes = [C.POINTER(vp), C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orbx_create.restype = C.c_int
    L.orbx_destroy.argtypes = [vp]
    L.orbx_destroy.restype = None
    L.orbx_set_option.argtypes = [vp, C.c_int, C.c_int]
    L.orbx_get_levels.argtypes = [vp]
    L.orbx_get_scale_factor.argtypes = [vp]
    L.orbx_get_scale_factor.restype = C.c_float
    L.orbx_get_tables.argtypes = [vp, vp, vp, vp, vp]
    L.orbx_get_features_per_level.argtypes = [vp, vp]
    L.orbx_capacity.argtypes = [vp]
    L.orbx_extract.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, ip]
    L.orbx_extract_begin.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
    L.orbx_extract_begin.restype = C.c_int
    L.orbx_extract_end.argtypes = [vp, vp, vp, C.c_int, ip]
    L.orbx_extract_end.restype = C.c_int
    L.orbx_extract_batch.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, vp, vp, C.c_int, vp]
    L.orbx_extract_batch_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, vp, vp, C.c_int, vp, vp, vp]
    L.orbx_level_size.argtypes = [vp, C.c_int, ip, ip]
    L.orbx_download_level.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int]
    L.orbx_extract_batch_multi.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, vp, vp, C.c_int, vp]
    L.orbx_extract_batch_multi.restype = C.c_int
    L.orbx_download_pyramid.argtypes = [vp, C.c_int, vp, vp, C.c_int]
    L.orbx_download_pyramid.restype = C.c_int
    L.orbx_download_candidates.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
    L.orbx_last_stage_ms.argtypes = [vp, vp]
    L.orbx_set_profiling.argtypes = [vp, C.c_int]
    L.orbx_stage_ms_ring.argtypes = [vp, vp, C.c_int]
    L.orbx_last_error.restype = C.c_char_p
    L.orbx_version.restype = C.c_char_p
    L.orbx_debug_sincos.argtypes = [vp, vp, vp, C.c_int]
    L.orbx_stereo_matches.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_float, C.c_float, vp, vp]
    L.orbx_stereo_matches.restype = C.c_int
    for name in ("orbx_set_option", "orbx_get_levels", "orbx_get_tables", "orbx_get_features_per_level",
                 "orbx_capacity", "orbx_extract", "orbx_extract_batch", "orbx_extract_batch_device",
                 "orbx_level_size", "orbx_download_level", "orbx_download_candidates", "orbx_last_stage_ms",
                 "orbx_set_profiling", "orbx_debug_sincos"):
        getattr(L, name).restype = C.c_int
    _bind_matcher(L)
    _bind_voc(L)
    _bind_pose(L)
    _lib = L
    return L


def _bind_voc(L):
    vp = C.c_void_p
    if not hasattr(L, "orbv_load_text"):
        return
    L.orbv_load_text.argtypes = [C.POINTER(vp), C.c_char_p, C.c_int]
    L.orbv_destroy.argtypes = [vp]
    L.orbv_destroy.restype = None
    L.orbv_info.argtypes = [vp] + [C.POINTER(C.c_int)] * 6
    L.orbv_transform_features.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp]
    L.orbv_bow_vector.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int]
    L.orbv_feature_vector.argtypes = [vp, vp, C.c_int, vp, vp, vp, C.c_int]
    L.orbv_score_l1.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int]
    L.orbv_score_l1.restype = C.c_double
    L.orbv_last_error.restype = C.c_char_p
    for name in ("orbv_load_text", "orbv_info", "orbv_transform_features", "orbv_bow_vector", "orbv_feature_vector"):
        getattr(L, name).restype = C.c_int


RAND_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


def _bind_pose(L):
    vp, ip = C.c_void_p, C.POINTER(C.c_int)
    if not hasattr(L, "orbp_pnp_create"):
        return
    L.orbp_pnp_create.argtypes = [C.POINTER(vp), C.c_int, vp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float]
    L.orbp_pnp_destroy.argtypes = [vp]
    L.orbp_pnp_destroy.restype = None
    L.orbp_pnp_set_rand.argtypes = [vp, RAND_FN, vp, C.c_int]
    L.orbp_pnp_set_rand.restype = None
    L.orbp_pnp_set_ransac_parameters.argtypes = [vp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]
    L.orbp_pnp_get_ransac_state.argtypes = [vp, ip, ip, C.POINTER(C.c_float), ip]
    L.orbp_pnp_get_ransac_state.restype = None
    L.orbp_pnp_iterate.argtypes = [vp, C.c_int, ip, vp, ip, vp]
    L.orbp_pnp_find.argtypes = [vp, vp, ip, vp]
    L.orbp_epnp.argtypes = [C.c_int, vp, vp, C.c_double, C.c_double, C.c_double, C.c_double, vp, vp]
    L.orbp_epnp.restype = C.c_double
    L.orbp_pose_optimization.argtypes = [C.c_int, vp, vp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, vp, vp]
    L.orbp_last_error.restype = C.c_char_p
    for name in ("orbp_pnp_create", "orbp_pnp_set_ransac_parameters", "orbp_pnp_iterate", "orbp_pnp_find",
                 "orbp_pose_optimization"):
        getattr(L, name).restype = C.c_int


def _bind_matcher(L):
    vp = C.c_void_p
    if not hasattr(L, "orbm_create"):
        return
    L.orbm_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int]
    L.orbm_destroy.argtypes = [vp]
    L.orbm_destroy.restype = None
    L.orbm_distance.argtypes = [vp, vp]
    L.orbm_best2.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp]
    L.orbm_distances.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, vp, vp]
    L.orbm_best2_batch_device.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]
    L.orbm_match_batch_device.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, vp, vp, vp]
    L.orbm_grid_build.argtypes = [vp, vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
    L.orbm_features_in_area.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, C.c_int]
    L.orbm_search_area_best2.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp]
    L.orbm_search_area_best2_device.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp]
    L.orbm_rot_filter.argtypes = [vp, vp, vp, C.c_int]
    L.orbm_search_by_bow.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int,
                                     vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_float, C.c_int, vp, vp]
    L.orbm_three_maxima.argtypes = [vp, C.c_int, vp]
    L.orbm_search_for_initialization.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, C.c_int, C.c_float, C.c_int, vp, vp]
    L.orbm_search_for_initialization.restype = C.c_int
    L.orbm_search_by_projection_last.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float,
                                                 C.c_float, C.c_float, vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_float, C.c_int, C.c_int, vp, vp, vp]
    L.orbm_search_by_projection_last.restype = C.c_int
    L.orbm_search_by_projection_map.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int,
                                                C.c_float, C.c_float, vp, vp, vp]
    L.orbm_search_by_projection_map.restype = C.c_int
    L.orbm_project_points.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_float, vp, vp, C.c_int, vp, vp, vp, vp, vp]
    L.orbm_project_points.restype = C.c_int
    L.orbm_predict_scale.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int]
    L.orbm_predict_scale.restype = C.c_int
    L.orbm_search_by_projection_kf.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_float, C.c_int, C.c_int,
                                               vp, vp, vp]
    L.orbm_search_by_projection_kf.restype = C.c_int
    L.orbm_undistort_keypoints.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, vp, C.c_int, vp]
    L.orbm_undistort_keypoints.restype = C.c_int
    L.orbm_image_bounds.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, vp, C.c_int, vp]
    L.orbm_image_bounds.restype = C.c_int
    f = C.c_float
    L.orbm_reserve.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.orbm_grid_build_kf.argtypes = [vp, vp, C.c_int, f, f, f, f, f, f]
    L.orbm_sim3_decompose.argtypes = [vp, vp, vp]
    L.orbm_sim3_relative.argtypes = [f, vp, vp, vp, vp, vp]
    L.orbm_project_points_kf.argtypes = [vp, vp, f, f, f, f, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp]
    L.orbm_project_points_sim3.argtypes = [vp, vp, vp, f, f, f, f, vp, vp, C.c_int, vp, vp, vp, vp]
    L.orbm_search_by_projection_sim3.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, vp]
    L.orbm_search_by_bow_kf.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int,
                                        vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, f, C.c_int, vp, vp]
    L.orbm_search_for_triangulation.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int,
                                                vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int,
                                                vp, vp, f, f, f, f, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]
    L.orbm_fuse.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, f, vp, vp]
    L.orbm_fuse_sim3.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, f, vp, vp]
    L.orbm_search_by_sim3.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp,
                                      vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, C.c_int, f, vp, vp]
    for name in ("orbm_reserve", "orbm_grid_build_kf", "orbm_sim3_decompose", "orbm_sim3_relative", "orbm_project_points_kf",
                 "orbm_project_points_sim3", "orbm_search_by_projection_sim3", "orbm_search_by_bow_kf", "orbm_search_for_triangulation",
                 "orbm_fuse", "orbm_fuse_sim3", "orbm_search_by_sim3", "orbm_search_by_bow"):
        getattr(L, name).restype = C.c_int
    L.orbm_last_error.restype = C.c_char_p
    for name in ("orbm_create", "orbm_distance", "orbm_best2", "orbm_distances", "orbm_best2_batch_device",
                 "orbm_match_batch_device", "orbm_rot_filter", "orbm_three_maxima", "orbm_grid_build",
                 "orbm_features_in_area", "orbm_search_area_best2", "orbm_search_area_best2_device"):
        getattr(L, name).restype = C.c_int


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _chk(rc):
    if rc != ORBX_OK:
        raise OrbxError(rc, lib().orbx_last_error().decode())


class ORBextractor:
    """Python mirror of ORB_SLAM2::ORBextractor (include/ORBextractor.h:46-112) over the C ABI."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7,
                 device=0, max_width=640, max_height=480, max_batch=1):
        self.L = lib()
        self.h = C.c_void_p()
        _chk(self.L.orbx_create(C.byref(self.h), nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST,
                                device, max_width, max_height, max_batch))
        self.nlevels = nlevels
        self.max_batch = max_batch
        self.cap = self.L.orbx_capacity(self.h)

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.orbx_destroy(self.h)
            self.h = None

    __del__ = close

    # getters (include/ORBextractor.h:63-83)
    def GetLevels(self):
        return self.L.orbx_get_levels(self.h)

    def GetScaleFactor(self):
        return self.L.orbx_get_scale_factor(self.h)

    def _tables(self):
        t = [np.zeros(self.nlevels, np.float32) for _ in range(4)]
        _chk(self.L.orbx_get_tables(self.h, *[_p(a) for a in t]))
        return t

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def features_per_level(self):
        q = np.zeros(self.nlevels, np.int32)
        _chk(self.L.orbx_get_features_per_level(self.h, _p(q)))
        return q

    def set_blur_rounding(self, mode):
        _chk(self.L.orbx_set_option(self.h, ORBX_OPT_BLUR_ROUNDING, mode))

    def set_subbatches(self, n):
        _chk(self.L.orbx_set_option(self.h, ORBX_OPT_SUBBATCHES, n))

    def set_batch_chunk(self, frames):
        """Frames per chunk of extract_batch's upload / extract / download pipeline (0 = one piece)."""
        _chk(self.L.orbx_set_option(self.h, ORBX_OPT_BATCH_CHUNK, int(frames)))

    def set_overlap_pyramid(self, on):
        _chk(self.L.orbx_set_option(self.h, ORBX_OPT_OVERLAP_PYRAMID, int(on)))

    def set_profiling(self, on):
        _chk(self.L.orbx_set_profiling(self.h, int(on)))

    def stage_ms(self):
        ms = np.zeros(4, np.float32)
        _chk(self.L.orbx_last_stage_ms(self.h, _p(ms)))
        return ms

    def stage_ms_ring(self, max_calls=16):
        """Stage times of the last calls under set_profiling(2) (rows: newest first; the stream must be synchronised)."""
        ms = np.zeros((max_calls, 4), np.float32)
        n = self.L.orbx_stage_ms_ring(self.h, _p(ms), max_calls)
        _chk(min(n, 0))
        return ms[:n]

    def __call__(self, image, mask=None):
        """operator()(image, mask, keypoints, descriptors) -> (keypoints[KP_DTYPE], descriptors[n,32])."""
        if image is None or image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2   # assert(image.type() == CV_8UC1), :1052
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        H, W = image.shape
        kps = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = C.c_int()
        _chk(self.L.orbx_extract(self.h, _p(image), W, H, image.strides[0], _p(kps), _p(desc), self.cap, C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_begin(self, image):
        """First half of __call__: stage + enqueue, no wait (include/orbx.h: orbx_extract_begin)."""
        if image is None or image.size == 0:
            _chk(self.L.orbx_extract_begin(self.h, None, 0, 0, 0))
            return
        assert image.dtype == np.uint8 and image.ndim == 2 and image.strides[1] == 1
        _chk(self.L.orbx_extract_begin(self.h, image.ctypes.data_as(C.c_void_p), image.shape[1], image.shape[0], image.strides[0]))

    def extract_end(self):
        """Second half: wait and return (keypoints, descriptors)."""
        kps = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = C.c_int(0)
        _chk(self.L.orbx_extract_end(self.h, _p(kps), _p(desc), self.cap, C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch(self, images):
        images = np.ascontiguousarray(images, dtype=np.uint8)
        B, H, W = images.shape
        kps = np.zeros((B, self.cap), KP_DTYPE)
        desc = np.zeros((B, self.cap, 32), np.uint8)
        counts = np.zeros(B, np.int32)
        _chk(self.L.orbx_extract_batch(self.h, _p(images), B, W, H, images.strides[1], images.strides[0],
                                       _p(kps), _p(desc), self.cap, _p(counts)))
        return [(kps[k, :counts[k]].copy(), desc[k, :counts[k]].copy()) for k in range(B)]

    def extract_batch_raw(self, images):
        """orbx_extract_batch without the per-frame slicing: (kps [B,cap], desc [B,cap,32], counts [B]).  A row / frame
        pitch larger than the width (a view into a bigger array) is passed through as it is.
        The three arrays are this object's reusable buffers (the zero-allocation benchmark path): the next call with the same
        batch size overwrites them in place, so copy what must survive it; rows at and beyond counts[f] are stale data of
        earlier calls."""
        images = np.asarray(images, dtype=np.uint8)
        if images.strides[2] != 1 or images.strides[1] < images.shape[2] or images.strides[0] < images.strides[1] * (images.shape[1] - 1) + images.shape[2]:
            images = np.ascontiguousarray(images)
        B, H, W = images.shape
        if getattr(self, "_raw", None) is None or self._raw[0].shape[0] != B:
            self._raw = (np.zeros((B, self.cap), KP_DTYPE), np.zeros((B, self.cap, 32), np.uint8), np.zeros(B, np.int32))
        kps, desc, counts = self._raw
        _chk(self.L.orbx_extract_batch(self.h, _p(images), B, W, H, images.strides[1], images.strides[0],
                                       _p(kps), _p(desc), self.cap, _p(counts)))
        return kps, desc, counts

    def extract_batch_device(self, d_images_ptr, B, W, H, row_stride, frame_stride,
                             d_kps_ptr, d_desc_ptr, d_counts_ptr, d_status_ptr, stream=None):
        """All pointers are device addresses (e.g. torch tensor .data_ptr()); asynchronous."""
        _chk(self.L.orbx_extract_batch_device(self.h, d_images_ptr, B, W, H, row_stride, frame_stride,
                                              d_kps_ptr, d_desc_ptr, self.cap, d_counts_ptr, d_status_ptr, stream))

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        _chk(self.L.orbx_level_size(self.h, level, C.byref(w), C.byref(h)))
        return w.value, h.value

    def image_pyramid(self, frame=0, border=0):
        """mvImagePyramid of the last call (host copies)."""
        out = []
        for l in range(self.nlevels):
            w, h = self.level_size(l)
            a = np.zeros((h + 2 * border, w + 2 * border), np.uint8)
            _chk(self.L.orbx_download_level(self.h, frame, l, _p(a), a.strides[0], border))
            out.append(a)
        return out

    def image_pyramid_all(self, frame=0, border=19):
        """All levels in one call (orbx_download_pyramid: one synchronisation), with the reference's 19-px reflect-101 border."""
        out = []
        for l in range(self.nlevels):
            w, h = self.level_size(l)
            out.append(np.zeros((h + 2 * border, w + 2 * border), np.uint8))
        ptrs = (C.c_void_p * self.nlevels)(*[a.ctypes.data for a in out])
        strides = (C.c_int * self.nlevels)(*[a.strides[0] for a in out])
        _chk(self.L.orbx_download_pyramid(self.h, frame, ptrs, strides, border))
        return out

    def candidates(self, frame, level, cap=1 << 20):
        a = np.zeros((cap, 3), np.int32)
        n = self.L.orbx_download_candidates(self.h, frame, level, _p(a), cap)
        if n < 0:
            _chk(n)
        return a[:n].copy()


def extract_batch_multi(extractors, images):
    """orbx_extract_batch_multi: one batch of host frames sharded over several ORBextractor handles (one per GPU; one host thread
    each).  Returns (kps [B, cap], desc [B, cap, 32], counts [B]) like ORBextractor.extract_batch_raw."""
    images = np.ascontiguousarray(images, dtype=np.uint8)
    B, H, W = images.shape
    cap = max(e.cap for e in extractors)
    kps = np.zeros((B, cap), KP_DTYPE); desc = np.zeros((B, cap, 32), np.uint8); counts = np.zeros(B, np.int32)
    hs = (C.c_void_p * len(extractors))(*[e.h for e in extractors])
    _chk(lib().orbx_extract_batch_multi(hs, len(extractors), _p(images), B, W, H, images.strides[1], images.strides[0],
                                        _p(kps), _p(desc), cap, _p(counts)))
    return kps, desc, counts


def ComputeStereoMatches(left, right, kl, dl, kr, dr, mb, mbf):
    """Frame::ComputeStereoMatches on the two extractors' last pyramids -> (mvuRight, mvDepth)."""
    kl = np.ascontiguousarray(kl, KP_DTYPE); kr = np.ascontiguousarray(kr, KP_DTYPE)
    dl = np.ascontiguousarray(dl, np.uint8); dr = np.ascontiguousarray(dr, np.uint8)
    u = np.zeros(len(kl), np.float32); d = np.zeros(len(kl), np.float32)
    _chk(lib().orbx_stereo_matches(left.h, right.h, _p(kl), _p(dl), len(kl), _p(kr), _p(dr), len(kr), mb, mbf, _p(u), _p(d)))
    return u, d


def debug_sincos(theta):
    theta = np.ascontiguousarray(theta, np.float32)
    c = np.empty_like(theta)
    s = np.empty_like(theta)
    _chk(lib().orbx_debug_sincos(_p(theta), _p(c), _p(s), len(theta)))
    return c, s


def _mchk(rc):
    if rc != ORBX_OK:
        raise OrbxError(rc, lib().orbm_last_error().decode())


class ORBmatcher:
    """Python mirror of the Hamming primitives of ORB_SLAM2::ORBmatcher (include/ORBmatcher.h:41-89)."""
    TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30

    def __init__(self, nnratio=0.6, checkOri=True, device=0, max_queries=8192, max_train=8192, max_pairs=1 << 22):
        self.L = lib()
        self.mfNNratio, self.mbCheckOrientation = float(nnratio), bool(checkOri)
        self.h = C.c_void_p()
        _mchk(self.L.orbm_create(C.byref(self.h), device, max_queries, max_train, max_pairs))

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.orbm_destroy(self.h)
            self.h = None

    __del__ = close

    @staticmethod
    def DescriptorDistance(a, b):
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        return lib().orbm_distance(_p(a), _p(b))

    def best2(self, q, t, cand_off=None, cand_idx=None):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        nq, nt = len(q), len(t)
        bi, bd, sd = (np.full(nq, -1, np.int32), np.full(nq, 256, np.int32), np.full(nq, 256, np.int32))
        if cand_off is not None:
            cand_off = np.ascontiguousarray(cand_off, np.int32)
            cand_idx = np.ascontiguousarray(cand_idx, np.int32)
        _mchk(self.L.orbm_best2(self.h, _p(q), nq, _p(t), nt, _p(cand_off), _p(cand_idx), _p(bi), _p(bd), _p(sd)))
        return bi, bd, sd

    def distances(self, q, t, cand_off=None, cand_idx=None):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        nq, nt = len(q), len(t)
        if cand_off is not None:
            cand_off = np.ascontiguousarray(cand_off, np.int32)
            cand_idx = np.ascontiguousarray(cand_idx, np.int32)
            out = np.zeros(int(cand_off[-1]), np.int32)
        else:
            out = np.zeros(nq * nt, np.int32)
        _mchk(self.L.orbm_distances(self.h, _p(q), nq, _p(t), nt, _p(cand_off), _p(cand_idx), _p(out)))
        return out

    def match_dense(self, q, kq, t, kt, th=None):
        """Dense SearchByBoW-style acceptance + rotation filter on host buffers (via best2)."""
        bi, bd, sd = self.best2(q, t)
        th = self.TH_LOW if th is None else th
        m = np.where((bd <= th) & (bd.astype(np.float32) < np.float32(self.mfNNratio) * sd.astype(np.float32)), bi, -1).astype(np.int32)
        if self.mbCheckOrientation:
            aq = np.ascontiguousarray(kq["angle"], np.float32)
            at = np.ascontiguousarray(kt["angle"], np.float32)
            n = self.L.orbm_rot_filter(_p(aq), _p(at), _p(m), len(m))
        else:
            n = int((m >= 0).sum())
        return n, m

    def match_batch_device(self, d_q, d_kq, d_nq, d_t, d_kt, d_nt, cap, nbatch, d_match12, d_nmatches,
                           th=None, stream=None):
        th = self.TH_LOW if th is None else th
        _mchk(self.L.orbm_match_batch_device(self.h, d_q, d_kq, d_nq, d_t, d_kt, d_nt, cap, nbatch, th,
                                             self.mfNNratio, int(self.mbCheckOrientation), d_match12, d_nmatches, stream))

    # ---- N1: Frame grid (src/Frame.cc:230-245, 327-392) ----
    def grid_build(self, kps_un, min_x, max_x, min_y, max_y):
        """Frame::AssignFeaturesToGrid for mvKeysUn with the image bounds mnMinX..mnMaxY."""
        kps_un = np.ascontiguousarray(kps_un, KP_DTYPE)
        self._grid_n = len(kps_un)
        _mchk(self.L.orbm_grid_build(self.h, _p(kps_un), len(kps_un), min_x, max_x, min_y, max_y))

    @staticmethod
    def _windows(x, y, r, min_level, max_level):
        x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y, np.float32)
        r = np.broadcast_to(np.asarray(r, np.float32), x.shape).copy()
        mn = np.broadcast_to(np.asarray(min_level, np.int32), x.shape).copy()
        mx = np.broadcast_to(np.asarray(max_level, np.int32), x.shape).copy()
        return x, y, r, mn, mx

    def GetFeaturesInArea(self, x, y, r, minLevel=-1, maxLevel=-1, cap=None):
        """Frame::GetFeaturesInArea for arrays of windows -> (cand_off[nq+1], cand_idx) in reference order."""
        x, y, r, mn, mx = self._windows(x, y, r, minLevel, maxLevel)
        nq = len(x)
        cap = cap or max(1, nq * max(getattr(self, "_grid_n", 0), 1))
        cap = min(cap, 1 << 22)
        off = np.zeros(nq + 1, np.int32); idx = np.zeros(cap, np.int32)
        n = self.L.orbm_features_in_area(self.h, _p(x), _p(y), _p(r), _p(mn), _p(mx), nq, _p(off), _p(idx), cap)
        if n < 0:
            _mchk(n)
        return off, idx[:n].copy()

    def search_area_best2(self, qdesc, x, y, r, minLevel, maxLevel, train_desc, skip=None):
        qdesc = np.ascontiguousarray(qdesc, np.uint8).reshape(-1, 32)
        train_desc = np.ascontiguousarray(train_desc, np.uint8).reshape(-1, 32)
        x, y, r, mn, mx = self._windows(x, y, r, minLevel, maxLevel)
        nq = len(x)
        if skip is not None:
            skip = np.ascontiguousarray(skip, np.uint8)
        bi, bd, sd = np.full(nq, -1, np.int32), np.full(nq, 256, np.int32), np.full(nq, 256, np.int32)
        _mchk(self.L.orbm_search_area_best2(self.h, _p(qdesc), _p(x), _p(y), _p(r), _p(mn), _p(mx), nq, _p(train_desc), _p(skip),
                                            _p(bi), _p(bd), _p(sd)))
        return bi, bd, sd

    def SearchByProjectionLast(self, has_point, xw, mp_desc, mp_obs, kps_last, Tcw, Tlw, K, mb, mbf, bounds, scale_factors,
                               kps_cur, desc_cur, cur_obs, th, mono, u_right=None):
        """ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) (src/ORBmatcher.cc:1328-1470);
        arguments as in include/orbm.h.  cur_obs (int32, in/out).  Returns (cur_match, nmatches)."""
        has_point = np.ascontiguousarray(has_point, np.uint8); xw = np.ascontiguousarray(xw, np.float32).reshape(-1, 3)
        mp_desc = np.ascontiguousarray(mp_desc, np.uint8).reshape(-1, 32); mp_obs = np.ascontiguousarray(mp_obs, np.int32)
        kps_last = np.ascontiguousarray(kps_last); kps_cur = np.ascontiguousarray(kps_cur)
        desc_cur = np.ascontiguousarray(desc_cur, np.uint8).reshape(-1, 32)
        Tcw = np.ascontiguousarray(Tcw, np.float32).reshape(16); Tlw = np.ascontiguousarray(Tlw, np.float32).reshape(16)
        b = np.ascontiguousarray(bounds, np.float32); sf = np.ascontiguousarray(scale_factors, np.float32)
        ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        assert cur_obs.dtype == np.int32 and cur_obs.flags["C_CONTIGUOUS"] and len(cur_obs) == len(kps_cur)
        cm = np.full(len(kps_cur), -1, np.int32)
        nm = C.c_int(0)
        fx, fy, cx, cy = K
        _mchk(self.L.orbm_search_by_projection_last(self.h, len(kps_last), _p(has_point), _p(xw), _p(mp_desc), _p(mp_obs), _p(kps_last), _p(Tcw), _p(Tlw),
                                                    fx, fy, cx, cy, mb, mbf, _p(b), _p(sf), len(sf), _p(kps_cur), _p(desc_cur), _p(ur), len(kps_cur),
                                                    th, int(mono), 1 if self.mbCheckOrientation else 0, _p(cur_obs), _p(cm), C.byref(nm)))
        return cm, nm.value

    def SearchByProjectionMap(self, in_view, proj_x, proj_y, pred_level, view_cos, mp_desc, mp_obs, scale_factors, kps_cur, desc_cur,
                              cur_obs, th, proj_xr=None, u_right=None):
        """ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th) (src/ORBmatcher.cc:45-125);
        arguments as in include/orbm.h.  Returns (cur_match, nmatches); cur_obs (int32) is updated in place."""
        f32 = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
        in_view = np.ascontiguousarray(in_view, np.uint8); px, py, pxr, vc = f32(proj_x), f32(proj_y), f32(proj_xr), f32(view_cos)
        lv = np.ascontiguousarray(pred_level, np.int32); mp_desc = np.ascontiguousarray(mp_desc, np.uint8).reshape(-1, 32)
        mp_obs = np.ascontiguousarray(mp_obs, np.int32); sf = f32(scale_factors); ur = f32(u_right)
        kps_cur = np.ascontiguousarray(kps_cur); desc_cur = np.ascontiguousarray(desc_cur, np.uint8).reshape(-1, 32)
        assert cur_obs.dtype == np.int32 and cur_obs.flags["C_CONTIGUOUS"] and len(cur_obs) == len(kps_cur)
        cm = np.full(len(kps_cur), -1, np.int32)
        nm = C.c_int(0)
        _mchk(self.L.orbm_search_by_projection_map(self.h, len(in_view), _p(in_view), _p(px), _p(py), _p(pxr), _p(lv), _p(vc), _p(mp_desc), _p(mp_obs),
                                                   _p(sf), len(sf), _p(kps_cur), _p(desc_cur), _p(ur), len(kps_cur), th, C.c_float(self.mfNNratio),
                                                   _p(cur_obs), _p(cm), C.byref(nm)))
        return cm, nm.value

    @staticmethod
    def ProjectPoints(Tcw, K, bounds, xw):
        """orbm_project_points: (u, v, invzc, dist3D, in_image) of world points under the pose Tcw (src/ORBmatcher.cc:1498-1514)."""
        L = lib()
        Tcw = np.ascontiguousarray(Tcw, np.float32).reshape(16); b = np.ascontiguousarray(bounds, np.float32)
        xw = np.ascontiguousarray(xw, np.float32).reshape(-1, 3)
        n = len(xw)
        u, v, iz, d3 = (np.zeros(n, np.float32) for _ in range(4))
        inside = np.zeros(n, np.uint8)
        fx, fy, cx, cy = K
        _mchk(L.orbm_project_points(_p(Tcw), fx, fy, cx, cy, _p(b), _p(xw), n, _p(u), _p(v), _p(iz), _p(d3), _p(inside)))
        return u, v, iz, d3, inside

    @staticmethod
    def PredictScale(mf_max_distance, dist, log_scale_factor, n_levels):
        L = lib()
        return np.array([L.orbm_predict_scale(float(a), float(b), float(log_scale_factor), int(n_levels))
                         for a, b in zip(np.asarray(mf_max_distance, np.float32), np.asarray(dist, np.float32))], np.int32)

    def SearchByProjectionKF(self, use, proj_u, proj_v, pred_level, mp_desc, kf_angle, scale_factors, kps_cur, desc_cur, cur_has_point, th, orb_dist):
        """ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, sAlreadyFound, th, ORBdist) (src/ORBmatcher.cc:1472-1599)
        after the projection step; arguments as in include/orbm.h.  cur_has_point (uint8, in/out).  Returns (cur_match, nmatches)."""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        use = np.ascontiguousarray(use, np.uint8); pu, pv, ka, sf = f32(proj_u), f32(proj_v), f32(kf_angle), f32(scale_factors)
        lv = np.ascontiguousarray(pred_level, np.int32); mp_desc = np.ascontiguousarray(mp_desc, np.uint8).reshape(-1, 32)
        kps_cur = np.ascontiguousarray(kps_cur); desc_cur = np.ascontiguousarray(desc_cur, np.uint8).reshape(-1, 32)
        assert cur_has_point.dtype == np.uint8 and cur_has_point.flags["C_CONTIGUOUS"] and len(cur_has_point) == len(kps_cur)
        cm = np.full(len(kps_cur), -1, np.int32)
        nm = C.c_int(0)
        _mchk(self.L.orbm_search_by_projection_kf(self.h, len(use), _p(use), _p(pu), _p(pv), _p(lv), _p(mp_desc), _p(ka), _p(sf), len(sf),
                                                  _p(kps_cur), _p(desc_cur), len(kps_cur), th, int(orb_dist), 1 if self.mbCheckOrientation else 0,
                                                  _p(cur_has_point), _p(cm), C.byref(nm)))
        return cm, nm.value

    # ---- the LocalMapping / LoopClosing matchers (include/orbm.h; src/ORBmatcher.cc:290-403, 522-655, 657-823, 825-975, 977-1100, 1102-1326) ----
    def reserve(self, max_queries=0, max_train=0, max_pairs=0):
        _mchk(self.L.orbm_reserve(self.h, int(max_queries), int(max_train), int(max_pairs)))

    def grid_build_kf(self, kps_un, grid):
        """KeyFrame grid: grid = (assign_min_x, assign_min_y, inv_w, inv_h, query_min_x, query_min_y)."""
        kps_un = np.ascontiguousarray(kps_un)
        _mchk(self.L.orbm_grid_build_kf(self.h, _p(kps_un), len(kps_un), *[float(v) for v in grid]))

    @staticmethod
    def Sim3Decompose(Scw):
        Scw = np.ascontiguousarray(Scw, np.float32).reshape(16)
        T, Ow = np.zeros(16, np.float32), np.zeros(3, np.float32)
        _mchk(lib().orbm_sim3_decompose(_p(Scw), _p(T), _p(Ow)))
        return T.reshape(4, 4), Ow

    @staticmethod
    def Sim3Relative(s12, R12, t12):
        R12 = np.ascontiguousarray(R12, np.float32).reshape(9); t12 = np.ascontiguousarray(t12, np.float32).reshape(3)
        sR12, sR21, t21 = np.zeros(9, np.float32), np.zeros(9, np.float32), np.zeros(3, np.float32)
        _mchk(lib().orbm_sim3_relative(C.c_float(s12), _p(R12), _p(t12), _p(sR12), _p(sR21), _p(t21)))
        return sR12.reshape(3, 3), sR21.reshape(3, 3), t21

    @staticmethod
    def ProjectPointsKF(Tcw, K, bounds, xw, normal=None, Ow=None):
        """orbm_project_points_kf: (u, v, invz, dist3D, ok)."""
        Tcw = np.ascontiguousarray(Tcw, np.float32).reshape(16); b = np.ascontiguousarray(bounds, np.float32)
        xw = np.ascontiguousarray(xw, np.float32).reshape(-1, 3)
        if normal is not None:
            normal = np.ascontiguousarray(normal, np.float32).reshape(-1, 3)
        if Ow is not None:
            Ow = np.ascontiguousarray(Ow, np.float32).reshape(3)
        n = len(xw)
        u, v, iz, d3 = (np.zeros(n, np.float32) for _ in range(4))
        ok = np.zeros(n, np.uint8)
        fx, fy, cx, cy = K
        _mchk(lib().orbm_project_points_kf(_p(Tcw), _p(Ow), fx, fy, cx, cy, _p(b), _p(xw), _p(normal), n, _p(u), _p(v), _p(iz), _p(d3), _p(ok)))
        return u, v, iz, d3, ok

    @staticmethod
    def ProjectPointsSim3(TAw, sR, t, K, boundsB, xw):
        TAw = np.ascontiguousarray(TAw, np.float32).reshape(16); sR = np.ascontiguousarray(sR, np.float32).reshape(9)
        t = np.ascontiguousarray(t, np.float32).reshape(3); b = np.ascontiguousarray(boundsB, np.float32)
        xw = np.ascontiguousarray(xw, np.float32).reshape(-1, 3)
        n = len(xw)
        u, v, d3 = (np.zeros(n, np.float32) for _ in range(3))
        ok = np.zeros(n, np.uint8)
        fx, fy, cx, cy = K
        _mchk(lib().orbm_project_points_sim3(_p(TAw), _p(sR), _p(t), fx, fy, cx, cy, _p(b), _p(xw), n, _p(u), _p(v), _p(d3), _p(ok)))
        return u, v, d3, ok

    def SearchByProjectionSim3(self, use, proj_u, proj_v, pred_level, mp_desc, scale_factors, kps_kf, desc_kf, kf_matched, th):
        """SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) after the projection step.  kf_matched uint8 in/out.
        Returns (kf_match, nmatches)."""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        use = np.ascontiguousarray(use, np.uint8); pu, pv, sf = f32(proj_u), f32(proj_v), f32(scale_factors)
        lv = np.ascontiguousarray(pred_level, np.int32); mp_desc = np.ascontiguousarray(mp_desc, np.uint8).reshape(-1, 32)
        kps_kf = np.ascontiguousarray(kps_kf); desc_kf = np.ascontiguousarray(desc_kf, np.uint8).reshape(-1, 32)
        assert kf_matched.dtype == np.uint8 and kf_matched.flags["C_CONTIGUOUS"] and len(kf_matched) == len(kps_kf)
        km = np.full(len(kps_kf), -1, np.int32)
        nm = C.c_int(0)
        _mchk(self.L.orbm_search_by_projection_sim3(self.h, len(use), _p(use), _p(pu), _p(pv), _p(lv), _p(mp_desc), _p(sf), len(sf), _p(kps_kf),
                                                    _p(desc_kf), len(kps_kf), int(th), _p(kf_matched), _p(km), C.byref(nm)))
        return km, nm.value

    def SearchByBoWKF(self, kps1, desc1, featvec1, valid1, kps2, desc2, featvec2, valid2):
        """SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12): returns (matches12, nmatches)."""
        desc1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32); desc2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
        kps1 = np.ascontiguousarray(kps1); kps2 = np.ascontiguousarray(kps2)
        n1, o1, i1 = [np.ascontiguousarray(a, np.int32) for a in featvec1]
        n2, o2, i2 = [np.ascontiguousarray(a, np.int32) for a in featvec2]
        valid1 = np.ascontiguousarray(valid1, np.uint8); valid2 = np.ascontiguousarray(valid2, np.uint8)
        m12 = np.full(len(desc1), -1, np.int32)
        nm = C.c_int(0)
        _mchk(self.L.orbm_search_by_bow_kf(self.h, _p(desc1), _p(kps1), len(desc1), _p(valid1), _p(n1), _p(o1), _p(i1), len(n1),
                                           _p(desc2), _p(kps2), len(desc2), _p(valid2), _p(n2), _p(o2), _p(i2), len(n2),
                                           C.c_float(self.mfNNratio), 1 if self.mbCheckOrientation else 0, _p(m12), C.byref(nm)))
        return m12, nm.value

    def SearchForTriangulation(self, kps1, desc1, has_mp1, u_right1, featvec1, kps2, desc2, has_mp2, u_right2, featvec2,
                               Cw, T2w, K2, F12, scale_factors2, level_sigma2_2, only_stereo=False):
        """SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo): returns (matches12, nmatches)."""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        desc1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32); desc2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
        kps1 = np.ascontiguousarray(kps1); kps2 = np.ascontiguousarray(kps2)
        n1, o1, i1 = [np.ascontiguousarray(a, np.int32) for a in featvec1]
        n2, o2, i2 = [np.ascontiguousarray(a, np.int32) for a in featvec2]
        h1 = np.ascontiguousarray(has_mp1, np.uint8); h2 = np.ascontiguousarray(has_mp2, np.uint8)
        ur1, ur2, Cw, T2w, F12, sf2, ls2 = f32(u_right1), f32(u_right2), f32(Cw).reshape(3), f32(T2w).reshape(16), f32(F12).reshape(9), f32(scale_factors2), f32(level_sigma2_2)
        m12 = np.full(len(desc1), -1, np.int32)
        nm = C.c_int(0)
        fx, fy, cx, cy = K2
        _mchk(self.L.orbm_search_for_triangulation(self.h, _p(kps1), _p(desc1), len(desc1), _p(h1), _p(ur1), _p(n1), _p(o1), _p(i1), len(n1),
                                                   _p(kps2), _p(desc2), len(desc2), _p(h2), _p(ur2), _p(n2), _p(o2), _p(i2), len(n2),
                                                   _p(Cw), _p(T2w), fx, fy, cx, cy, _p(F12), _p(sf2), _p(ls2), len(sf2), 1 if only_stereo else 0,
                                                   1 if self.mbCheckOrientation else 0, _p(m12), C.byref(nm)))
        return m12, nm.value

    def Fuse(self, use, proj_u, proj_v, proj_ur, pred_level, mp_desc, scale_factors, inv_level_sigma2, kps_kf, u_right_kf, desc_kf, th=3.0):
        """Fuse(KeyFrame*, vpMapPoints, th), the search half: returns (best_idx, nfused)."""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        use = np.ascontiguousarray(use, np.uint8); pu, pv, pr, sf, inv, urk = f32(proj_u), f32(proj_v), f32(proj_ur), f32(scale_factors), f32(inv_level_sigma2), f32(u_right_kf)
        lv = np.ascontiguousarray(pred_level, np.int32); mp_desc = np.ascontiguousarray(mp_desc, np.uint8).reshape(-1, 32)
        kps_kf = np.ascontiguousarray(kps_kf); desc_kf = np.ascontiguousarray(desc_kf, np.uint8).reshape(-1, 32)
        bi = np.full(len(use), -1, np.int32)
        nf = C.c_int(0)
        _mchk(self.L.orbm_fuse(self.h, len(use), _p(use), _p(pu), _p(pv), _p(pr), _p(lv), _p(mp_desc), _p(sf), _p(inv), len(sf), _p(kps_kf), _p(urk),
                               _p(desc_kf), len(kps_kf), C.c_float(th), _p(bi), C.byref(nf)))
        return bi, nf.value

    def FuseSim3(self, use, proj_u, proj_v, pred_level, mp_desc, scale_factors, kps_kf, desc_kf, th):
        """Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint), the search half: returns (best_idx, nfused)."""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        use = np.ascontiguousarray(use, np.uint8); pu, pv, sf = f32(proj_u), f32(proj_v), f32(scale_factors)
        lv = np.ascontiguousarray(pred_level, np.int32); mp_desc = np.ascontiguousarray(mp_desc, np.uint8).reshape(-1, 32)
        kps_kf = np.ascontiguousarray(kps_kf); desc_kf = np.ascontiguousarray(desc_kf, np.uint8).reshape(-1, 32)
        bi = np.full(len(use), -1, np.int32)
        nf = C.c_int(0)
        _mchk(self.L.orbm_fuse_sim3(self.h, len(use), _p(use), _p(pu), _p(pv), _p(lv), _p(mp_desc), _p(sf), len(sf), _p(kps_kf), _p(desc_kf), len(kps_kf),
                                    C.c_float(th), _p(bi), C.byref(nf)))
        return bi, nf.value

    def SearchBySim3(self, use1, u1, v1, lv1, mp_desc1, use2, u2, v2, lv2, mp_desc2, kps1, desc1, grid1, sf1, kps2, desc2, grid2, sf2, th):
        """SearchBySim3, the two searches and the agreement check: returns (match12, nfound).  grid1 / grid2 as for grid_build_kf."""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        u8 = lambda a: np.ascontiguousarray(a, np.uint8)
        i32 = lambda a: np.ascontiguousarray(a, np.int32)
        use1, use2, u1, v1, u2, v2, lv1, lv2, sf1, sf2 = u8(use1), u8(use2), f32(u1), f32(v1), f32(u2), f32(v2), i32(lv1), i32(lv2), f32(sf1), f32(sf2)
        mp_desc1 = u8(mp_desc1).reshape(-1, 32); mp_desc2 = u8(mp_desc2).reshape(-1, 32)
        kps1 = np.ascontiguousarray(kps1); kps2 = np.ascontiguousarray(kps2); desc1 = u8(desc1).reshape(-1, 32); desc2 = u8(desc2).reshape(-1, 32)
        g1, g2 = f32(grid1), f32(grid2)
        m12 = np.full(len(use1), -1, np.int32)
        nf = C.c_int(0)
        _mchk(self.L.orbm_search_by_sim3(self.h, len(use1), _p(use1), _p(u1), _p(v1), _p(lv1), _p(mp_desc1), len(use2), _p(use2), _p(u2), _p(v2), _p(lv2),
                                         _p(mp_desc2), _p(kps1), _p(desc1), len(kps1), _p(g1), _p(sf1), len(sf1), _p(kps2), _p(desc2), len(kps2), _p(g2),
                                         _p(sf2), len(sf2), C.c_float(th), _p(m12), C.byref(nf)))
        return m12, nf.value

    def SearchForInitialization(self, kps1, desc1, kps2, desc2, prev_matched, windowSize=10):
        """ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (src/ORBmatcher.cc:405-520).
        The grid of frame 2 must be current (grid_build(kps2, ...)).  prev_matched (n1, 2) float32 is updated in place;
        returns (vnMatches12, nmatches)."""
        kps1 = np.ascontiguousarray(kps1); kps2 = np.ascontiguousarray(kps2)
        desc1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32); desc2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
        assert prev_matched.dtype == np.float32 and prev_matched.flags["C_CONTIGUOUS"] and prev_matched.shape == (len(kps1), 2)
        m12 = np.full(len(kps1), -1, np.int32)
        nm = C.c_int(0)
        _mchk(self.L.orbm_search_for_initialization(self.h, _p(kps1), _p(desc1), len(kps1), _p(kps2), _p(desc2), len(kps2), _p(prev_matched),
                                                    int(windowSize), C.c_float(self.mfNNratio), 1 if self.mbCheckOrientation else 0, _p(m12), C.byref(nm)))
        return m12, nm.value

    def SearchByBoW(self, kps_kf, desc_kf, featvec_kf, kps_f, desc_f, featvec_f, valid_kf=None):
        """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (src/ORBmatcher.cc:159-288).  featvec_* are the
        (node ids, offsets, indices) triples of ORBVocabulary.transform; returns (match_f, nmatches) with match_f[i] =
        key-frame feature matched to frame feature i or -1."""
        desc_kf = np.ascontiguousarray(desc_kf, np.uint8).reshape(-1, 32)
        desc_f = np.ascontiguousarray(desc_f, np.uint8).reshape(-1, 32)
        kps_kf = np.ascontiguousarray(kps_kf); kps_f = np.ascontiguousarray(kps_f)
        nk, ok, ik = [np.ascontiguousarray(a, np.int32) for a in featvec_kf]
        nf, of, i_f = [np.ascontiguousarray(a, np.int32) for a in featvec_f]
        if valid_kf is not None:
            valid_kf = np.ascontiguousarray(valid_kf, np.uint8)
        match_f = np.full(len(desc_f), -1, np.int32)
        nm = C.c_int(0)
        _mchk(self.L.orbm_search_by_bow(self.h, _p(desc_kf), _p(kps_kf), len(desc_kf), _p(valid_kf), _p(nk), _p(ok), _p(ik), len(nk),
                                        _p(desc_f), _p(kps_f), len(desc_f), _p(nf), _p(of), _p(i_f), len(nf),
                                        C.c_float(self.mfNNratio), 1 if self.mbCheckOrientation else 0, _p(match_f), C.byref(nm)))
        return match_f, nm.value


class ORBVocabulary:
    """DBoW2 TemplatedVocabulary<FORB> (text format) with the tree descent on the GPU (include/orbv.h)."""

    def __init__(self, path, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        rc = self.L.orbv_load_text(C.byref(self.h), path.encode(), device)
        if rc != ORBX_OK:
            raise OrbxError(rc, self.L.orbv_last_error().decode())
        v = [C.c_int() for _ in range(6)]
        self.L.orbv_info(self.h, *[C.byref(x) for x in v])
        self.k, self.depth, self.nnodes, self.nwords, self.scoring, self.weighting = [x.value for x in v]

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.orbv_destroy(self.h)
            self.h = None

    __del__ = close

    def transform_features(self, desc, levelsup=4):
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(desc)
        w = np.zeros(n, np.int32); nd = np.zeros(n, np.int32); wt = np.zeros(n, np.float64)
        rc = self.L.orbv_transform_features(self.h, _p(desc), n, levelsup, _p(w), _p(nd), _p(wt))
        if rc != ORBX_OK:
            raise OrbxError(rc, self.L.orbv_last_error().decode())
        return w, nd, wt

    def transform(self, desc, levelsup=4):
        """transform(features, BowVector, FeatureVector, levelsup) -> ((ids, vals), (node_ids, off, idx))."""
        w, nd, wt = self.transform_features(desc, levelsup)
        n = len(w)
        ids = np.zeros(max(n, 1), np.int32); vals = np.zeros(max(n, 1), np.float64)
        nb = self.L.orbv_bow_vector(self.h, _p(w), _p(wt), n, _p(ids), _p(vals), len(ids))
        nids = np.zeros(max(n, 1), np.int32); off = np.zeros(n + 1, np.int32); idx = np.zeros(max(n, 1), np.int32)
        nn = self.L.orbv_feature_vector(_p(nd), _p(wt), n, _p(nids), _p(off), _p(idx), len(nids))
        if nb < 0 or nn < 0:
            raise OrbxError(min(nb, nn), self.L.orbv_last_error().decode())
        return (ids[:nb].copy(), vals[:nb].copy()), (nids[:nn].copy(), off[:nn + 1].copy(), idx[:off[nn]].copy())

    def score(self, bow1, bow2):
        return self.L.orbv_score_l1(_p(bow1[0]), _p(bow1[1]), len(bow1[0]), _p(bow2[0]), _p(bow2[1]), len(bow2[0]))


def UndistortKeyPoints(kps, fx, fy, cx, cy, dist_coef):
    """Frame::UndistortKeyPoints (src/Frame.cc:404-434): mvKeys -> mvKeysUn for mDistCoef = (k1, k2, p1, p2[, k3])."""
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    d = np.ascontiguousarray(dist_coef, np.float32)
    out = np.empty_like(kps)
    _mchk(lib().orbm_undistort_keypoints(_p(kps), len(kps), fx, fy, cx, cy, _p(d), len(d), _p(out)))
    return out


def ComputeImageBounds(width, height, fx, fy, cx, cy, dist_coef):
    """Frame::ComputeImageBounds (src/Frame.cc:436-463) -> (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
    d = np.ascontiguousarray(dist_coef, np.float32)
    b = np.zeros(4, np.float32)
    _mchk(lib().orbm_image_bounds(width, height, fx, fy, cx, cy, _p(d), len(d), _p(b)))
    return tuple(float(v) for v in b)


def _pchk(rc):
    if rc < 0:
        raise OrbxError(rc, lib().orbp_last_error().decode())
    return rc


class PnPsolver:
    """Mirror of ORB_SLAM2::PnPsolver (include/PnPsolver.h:66-78, src/PnPsolver.cc:66-344) over orbp_pnp_*: host-side
    EPnP RANSAC.  The Frame / MapPoint arguments of the reference constructor become the arrays it reads from them:
    p2d = mvKeysUn[i].pt, sigma2 = mvLevelSigma2[octave], p3d = GetWorldPos() of the non-bad matches."""

    def __init__(self, p2d, sigma2, p3d, fx, fy, cx, cy, rand=None, rand_max=None):
        self.L = lib()
        self.p2d = np.ascontiguousarray(p2d, np.float32).reshape(-1, 2)
        self.sigma2 = np.ascontiguousarray(sigma2, np.float32)
        self.p3d = np.ascontiguousarray(p3d, np.float32).reshape(-1, 3)
        self.N = len(self.p2d)
        self.h = C.c_void_p()
        _pchk(self.L.orbp_pnp_create(C.byref(self.h), self.N, _p(self.p2d), _p(self.sigma2), _p(self.p3d), fx, fy, cx, cy))
        self._cb = None
        if rand is not None:
            self._cb = RAND_FN(lambda ctx: int(rand()))
            self.L.orbp_pnp_set_rand(self.h, self._cb, None, int(rand_max))

    def close(self):
        if self.h:
            self.L.orbp_pnp_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def SetRansacParameters(self, probability=0.99, minInliers=8, maxIterations=300, minSet=4, epsilon=0.4, th2=5.991):
        _pchk(self.L.orbp_pnp_set_ransac_parameters(self.h, probability, minInliers, maxIterations, minSet, epsilon, th2))

    def ransac_state(self):
        mi, mx, it, ep = C.c_int(), C.c_int(), C.c_int(), C.c_float()
        self.L.orbp_pnp_get_ransac_state(self.h, C.byref(mi), C.byref(mx), C.byref(ep), C.byref(it))
        return {"min_inliers": mi.value, "max_its": mx.value, "epsilon": ep.value, "iterations": it.value}

    def iterate(self, nIterations):
        """-> (Tcw 4x4 float32 or None, bNoMore, vbInliers (bool[N]) or None, nInliers)"""
        no_more, ninl = C.c_int(), C.c_int()
        inl = np.zeros(max(self.N, 1), np.uint8)
        T = np.zeros(16, np.float32)
        got = _pchk(self.L.orbp_pnp_iterate(self.h, nIterations, C.byref(no_more), _p(inl), C.byref(ninl), _p(T)))
        if not got:
            return None, bool(no_more.value), None, 0
        return T.reshape(4, 4), bool(no_more.value), inl[:self.N].astype(bool), ninl.value

    def find(self):
        st = self.ransac_state()
        T, _, inl, n = self.iterate(st["max_its"])
        return T, inl, n


def epnp(pws, us, fu, fv, uc, vc):
    """PnPsolver::compute_pose (src/PnPsolver.cc:458-508) on all given correspondences -> (R, t, mean reprojection error)."""
    pws = np.ascontiguousarray(pws, np.float64).reshape(-1, 3)
    us = np.ascontiguousarray(us, np.float64).reshape(-1, 2)
    R, t = np.zeros((3, 3)), np.zeros(3)
    err = lib().orbp_epnp(len(pws), _p(pws), _p(us), fu, fv, uc, vc, _p(R), _p(t))
    if err < 0:
        raise OrbxError(ORBX_E_INVALID, lib().orbp_last_error().decode())
    return R, t, err


def PoseOptimization(obs, inv_sigma2, xw, fx, fy, cx, cy, Tcw, u_right=None, bf=0.0):
    """Optimizer::PoseOptimization (src/Optimizer.cc:239-451) -> (Tcw 4x4 float32, mvbOutlier bool[n], n_inliers)."""
    obs = np.ascontiguousarray(obs, np.float32).reshape(-1, 2)
    inv_sigma2 = np.ascontiguousarray(inv_sigma2, np.float32)
    xw = np.ascontiguousarray(xw, np.float32).reshape(-1, 3)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    T = np.ascontiguousarray(Tcw, np.float32).reshape(16).copy()
    out = np.zeros(max(len(obs), 1), np.uint8)
    n = _pchk(lib().orbp_pose_optimization(len(obs), _p(obs), _p(ur), _p(inv_sigma2), _p(xw), fx, fy, cx, cy, bf, _p(T), _p(out)))
    return T.reshape(4, 4), out[:len(obs)].astype(bool), n
