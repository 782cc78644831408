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


@pytest.mark.parametrize("W,H,n,seed,what", [
    (1024, 768, 3000, 31, "512-thread quadtree, > 6144 level-0 candidates: keys in global memory"),
    (1920, 1080, 600, 32, "1024-thread quadtree, few features: tiny lists, phase B early"),
    (1600, 1200, 6000, 33, "1024-thread quadtree, 32 keys per thread"),
    (200, 150, 3000, 34, "more features requested than candidates exist"),
])
def test_quadtree_key_storage_tiers(orbx, synth, W, H, n, seed, what):
    """k_octree keeps the keys in registers (8 / 12 / 32 per thread) or, beyond that, in global memory."""
    img = synth.texture(seed, W, H)
    ex = orbx.ORBextractor(n, 1.2, 8, 20, 7, max_width=W, max_height=H)
    kps, desc = ex(img)
    okps, odesc, _ = O.Extractor(n).extract(img)
    _compare(kps, desc, okps, odesc, what)
    if W == 1024:
        assert len(ex.candidates(0, 0)) > 6144, "case no longer reaches the global-memory key path"


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


@pytest.mark.parametrize("cfg,W,H", [("2,32,32,1", 640, 480), ("1,24,24,1", 1241, 376), ("3,40,16,1", 322, 241)])
def test_pyramid_tile_kernel_switch(orbx, synth, monkeypatch, cfg, W, H):
    """ORBX_PYRAMID_TILES: the upper levels in one launch, one wave per 2-D tile (k_resize_tiles; off by default, it measured slower):
    every level byte-equal to the oracle's pyramid whatever the first fused level and the tile shape."""
    monkeypatch.setenv("ORBX_PYRAMID_TILES", cfg)
    img = synth.texture(11, W, H)
    ex = orbx.ORBextractor(1000, max_width=W, max_height=H)
    ex(img)
    opyr = O.Extractor(1000).pyramid(img)
    pyr = ex.image_pyramid()
    for l in range(8):
        assert np.array_equal(pyr[l], opyr[l]), "level %d differs (tiles %s)" % (l, cfg)


def test_batch_equals_single(orbx, synth):
    frames = synth.stream(4, 640, 480, 4)
    ex = orbx.ORBextractor(1000, max_width=640, max_height=480, max_batch=4)
    res = ex.extract_batch(frames)
    oex = O.Extractor(1000)
    for k in range(4):
        okps, odesc, _ = oex.extract(frames[k])
        _compare(res[k][0], res[k][1], okps, odesc, "frame %d" % k)


def test_batch_chunked_pipeline_equals_plain(orbx, synth):
    """orbx_extract_batch cuts a host batch into chunks (staging threads, two streams, one HIP graph per chunk); the result must be
    the plain one-piece path's, for a ragged last chunk, a padded row pitch and repeated calls (graph replay)."""
    W, H, B = 322, 241, 21
    pad = np.zeros((B, H, W + 14), np.uint8)
    pad[:, :, :W] = synth.stream(9, W, H, B)
    frames = pad[:, :, :W]                                   # row stride 336 != width
    assert frames.strides[1] == W + 14
    ex = orbx.ORBextractor(500, max_width=W, max_height=H, max_batch=B)
    ex.set_batch_chunk(0)
    plain = ex.extract_batch_raw(frames)
    plain = tuple(a.copy() for a in plain)
    for chunk in (4, 8, 10):
        ex.set_batch_chunk(chunk)
        for rep in range(3):                                 # capture, replay, replay
            k, d, c = ex.extract_batch_raw(frames)
            assert np.array_equal(c, plain[2]), (chunk, rep)
            for f in range(B):
                n = int(c[f])
                assert k[f, :n].tobytes() == plain[0][f, :n].tobytes() and np.array_equal(d[f, :n], plain[1][f, :n]), (chunk, rep, f)
    okps, odesc, _ = O.Extractor(500).extract(np.ascontiguousarray(frames[B - 1]))
    n = int(plain[2][B - 1])
    _compare(plain[0][B - 1, :n], plain[1][B - 1, :n], okps, odesc, "last frame")


def test_batch_page_locked_input_equals_pageable(orbx, synth):
    """orbx_extract_batch uploads page-locked caller memory where it lies (no staging copy): same result as from pageable memory, for a
    contiguous batch (one tall 2-D copy per chunk) and for a padded row pitch (re-pitched by the copy)."""
    import torch
    W, H, B = 322, 241, 18
    for padw in (0, 14):
        pad = np.zeros((B, H, W + padw), np.uint8)
        pad[:, :, :W] = synth.stream(11, W, H, B)
        frames = pad[:, :, :W]
        ex = orbx.ORBextractor(500, max_width=W, max_height=H, max_batch=B)
        ex.set_batch_chunk(4)
        ref = tuple(a.copy() for a in ex.extract_batch_raw(frames))
        pinned_full = torch.from_numpy(pad).pin_memory().numpy()
        pinned = pinned_full[:, :, :W]
        assert pinned.strides == frames.strides
        for rep in range(3):
            k, d, c = ex.extract_batch_raw(pinned)
            assert np.array_equal(c, ref[2]), (padw, rep)
            for f in range(B):
                n = int(c[f])
                assert k[f, :n].tobytes() == ref[0][f, :n].tobytes() and np.array_equal(d[f, :n], ref[1][f, :n]), (padw, rep, f)


# ---- committed fixtures: no oracle build needed for these ----
import glob
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GCASES = sorted(p for p in glob.glob(os.path.join(GOLDEN, "*.npz")) if not os.path.basename(p).startswith("match"))


@pytest.mark.parametrize("path", GCASES, ids=[os.path.basename(p)[:-4] for p in GCASES])
def test_hip_reproduces_golden(orbx, synth, path):
    g = np.load(path)
    W, H = int(g["W"]), int(g["H"])
    img = synth.texture(int(g["seed"]), W, H)
    ex = orbx.ORBextractor(int(g["nfeatures"]), max_width=W, max_height=H)
    ex.set_blur_rounding(int(g["blur_mode"]))
    kps, desc = ex(img)
    assert kps.tobytes() == g["keypoints"].tobytes()
    assert np.array_equal(desc, g["descriptors"])


def test_config3_1080p_n4000(orbx, synth):
    """BASELINE configs[2]: 1920x1080, nFeatures=4000, extract both frames + dense match vs previous."""
    f0, f1 = synth.frame_pair(2, 1920, 1080)
    ex = orbx.ORBextractor(4000, max_width=1920, max_height=1080)
    oex = O.Extractor(4000)
    k0, d0 = ex(f0)
    ok0, od0, _ = oex.extract(f0)
    _compare(k0, d0, ok0, od0, "1080p f0")
    k1, d1 = ex(f1)
    ok1, od1, _ = oex.extract(f1)
    _compare(k1, d1, ok1, od1, "1080p f1")
    m = orbx.ORBmatcher(0.9, True)
    nm, m12 = m.match_dense(d1, k1, d0, k0)
    onm, om12 = O.match_dense(od1, ok1["angle"], od0, ok0["angle"], 50, 0.9, True)
    assert nm == onm and np.array_equal(m12, om12) and nm > 1000


def test_config4_batch64_stream(orbx, synth):
    """BASELINE configs[3] on one GPU: 64 x 640x480 stream in one call; every frame == oracle on a
    sample, and the call is idempotent (size-independent property for the rest)."""
    frames = synth.stream(4, 640, 480, 64)
    ex = orbx.ORBextractor(1000, max_width=640, max_height=480, max_batch=64)
    r1 = ex.extract_batch(frames)
    r2 = ex.extract_batch(frames)
    oex = O.Extractor(1000)
    for k in range(64):
        assert r1[k][0].tobytes() == r2[k][0].tobytes() and np.array_equal(r1[k][1], r2[k][1])
        assert 900 <= len(r1[k][0]) <= ex.cap
    for k in (0, 17, 63):
        okps, odesc, _ = oex.extract(frames[k])
        _compare(r1[k][0], r1[k][1], okps, odesc, "frame %d" % k)


@pytest.mark.parametrize("params", [
    dict(nfeatures=50, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7),
    dict(nfeatures=700, scaleFactor=1.5, nlevels=4, iniThFAST=30, minThFAST=10),
    dict(nfeatures=300, scaleFactor=2.0, nlevels=3, iniThFAST=20, minThFAST=7),      # exact 2x: INTER_AREA path
    dict(nfeatures=2000, scaleFactor=1.2, nlevels=8, iniThFAST=7, minThFAST=7),
    dict(nfeatures=500, scaleFactor=1.1, nlevels=12, iniThFAST=12, minThFAST=20),   # ini < min
    dict(nfeatures=1, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7),       # quotas of 0
])
def test_parameter_sweep(orbx, synth, params):
    W, H = 512, 384
    img = synth.texture(21, W, H)
    ex = orbx.ORBextractor(max_width=W, max_height=H, **params)
    kps, desc = ex(img)
    okps, odesc, _ = O.Extractor(params["nfeatures"], params["scaleFactor"], params["nlevels"],
                                 params["iniThFAST"], params["minThFAST"]).extract(img)
    _compare(kps, desc, okps, odesc, str(params))
    assert np.array_equal(ex.GetScaleFactors(), np.array(list(O.Extractor(1, params["scaleFactor"], params["nlevels"]).e.scale)[:params["nlevels"]], np.float32))


def test_edge_inputs(orbx, synth):
    ex = orbx.ORBextractor(500, max_width=640, max_height=480)
    # empty image: the reference returns silently (src/ORBextractor.cc:1048)
    k, d = ex(np.zeros((0, 0), np.uint8))
    assert len(k) == 0 and d.shape == (0, 32)
    # flat image: no corner anywhere
    k, d = ex(np.full((480, 640), 93, np.uint8))
    assert len(k) == 0
    # row stride > width (a cv::Mat ROI)
    big = synth.texture(5, 700, 480)
    roi = big[:, 30:670]
    assert roi.strides[0] == 700
    k, d = ex(roi)
    ok, od, _ = O.Extractor(500).extract(np.ascontiguousarray(roi))
    _compare(k, d, ok, od, "roi")
    # a different (smaller) shape on the same handle re-plans; tiny levels have no cells
    small = synth.texture(6, 150, 120)
    k, d = ex(small)
    ok, od, _ = O.Extractor(500).extract(small)
    _compare(k, d, ok, od, "small")
    # and back
    k, d = ex(np.ascontiguousarray(roi))
    _compare(k, d, ok if False else O.Extractor(500).extract(np.ascontiguousarray(roi))[0], od if False else O.Extractor(500).extract(np.ascontiguousarray(roi))[1], "back")
    # larger than the handle's maximum / portrait shapes the reference cannot process
    with pytest.raises(orbx.OrbxError) as ei:
        ex(np.zeros((481, 640), np.uint8))
    assert ei.value.code == orbx.ORBX_E_SHAPE
    with pytest.raises(orbx.OrbxError) as ei:
        orbx.ORBextractor(500, max_width=200, max_height=480)
    assert ei.value.code == orbx.ORBX_E_SHAPE
    # caller capacity too small
    kp = np.zeros(10, orbx.KP_DTYPE); de = np.zeros((10, 32), np.uint8)
    import ctypes as C
    n = C.c_int()
    img = synth.texture(5, 640, 480)
    rc = ex.L.orbx_extract(ex.h, img.ctypes.data, 640, 480, 640, kp.ctypes.data, de.ctypes.data, 10, C.byref(n))
    assert rc == orbx.ORBX_E_CAPACITY


def test_saturated_and_noisy_images(orbx):
    """Extreme inputs: white frame with dark dots (blur saturation at 255), uniform noise and a 2-px checkerboard of
    isolated bright pixels (the densest candidate fields there are).  The candidate buffers hold the provable maximum of
    NMS survivors (strict 8-neighbour maxima: at most every other pixel of every other row of a cell), so these must equal
    the oracle like any other image -- the reference has no "too many corners" failure."""
    rng = np.random.default_rng(3)
    img = np.full((240, 320), 255, np.uint8)
    ys, xs = rng.integers(25, 215, 300), rng.integers(25, 295, 300)
    img[ys, xs] = 0
    ex = orbx.ORBextractor(500, max_width=320, max_height=240)
    k, d = ex(img)
    ok, od, _ = O.Extractor(500).extract(img)
    _compare(k, d, ok, od, "dots")
    noise = rng.integers(0, 256, (240, 320), dtype=np.uint8)
    k, d = ex(noise)
    ok, od, _ = O.Extractor(500).extract(noise)
    _compare(k, d, ok, od, "noise")
    lattice = np.zeros((240, 320), np.uint8)
    lattice[::2, ::2] = rng.integers(60, 256, (120, 160), dtype=np.uint8)     # every lit pixel is a corner and a local maximum
    k, d = ex(lattice)
    ok, od, _ = O.Extractor(500).extract(lattice)
    _compare(k, d, ok, od, "lattice")
    big = rng.integers(0, 256, (480, 640), dtype=np.uint8)                    # noise at the headline shape, batched path too
    ex2 = orbx.ORBextractor(1000, max_width=640, max_height=480, max_batch=2)
    ok, od, _ = O.Extractor(1000).extract(big)
    for kk, dd in ex2.extract_batch(np.stack([big, big])):
        _compare(kk, dd, ok, od, "noise 640x480")


def test_pyramid_border_download(orbx, synth):
    img = synth.texture(8, 320, 240)
    ex = orbx.ORBextractor(300, max_width=320, max_height=240)
    ex(img)
    pyr = ex.image_pyramid(border=19)
    opyr = O.Extractor(300).pyramid(img)
    for l in range(8):
        assert np.array_equal(pyr[l], np.pad(opyr[l], 19, mode="reflect"))     # copyMakeBorder REFLECT_101
    allp = ex.image_pyramid_all(border=19)                                      # one call, one synchronisation (the adapter's mvImagePyramid)
    assert all(np.array_equal(a, b) for a, b in zip(allp, pyr))
    assert all(np.array_equal(a, b) for a, b in zip(ex.image_pyramid_all(border=0), opyr))
    ex2 = orbx.ORBextractor(300, max_width=320, max_height=240, max_batch=2)
    ex2.extract_batch(np.stack([img, img[::-1].copy()]))
    flipped = ex2.image_pyramid_all(frame=1, border=19)
    assert np.array_equal(flipped[0][19:-19, 19:-19], img[::-1]) and not np.array_equal(flipped[1], pyr[1])


def test_ring_profiling_mode(orbx, synth):
    """orbx_set_profiling(2): stage events dropped into the caller's stream, read back after the caller synchronises."""
    import torch
    W, H, B = 320, 240, 2
    fr = torch.from_numpy(synth.stream(3, W, H, B)).cuda()
    ex = orbx.ORBextractor(300, max_width=W, max_height=H, max_batch=B)
    cap = ex.cap
    k = torch.zeros((B, cap, 7), device="cuda"); d = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    c = torch.zeros(B, dtype=torch.int32, device="cuda"); s = torch.zeros(B, dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    ex.set_profiling(2)
    for _ in range(20):                       # more calls than the ring holds
        ex.extract_batch_device(fr.data_ptr(), B, W, H, fr.stride(1), fr.stride(0), k.data_ptr(), d.data_ptr(), c.data_ptr(), s.data_ptr(), st.cuda_stream)
    st.synchronize()
    ms = ex.stage_ms_ring(16)
    assert ms.shape == (16, 4) and (ms > 0).all() and (ms < 50).all()
    ex.set_profiling(0)
    ref = ex(synth.stream(3, W, H, B)[0])
    assert int(c[0]) == len(ref[0])


def test_extract_begin_end_pipelined(orbx, synth):
    """orbx_extract_begin / orbx_extract_end: two handles with a frame in flight on each give what the blocking call gives;
    misuse (second begin, end without begin, blocking call while in flight) is an error."""
    W, H = 640, 480
    frames = synth.stream(9, W, H, 6)
    ref = orbx.ORBextractor(1000, max_width=W, max_height=H)
    want = [ref(frames[k]) for k in range(6)]               # calls 2.. of the handle replay its HIP graph
    oex = O.Extractor(1000)
    for k in (0, 1, 5):
        okps, odesc, _ = oex.extract(frames[k])
        assert want[k][0].tobytes() == okps.tobytes() and np.array_equal(want[k][1], odesc)
    ex = [orbx.ORBextractor(1000, max_width=W, max_height=H), orbx.ORBextractor(1000, max_width=W, max_height=H)]
    ex[0].extract_begin(frames[0])
    for k in range(6):
        if k + 1 < 6:
            ex[(k + 1) & 1].extract_begin(frames[k + 1])
        kp, de = ex[k & 1].extract_end()
        assert kp.tobytes() == want[k][0].tobytes() and np.array_equal(de, want[k][1])
    ex[0].extract_begin(frames[0])
    with pytest.raises(orbx.OrbxError):
        ex[0].extract_begin(frames[1])
    with pytest.raises(orbx.OrbxError):
        ex[0](frames[1])
    import torch                                # the device-buffer entry point would overwrite the pending call's workspace too
    fr = torch.from_numpy(frames[:1]).cuda()
    cap = ex[0].cap
    dk = torch.zeros((1, cap, 7), dtype=torch.float32, device="cuda"); dd = torch.zeros((1, cap, 32), dtype=torch.uint8, device="cuda")
    dc = torch.zeros(1, dtype=torch.int32, device="cuda"); ds = torch.zeros(1, dtype=torch.int32, device="cuda")
    with pytest.raises(orbx.OrbxError):
        ex[0].extract_batch_device(fr.data_ptr(), 1, W, H, fr.stride(1), fr.stride(0), dk.data_ptr(), dd.data_ptr(), dc.data_ptr(), ds.data_ptr())
    kp, de = ex[0].extract_end()
    assert kp.tobytes() == want[0][0].tobytes()
    with pytest.raises(orbx.OrbxError):
        ex[0].extract_end()
    ex[1].extract_begin(None)                   # the reference's silent return on an empty image
    kp, de = ex[1].extract_end()
    assert len(kp) == 0 and de.shape == (0, 32)


def test_batch_sharded_over_handles(orbx, synth):
    """orbx_extract_batch_multi: one batch over several handles (one per GPU on a multi-GPU node; here three handles on the box's one
    GPU, and as many devices as the box shows), one host thread each, contiguous blocks -- the result is the one-handle result
    whatever the split, including a split with more handles than frames."""
    import torch
    W, H, B = 640, 480, 11
    frames = synth.stream(4, W, H, B)
    one = orbx.ORBextractor(1000, max_width=W, max_height=H, max_batch=B)
    k1, d1, c1 = [a.copy() for a in one.extract_batch_raw(frames)]
    ndev = max(1, torch.cuda.device_count())
    hs = [orbx.ORBextractor(1000, max_width=W, max_height=H, max_batch=4, device=i % ndev) for i in range(3)]
    k3, d3, c3 = orbx.extract_batch_multi(hs, frames)
    assert np.array_equal(c1, c3) and c3.min() > 900
    for f in range(B):
        n = int(c1[f])
        assert k1[f, :n].tobytes() == k3[f, :n].tobytes() and np.array_equal(d1[f, :n], d3[f, :n])
    k2, d2, c2 = orbx.extract_batch_multi(hs, frames[:2])                      # fewer frames than handles
    assert np.array_equal(c2, c1[:2]) and k2[1, :c2[1]].tobytes() == k1[1, :c1[1]].tobytes()
    with pytest.raises(orbx.OrbxError):                                        # a block larger than its handle's max_batch
        orbx.extract_batch_multi(hs[:2], frames)
    with pytest.raises(orbx.OrbxError):
        orbx.extract_batch_multi([hs[0], hs[0]], frames[:4])
