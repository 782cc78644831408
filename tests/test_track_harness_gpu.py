"""The config-5-shaped per-frame loop with the host side in C++ (tools/track/track_harness.cc) builds against include/*.h,
links liborbx.so and runs: extract -> grid -> BoW -> SearchByBoW -> windowed search through the C ABI."""
import json
import os
import subprocess

import numpy as np
import pytest

import conftest  # noqa
import oracle_lib as O
from test_vocabulary import make_vocabulary

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cxx_tracking_harness(orbx, synth, tmp_path):
    exe = str(tmp_path / "track_harness")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools/track/track_harness.cc"),
                    "-L", os.path.join(ROOT, "my-slam_amd/lib"), "-lorbx", "-Wl,-rpath," + os.path.join(ROOT, "my-slam_amd/lib"), "-o", exe],
                   check=True)
    W, H, K = 640, 480, 8
    synth.stream(5, W, H, K).tofile(str(tmp_path / "frames.raw"))
    make_vocabulary(str(tmp_path / "voc.txt"), 10, 3, seed=1)
    out = subprocess.run([exe, str(tmp_path / "frames.raw"), str(W), str(H), str(K), str(tmp_path / "voc.txt"), "1000", "2"],
                         check=True, capture_output=True, text=True, timeout=120).stdout
    r = json.loads(out.strip().splitlines()[-1])
    assert r["frames_timed"] == K - 5
    assert r["matches_per_frame"]["bow"] > 100 and r["matches_per_frame"]["projection"] > 300
    # the same SearchByBoW through the Python binding on frames 6 -> 7 agrees with what the harness counted on average
    ex = orbx.ORBextractor(1000, max_width=W, max_height=H)
    assert len(ex(np.fromfile(str(tmp_path / "frames.raw"), np.uint8).reshape(K, H, W)[0])[0]) > 900


def test_cxx_tracking_harness_pose_stages(orbx, synth, tmp_path):
    """BASELINE config 5's whole loop on the three-depth scene: the chained PoseOptimization estimates
    follow the camera (ground truth: +baseline along x per frame) and EPnP RANSAC relocalises without a prior."""
    exe = str(tmp_path / "track_harness")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools/track/track_harness.cc"),
                    "-L", os.path.join(ROOT, "my-slam_amd/lib"), "-lorbx", "-Wl,-rpath," + os.path.join(ROOT, "my-slam_amd/lib"), "-o", exe],
                   check=True)
    W, H, K = 1241, 376, 17
    frames, layer = synth.stream_layers(5, W, H, K, shifts=(2, 4, 6))
    frames.tofile(str(tmp_path / "frames.raw"))
    layer.tofile(str(tmp_path / "layer.raw"))
    make_vocabulary(str(tmp_path / "voc.txt"), 10, 3, seed=1)
    out = subprocess.run([exe, str(tmp_path / "frames.raw"), str(W), str(H), str(K), str(tmp_path / "voc.txt"), "2000", "2",
                          str(tmp_path / "layer.raw"), "0.5", "2", "4", "6"],
                         check=True, capture_output=True, text=True, timeout=180).stdout
    lines = [json.loads(l) for l in out.strip().splitlines()]
    p = lines[-2]["pose"]
    assert p["inliers_per_frame"] > 100
    # chained over 16 frames; 1 px of the nearest layer is 1/6 of a baseline
    assert p["max_translation_error_in_baselines"] < 0.25
    assert p["relocalizations"] == "2/2" and p["max_reloc_error_in_baselines"] < 0.25
    assert p["reloc_search_by_projection_kf_matches"] > 100      # SearchByProjection(F, pKF, sFound, 10, 100) after PnP finds more points


def _read_dump(path, pose):
    raw = np.fromfile(path, np.uint8)
    o = [0]

    def take(dtype, n):
        a = raw[o[0]:o[0] + n * np.dtype(dtype).itemsize].view(dtype).copy()
        o[0] += n * np.dtype(dtype).itemsize
        return a
    npv, ncu, fpn, fpi, fcn, fci, pflag, nm_bow = take(np.int32, 8)
    assert bool(pflag) == pose
    d = {"nm_bow": int(nm_bow)}
    d["pk"] = take(O.KP_DTYPE, npv); d["pd"] = take(np.uint8, npv * 32).reshape(-1, 32)
    d["ck"] = take(O.KP_DTYPE, ncu); d["cd"] = take(np.uint8, ncu * 32).reshape(-1, 32)
    d["pfv"] = (take(np.int32, fpn), take(np.int32, fpn + 1), take(np.int32, fpi))
    d["cfv"] = (take(np.int32, fcn), take(np.int32, fcn + 1), take(np.int32, fci))
    d["match_f"] = take(np.int32, ncu)
    if pose:
        d["Tcw"] = take(np.float32, 16); d["Tlw"] = take(np.float32, 16); d["xw"] = take(np.float32, 3 * npv).reshape(-1, 3)
        d["cur_match"] = take(np.int32, ncu); d["nm_proj"] = int(take(np.int32, 1)[0])
    else:
        d["qr"] = take(np.float32, npv); d["m12"] = take(np.int32, npv); d["nm_proj"] = int(take(np.int32, 1)[0])
    assert o[0] == len(raw)
    return d


@pytest.mark.parametrize("pose", [False, True])
def test_tracking_loop_tables_equal_oracle(orbx, synth, tmp_path, pose):
    """N4 as parity: the C++ loop dumps, for every frame, the features it extracted, the FeatureVectors, the poses it used and the
    match tables its two matchers returned (ORBX_TRACK_DUMP); the oracle's restatements of ORBmatcher::SearchByBoW and of
    SearchByProjection(CurrentFrame, LastFrame, 15, mono) (with pose) / the windowed best-match search + rotation cull (without) run
    on exactly those inputs and must return exactly those tables and counts."""
    import oracle_lib as O  # noqa: F811
    exe = str(tmp_path / "track_harness")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools/track/track_harness.cc"),
                    "-L", os.path.join(ROOT, "my-slam_amd/lib"), "-lorbx", "-Wl,-rpath," + os.path.join(ROOT, "my-slam_amd/lib"), "-o", exe],
                   check=True)
    W, H, K = 1241, 376, 12
    frames, layer = synth.stream_layers(5, W, H, K, shifts=(2, 4, 6))
    frames.tofile(str(tmp_path / "frames.raw")); layer.tofile(str(tmp_path / "layer.raw"))
    make_vocabulary(str(tmp_path / "voc.txt"), 10, 4, seed=1)
    dump = tmp_path / "dump"
    dump.mkdir()
    cmd = [exe, str(tmp_path / "frames.raw"), str(W), str(H), str(K), str(tmp_path / "voc.txt"), "2000", "2"]
    if pose:
        cmd += [str(tmp_path / "layer.raw"), "0.5", "2", "4", "6"]
    subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=300, env=dict(os.environ, ORBX_TRACK_DUMP=str(dump)))
    fx, fy, cx, cy = 718.856, 718.856, 607.1928, 185.2157
    sf = np.cumprod(np.concatenate([[np.float32(1)], np.full(7, np.float32(1.2), np.float32)])).astype(np.float32)
    bounds = (0.0, float(W), 0.0, float(H))
    checked = 0
    for k in range(1, K):
        d = _read_dump(str(dump / ("frame_%03d.bin" % k)), pose)
        mf, nm = O.search_by_bow(d["pd"], d["pk"]["angle"], d["pfv"], d["cd"], d["ck"]["angle"], d["cfv"], 0.7, True, None)
        assert nm == d["nm_bow"] and np.array_equal(mf, d["match_f"]), "SearchByBoW differs at frame %d" % k
        assert nm > 150
        g = O.FrameGrid(d["ck"], *bounds)
        n0 = len(d["pk"])
        if pose:
            cur_obs = np.full(len(d["ck"]), -1, np.int32)
            cm, nmp = O.search_by_projection_last(np.ones(n0, np.uint8), d["xw"], d["pd"], np.ones(n0, np.int32), d["pk"], d["Tcw"], d["Tlw"],
                                                  (fx, fy, cx, cy), 0.0, 0.0, bounds, sf, g, d["cd"], cur_obs, 15.0, True, True, None)
            assert nmp == d["nm_proj"] and np.array_equal(cm, d["cur_match"]), "SearchByProjection(last) differs at frame %d" % k
            assert nmp > 600
        else:
            pk = d["pk"]
            mn = np.maximum(pk["octave"] - 1, -1).astype(np.int32); mx = (pk["octave"] + 1).astype(np.int32)
            bi, bd, _ = g.search_area_best2(d["pd"], pk["x"].copy(), pk["y"].copy(), d["qr"], mn, mx, d["cd"])
            m12 = np.where(bd <= 100, bi, -1).astype(np.int32)
            aq = np.ascontiguousarray(pk["angle"], np.float32); at = np.ascontiguousarray(d["ck"]["angle"], np.float32)
            n = O.lib().oro_rot_filter(O._p(aq), O._p(at), O._p(m12), n0)
            assert n == d["nm_proj"] and np.array_equal(m12, d["m12"]), "windowed search differs at frame %d" % k
            assert n > 800
        checked += 1
    assert checked == K - 1


def test_pose_from_gpu_matches_python(orbx, synth, tmp_path):
    """The same chain through the Python mirror: extract (GPU) -> BoW (GPU) -> SearchByBoW -> PnPsolver / PoseOptimization."""
    W, H, K = 1241, 376, 3
    fx, fy, cx, cy, base, shifts = 718.856, 718.856, 607.1928, 185.2157, 0.5, (2, 4, 6)
    frames, layer = synth.stream_layers(5, W, H, K, shifts=shifts)
    make_vocabulary(str(tmp_path / "voc.txt"), 10, 3, seed=1)
    voc = orbx.ORBVocabulary(str(tmp_path / "voc.txt"))
    ex = orbx.ORBextractor(2000, max_width=W, max_height=H)
    mt = orbx.ORBmatcher(0.7, True, max_queries=4096, max_train=4096, max_pairs=1 << 21)
    feats = []
    for k in range(K):
        kp, de = ex(frames[k])
        feats.append((kp, de, voc.transform(de, 2)[1]))
    for k in (1, 2):
        pk, pd, pfv = feats[k - 1]
        ck, cd, cfv = feats[k]
        match_f, nm = mt.SearchByBoW(pk, pd, pfv, ck, cd, cfv)
        sel = np.flatnonzero(match_f >= 0)
        assert nm == len(sel) and nm > 150
        j = match_f[sel]
        Z = (fx * base / np.array(shifts, np.float64))[layer[np.clip(np.rint(pk["y"][j]).astype(int), 0, H - 1),
                                                             np.clip(np.rint(pk["x"][j]).astype(int), 0, W - 1)]]
        xw = np.stack([(pk["x"][j] - cx) * Z / fx + (k - 1) * base, (pk["y"][j] - cy) * Z / fy, Z], 1)     # true previous pose
        obs = np.stack([ck["x"][sel], ck["y"][sel]], 1)
        s2 = (np.float32(1.2) ** ck["octave"][sel].astype(np.float32)) ** 2
        Tprev = np.eye(4, dtype=np.float32)
        Tprev[0, 3] = -(k - 1) * base
        T, outl, ninl = orbx.PoseOptimization(obs, 1.0 / s2, xw, fx, fy, cx, cy, Tprev)
        assert ninl > 100 and abs(T[0, 3] + k * base) < 0.1 * base and abs(T[1, 3]) < 0.1 * base and abs(T[2, 3]) < 0.25 * base
        assert np.abs(T[:3, :3] - np.eye(3)).max() < 2e-3
        s = orbx.PnPsolver(obs, s2, xw, fx, fy, cx, cy)
        s.SetRansacParameters(0.99, 10, 300, 4, 0.5, 5.991)
        Tr = None
        for _ in range(100):
            Tr, no_more, inl, n_in = s.iterate(5)
            if Tr is not None or no_more:
                break
        assert Tr is not None and n_in > 50
        T2, _, n2 = orbx.PoseOptimization(obs, 1.0 / s2, xw, fx, fy, cx, cy, Tr)
        assert n2 > 100 and np.abs(T2 - T).max() < 0.1 * base
