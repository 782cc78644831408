"""The C++ adapters keep the reference's class names, constructors and call shapes (my-slam_amd/host/).  CPU: they compile
against the C-ABI headers, written for OpenCV 3.1.0's InputArray / OutputArray / CV_8U API (modelled by orbx_cv_compat.h on
hosts without OpenCV).  GPU: a Frame.cc-shaped caller gets exactly what the Python/C-ABI path returns, and a Tracking.cc-shaped
caller (tests/cxx/tracking_callsites.cc: the reference's own call expressions over minimal Frame / KeyFrame / MapPoint classes)
gets exactly what direct C-ABI calls return for all five Tracking-thread matchers."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CXX = os.path.join(ROOT, "tests", "cxx")
INC = ["-I" + os.path.join(ROOT, "my-slam_amd", "host"), "-I" + os.path.join(CXX, "slam_shims"), "-I" + os.path.join(ROOT, "include")]


def _build(orbx, src, out):
    orbx.build()
    libdir = os.path.dirname(orbx.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Wno-unused-parameter"] + INC +
                          [os.path.join(CXX, src), "-o", out, "-L" + libdir, "-lorbx", "-Wl,-rpath," + libdir])
    return out


def test_adapters_compile_and_link(orbx, tmp_path):
    for src in ("adapter_check.cc", "tracking_callsites.cc", "mapping_callsites.cc"):
        exe = _build(orbx, src, str(tmp_path / src[:-3]))
        assert subprocess.run([exe, "compile-only"]).returncode == 0


def test_compat_header_is_as_strict_as_opencv(orbx, tmp_path):
    """orbx_cv_compat.h must reject what OpenCV rejects, so that "the adapters compile here" means "they compile there": the
    round-1 adapter's cv::CV_8U, its `const cv::Mat &im = image` from an InputArray and its element access on an OutputArray
    must all fail to compile against it."""
    bad = {
        "qualified_macro.cc": "void f() { cv::Mat m(4, 4, cv::CV_8U); }",
        "mat_from_inputarray.cc": "void f(cv::InputArray image) { const cv::Mat &im = image; (void)im; }",
        "ptr_on_outputarray.cc": "void f(cv::OutputArray d) { d.ptr<unsigned char>(0); }",
    }
    good = "void f(cv::InputArray a, cv::OutputArray d) { cv::Mat m = a.getMat(); d.create(m.rows, 32, CV_8U); cv::Mat o = d.getMat(); (void)o; }"
    hdr = '#include "orbx_cv_compat.h"\n'
    for name, body in list(bad.items()) + [("good.cc", good)]:
        src = tmp_path / name
        src.write_text(hdr + body + "\n")
        rc = subprocess.run(["g++", "-std=c++17", "-fsyntax-only"] + INC + [str(src)], capture_output=True).returncode
        assert (rc == 0) == (name == "good.cc"), name


def test_pose_adapters_relocalise(orbx, tmp_path):
    """host/PnPsolver.h: Relocalization's call sequence on a synthetic scene with a third of the matches wrong (host code,
    no GPU needed)."""
    exe = _build(orbx, "adapter_check.cc", str(tmp_path / "adapter_check"))
    out = subprocess.run([exe, "pose"], capture_output=True, text=True, check=True).stdout.split()
    n_inl, good, wrong_kept = int(out[0]), int(out[1]), int(out[2])
    assert n_inl >= 38 and good == 40 and wrong_kept == 0
    assert np.abs(np.array([float(v) for v in out[3:6]]) - [0.3, -0.1, 0.2]).max() < 2e-3


@pytest.mark.gpu
def test_adapter_equals_cabi(orbx, synth, tmp_path):
    exe = _build(orbx, "adapter_check.cc", str(tmp_path / "adapter_check"))
    img = synth.texture(1, 640, 480)
    raw = tmp_path / "img.u8"
    img.tofile(raw)
    out = subprocess.run([exe, str(raw), "640", "480"], capture_output=True, text=True, check=True).stdout.split()
    kps, desc = orbx.ORBextractor(1000, max_width=640, max_height=480)(img)
    h = 1469598103934665603
    for i in range(len(kps)):
        for b in kps[i:i + 1].tobytes() + desc[i].tobytes():
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert int(out[0]) == len(kps) and out[1] == "%016x" % h
    assert int(out[2]) == len(kps)            # every descriptor matches itself at distance 0
    assert int(out[3]) == 179
    assert int(out[4]) == orbx.ORBmatcher.DescriptorDistance(desc[0], desc[1])
    assert int(out[5]) == 1                   # a non-CV_8UC1 image gives the empty result
    assert int(out[6]) == 1                   # an image larger than the handle's first size is extracted, not refused
    assert int(out[7]) == 1                   # mvImagePyramid is valid after operator(): interior views of reflect-101 bordered buffers
    ex = orbx.ORBextractor(1000, max_width=640, max_height=480)
    ex(img)
    hp = 1469598103934665603
    for lvl in ex.image_pyramid():
        for b in lvl.tobytes():
            hp = ((hp ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert out[8] == "%016x" % hp             # ... whose pixels are the ones the C ABI's per-level download returns


@pytest.mark.gpu
def test_tracking_callsites_equal_cabi(orbx, synth, tmp_path):
    """The reference's Tracking-thread call expressions on the drop-in classes == direct C-ABI calls (which test_frame_grid.py /
    test_vocabulary.py tie to the oracle), and building a matcher on the stack costs < 1 us after the first."""
    import test_vocabulary as TV
    exe = _build(orbx, "tracking_callsites.cc", str(tmp_path / "tracking_callsites"))
    W, H = 1241, 376
    frames, layer = synth.stream_layers(5, W, H, 2, shifts=(2, 4, 6))
    frames.tofile(tmp_path / "frames.u8"); layer.astype(np.uint8).tofile(tmp_path / "layer.u8")
    voc = str(tmp_path / "voc.txt")
    TV.make_vocabulary(voc, k=8, depth=3, seed=5)
    p = subprocess.run([exe, str(tmp_path / "frames.u8"), str(tmp_path / "layer.u8"), str(W), str(H), voc], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    rows = {ln.split()[0]: ln.split()[1:] for ln in p.stdout.splitlines() if ln.strip()}
    want = {"SearchForInitialization": 100, "SearchByBoW": 30, "SearchByProjection(last)": 200, "SearchByProjection(last,occupied)": 150,
            "SearchByProjection(map)": 100, "SearchByProjection(KF,10,100)": 50, "SearchByProjection(KF,3,64)": 0}
    for name, floor in want.items():
        assert name in rows, p.stdout
        assert int(rows[name][1]) == 1, "%s differs from the C ABI" % name
        assert int(rows[name][0]) >= floor, "%s: only %s matches" % (name, rows[name][0])
    assert float(rows["construct_ns"][0]) < 1000.0


@pytest.mark.gpu
def test_mapping_callsites_equal_cabi(orbx, synth, tmp_path):
    """The reference's LocalMapping / LoopClosing call expressions (SearchForTriangulation, Fuse x 2, SearchByBoW(KF, KF), SearchBySim3,
    SearchByProjection(KF, Scw)) on the drop-in class == direct C-ABI calls (which tests/test_kf_matchers.py ties to the oracle); for
    the Fuse variants the object graph after the call equals a replay of the reference's bookkeeping on the C ABI's match table."""
    import test_vocabulary as TV
    exe = _build(orbx, "mapping_callsites.cc", str(tmp_path / "mapping_callsites"))
    W, H = 1241, 376
    frames, layer = synth.stream_layers(5, W, H, 2, shifts=(2, 4, 6))
    frames.tofile(tmp_path / "frames.u8"); layer.astype(np.uint8).tofile(tmp_path / "layer.u8")
    voc = str(tmp_path / "voc.txt")
    TV.make_vocabulary(voc, k=8, depth=3, seed=5)
    p = subprocess.run([exe, str(tmp_path / "frames.u8"), str(tmp_path / "layer.u8"), str(W), str(H), voc], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    rows = {ln.split()[0]: ln.split()[1:] for ln in p.stdout.splitlines() if ln.strip()}
    want = {"SearchForTriangulation": 40, "SearchByBoW(KF,KF)": 30, "SearchBySim3": 100, "SearchByProjection(KF,Scw)": 100, "Fuse(KF,Scw)": 100,
            "Fuse(KF,points)": 100, "Fuse(KF,candidates)": 20}
    for name, floor in want.items():
        assert name in rows, p.stdout
        assert int(rows[name][1]) == 1, "%s differs from the C ABI\n%s" % (name, p.stdout)
        assert int(rows[name][0]) >= floor, "%s: only %s\n%s" % (name, rows[name][0], p.stdout)
