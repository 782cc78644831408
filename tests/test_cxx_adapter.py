"""The C++ adapters keep the reference's class names and call shapes.  CPU: they compile against the
C-ABI headers.  GPU: a Frame.cc-shaped caller gets exactly what the Python/C-ABI path returns."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cxx_adapter_check.cc")


def _build(orbx, out):
    orbx.build()
    libdir = os.path.dirname(orbx.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", SRC, "-o", out, "-L" + libdir, "-lorbx",
                           "-Wl,-rpath," + libdir])
    return out


def test_adapters_compile_and_link(orbx, tmp_path):
    exe = _build(orbx, str(tmp_path / "adapter_check"))
    assert subprocess.run([exe, "compile-only"]).returncode == 0


def test_pose_adapters_relocalise(orbx, tmp_path):
    """host/PnPsolver.h: Relocalization's call sequence on a synthetic scene with a third of the matches wrong (host code,
    no GPU needed)."""
    exe = _build(orbx, str(tmp_path / "adapter_check"))
    out = subprocess.run([exe, "pose"], capture_output=True, text=True, check=True).stdout.split()
    n_inl, good, wrong_kept = int(out[0]), int(out[1]), int(out[2])
    assert n_inl >= 38 and good == 40 and wrong_kept == 0
    assert np.abs(np.array([float(v) for v in out[3:6]]) - [0.3, -0.1, 0.2]).max() < 2e-3


@pytest.mark.gpu
def test_adapter_equals_cabi(orbx, synth, tmp_path):
    exe = _build(orbx, str(tmp_path / "adapter_check"))
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
    n_init, self_init, level0 = int(out[5]), int(out[6]), int(out[7])
    assert level0 == int((kps["octave"] == 0).sum()) and n_init == self_init and 0.9 * level0 <= n_init <= level0
