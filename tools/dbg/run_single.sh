#!/bin/bash
# GPU call: single-frame latency (blocking host API, one HIP graph per call) with and without the resize chain on a side stream
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 0 1 0 1; do
ORBX_OVERLAP_PYRAMID=$v python3 - <<PY
import sys, time, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import conftest
import my_slam_amd as M
import my_slam_amd.synth as synth
for (W, H, n) in ((640, 480, 1000), (1241, 376, 2000)):
    img = synth.texture(2, W, H)
    ex = M.ORBextractor(n, max_width=W, max_height=H)
    for _ in range(30): ex(img)
    ts = []
    for _ in range(300):
        t0 = time.perf_counter(); k, d = ex(img); ts.append(time.perf_counter() - t0)
    print("overlap=$v %dx%d n=%d: median %.4f ms, %d keypoints" % (W, H, n, float(np.median(ts)) * 1e3, len(k)))
PY
done
