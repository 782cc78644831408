#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > /tmp/hb.py <<'PY'
import sys, os, time
sys.path.insert(0, "tests"); import conftest
import numpy as np
import my_slam_amd as M, my_slam_amd.synth as synth
fr = synth.stream(4, 640, 480, 64)
for chunk in (16, 8, 32):
    ex = M.ORBextractor(1000, max_width=640, max_height=480, max_batch=64)
    ex.set_batch_chunk(chunk)
    for _ in range(4): ex.extract_batch_raw(fr)
    ts = []
    for _ in range(25):
        t0 = time.perf_counter(); ex.extract_batch_raw(fr); ts.append(time.perf_counter() - t0)
    print("chunk %2d %s median %.3f ms  min %.3f ms" % (chunk, "equal" if os.environ.get("ORBX_BATCH_EQUAL") else "half first/last", 1e3 * float(np.median(ts)), 1e3 * min(ts)))
PY
for rep in 1 2; do
python3 /tmp/hb.py 2>&1 | grep chunk
ORBX_BATCH_EQUAL=1 python3 /tmp/hb.py 2>&1 | grep chunk
done
