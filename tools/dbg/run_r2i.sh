#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2i; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_extractor_gpu.py tests/test_threads_gpu.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
ORBX_BATCH_TRACE=1 python3 - 2>&1 <<'PY' | tail -14
import sys, os, time
sys.path.insert(0, "tests"); import conftest
import numpy as np
import my_slam_amd as M, my_slam_amd.synth as synth
fr = synth.stream(4, 640, 480, 64)
for chunk in (8, 16, 32):
    ex = M.ORBextractor(1000, max_width=640, max_height=480, max_batch=64)
    ex.set_batch_chunk(chunk)
    for _ in range(4): ex.extract_batch_raw(fr)
PY
