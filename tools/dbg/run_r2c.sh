#!/bin/bash
# GPU call: adapters + KF projection search + harness; randomised projection sweep
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_cxx_adapter.py tests/test_frame_grid.py tests/test_track_harness_gpu.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_gpu.log
timeout -k 10 200 python3 tests/tools/stress_proj.py 90 2 > $O/stress_proj.txt 2>&1; echo "stress rc=$?"; tail -3 $O/stress_proj.txt
