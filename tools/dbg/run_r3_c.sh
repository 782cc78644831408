#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3c
python -m pytest tests -m gpu -q -x > gpurun_out/r3c/pytest_gpu.log 2>&1; tail -5 gpurun_out/r3c/pytest_gpu.log
timeout -k 10 200 python tests/tools/stress_kf.py 120 7 > gpurun_out/r3c/stress_kf.txt 2>&1; tail -5 gpurun_out/r3c/stress_kf.txt
