#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3bench
( time python bench.py ) > gpurun_out/r3bench/bench_default.log 2>&1
tail -4 gpurun_out/r3bench/bench_default.log | cut -c1-6000
python -m pytest tests/test_bench_gpu.py -m gpu -x -q > gpurun_out/r3bench/pytest_bench.log 2>&1; tail -15 gpurun_out/r3bench/pytest_bench.log
