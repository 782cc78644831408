#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2r; mkdir -p $O
timeout -k 10 400 python3 tests/tools/stress_parity.py 240 11 > $O/stress_parity.txt 2>&1; echo "stress rc=$?"; head -3 $O/stress_parity.txt; tail -2 $O/stress_parity.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
