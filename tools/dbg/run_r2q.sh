#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2q; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_extractor_gpu.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_gpu.log
for of in 0 1; do
  ORBX_OCT_FAST=$of python3 tools/dbg/ab_run.py 640 480 1000 64 30 2>&1 | tail -1 | cut -c1-200
  ORBX_OCT_FAST=$of python3 tools/dbg/ab_run.py 640 480 1000 1 30 2>&1 | tail -1| cut -c1-200
  ORBX_OCT_FAST=$of python3 tools/dbg/ab_run.py 1920 1080 4000 8 20 2>&1 | tail -1| cut -c1-200
  ORBX_OCT_FAST=$of python3 tools/dbg/ab_run.py 1241 376 2000 32 20 2>&1 | tail -1| cut -c1-200
done
timeout -k 10 200 python3 tests/tools/stress_parity.py 100 7 2>&1 | tail -3
