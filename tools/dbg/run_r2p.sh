#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2p; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_extractor_gpu.py tests/test_stereo_gpu.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
ORBX_LIB=$PWD/my-slam_amd/lib/variants/liborbx_fusetrace.so python3 tools/dbg/fuse_trace.py 640 480 1000 64
for fz in 0 1; do
  ORBX_PYRAMID_FUSE=$fz python3 tools/dbg/ab_run.py 640 480 1000 64 30 2>&1 | tail -1 | cut -c1-200
  ORBX_PYRAMID_FUSE=$fz python3 tools/dbg/ab_run.py 640 480 1000 1 30 2>&1 | tail -1| cut -c1-200
  ORBX_PYRAMID_FUSE=$fz python3 tools/dbg/ab_run.py 1920 1080 4000 8 20 2>&1 | tail -1| cut -c1-200
  ORBX_PYRAMID_FUSE=$fz python3 tools/dbg/ab_run.py 1241 376 2000 32 20 2>&1 | tail -1| cut -c1-200
done
