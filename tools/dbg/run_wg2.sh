#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3wg2; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -3 $O/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
SKIP_TESTS=1 bash tools/dbg/run_ab_stats.sh default rs64 rs128 rs512 kpre 2>&1 | tee $O/ab.txt
ORBX_LIB=$GRAFT_REPO_ROOT/my-slam_amd/lib/variants/liborbx_kpre.so timeout -k 10 600 python3 -m pytest tests/test_extractor_gpu.py tests/test_matcher_gpu.py -m gpu -x -q 2>&1 | tail -2
