#!/bin/bash
# full GPU suite on the built library, then run_ab_env.sh with the given specs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3s; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -3 $O/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
bash tools/dbg/run_ab_env.sh "$@"
