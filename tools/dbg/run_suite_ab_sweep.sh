#!/bin/bash
# full GPU suite on the built library, A/B of the given specs (run_ab_env.sh), then a randomised extraction sweep of $SWEEP seconds
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3s; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -3 $O/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
bash tools/dbg/run_ab_env.sh "$@" || exit 1
bash tools/dbg/run_sweep_mid.sh ${SWEEP:-200}
