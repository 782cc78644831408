#!/bin/bash
# GPU call: a randomised extraction sweep (GPU vs oracle, bit patterns) on the code as built; seconds as $1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/sweep; mkdir -p $O
S=${1:-400}
timeout -k 10 $((S + 100)) python3 tests/tools/stress_parity.py $S > $O/stress_parity.txt 2>&1; echo "parity rc=$?"; tail -4 $O/stress_parity.txt
