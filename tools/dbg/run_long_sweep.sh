#!/bin/bash
# GPU call: a long randomised extraction sweep on the final code (evidence for profiles/)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
timeout -k 10 1000 python3 tests/tools/stress_parity.py 900 > $O/stress_parity_long.txt 2>&1; echo "parity rc=$?"; grep -n "PASS\|FAIL\|mismatch" $O/stress_parity_long.txt | tail -3
