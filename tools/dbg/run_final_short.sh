#!/bin/bash
# GPU call: parity suite, the round's evidence (tools/prof_round.sh), memory counters and a short randomised sweep on the code as built
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $O/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
bash tools/prof_round.sh r03 > $O/prof_round.log 2>&1; echo "prof_round rc=$?"
bash tools/prof_mem.sh r3mem > $O/mem.log 2>&1; python3 tools/pmc_summary.py r3mem > $O/mem_summary.txt 2>&1
timeout -k 10 300 python3 tests/tools/stress_parity.py ${SWEEP:-200} > $O/stress_parity_last.txt 2>&1; echo "parity rc=$?"; grep "cases ok" $O/stress_parity_last.txt | tail -1
