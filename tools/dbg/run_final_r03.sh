#!/bin/bash
# GPU call: the round's final evidence -- full parity suite, bench line + kernel stats + PMC + traffic (tools/prof_round.sh), memory-system
# counters (tools/prof_mem.sh), then the randomised sweeps on the same build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $O/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
bash tools/prof_round.sh r03 > $O/prof_round.log 2>&1; echo "prof_round rc=$?"; tail -c 300 $O/prof_round.log; echo
bash tools/prof_mem.sh r3mem > $O/mem.log 2>&1; python3 tools/pmc_summary.py r3mem > $O/mem_summary.txt 2>&1; echo "mem counters: $(grep -c mean $O/mem_summary.txt) lines"
timeout -k 10 700 python3 tests/tools/stress_parity.py ${SWEEP:-600} > $O/stress_parity.txt 2>&1; echo "parity rc=$?"; grep "cases ok" $O/stress_parity.txt | tail -1
timeout -k 10 130 python3 tests/tools/stress_kf.py 80 > $O/stress_kf.txt 2>&1; echo "kf rc=$?"; tail -1 $O/stress_kf.txt
timeout -k 10 130 python3 tests/tools/stress_proj.py 80 > $O/stress_proj.txt 2>&1; echo "proj rc=$?"; tail -1 $O/stress_proj.txt
timeout -k 10 130 python3 tests/tools/stress_bow.py 80 > $O/stress_bow.txt 2>&1; echo "bow rc=$?"; tail -1 $O/stress_bow.txt
