#!/bin/bash
# GPU call: randomised GPU-vs-oracle sweeps on the round's final code, the other BASELINE shapes, the C++ tracking loop
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
timeout -k 10 420 python3 tests/tools/stress_parity.py 330 > $O/stress_parity.txt 2>&1; echo "parity rc=$?"; tail -3 $O/stress_parity.txt
timeout -k 10 160 python3 tests/tools/stress_proj.py 100 > $O/stress_proj.txt 2>&1; echo "proj rc=$?"; tail -3 $O/stress_proj.txt
timeout -k 10 160 python3 tests/tools/stress_bow.py 100 > $O/stress_bow.txt 2>&1; echo "bow rc=$?"; tail -3 $O/stress_bow.txt
timeout -k 10 200 python3 tests/tools/bench_extra.py > $O/bench_extra.json 2> $O/bench_extra.err; echo "extra rc=$?"; tail -c 600 $O/bench_extra.json
timeout -k 10 200 bash tools/track/run.sh > $O/track.log 2>&1; echo "track rc=$?"; tail -5 $O/track.log
