#!/bin/bash
# GPU call: the round's final evidence in one go -- parity suite, bench line + kernel stats + PMC + traffic (run_profiles_r02.sh),
# randomised sweeps, other shapes, tracking loop
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
bash tools/dbg/run_profiles_r02.sh > $O/profiles.log 2>&1; echo "profiles rc=$?"; tail -c 400 $O/profiles.log
timeout -k 10 300 python3 tests/tools/stress_parity.py 240 > $O/stress_parity.txt 2>&1; echo "parity rc=$?"; tail -1 $O/stress_parity.txt
timeout -k 10 130 python3 tests/tools/stress_proj.py 80 > $O/stress_proj.txt 2>&1; echo "proj rc=$?"; tail -1 $O/stress_proj.txt
timeout -k 10 130 python3 tests/tools/stress_bow.py 80 > $O/stress_bow.txt 2>&1; echo "bow rc=$?"; tail -1 $O/stress_bow.txt
timeout -k 10 200 python3 tests/tools/bench_extra.py > $O/bench_extra.json 2> $O/bench_extra.err; echo "extra rc=$?"
timeout -k 10 200 bash tools/track/run.sh > $O/track.log 2>&1; echo "track rc=$?"; grep "^{" $O/track.log | cut -c1-300
