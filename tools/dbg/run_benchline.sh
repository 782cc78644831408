#!/bin/bash
# GPU call: the default bench line (as the driver runs it) + the side measurements file
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 400 python3 bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1500 $O/bench_line.json
timeout -k 10 200 python3 tests/tools/bench_extra.py > gpurun_out/final/bench_extra.json 2> gpurun_out/final/bench_extra.err; echo "extra rc=$?"
