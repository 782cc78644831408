#!/bin/bash
# GPU call: bench stage times under several values of one environment variable:  run_env.sh VAR v1 v2 ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/env; mkdir -p $O
var=$1; shift
if [ "${WITH_TESTS:-0}" = "1" ]; then
  timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest_gpu.log
  [ $rc -ne 0 ] && exit $rc
fi
for v in "$@"; do
  export $var=$v
  timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-host-api --no-pipelined > $O/bench_$v.json 2> $O/bench_$v.err || { echo "bench $var=$v failed"; tail -5 $O/bench_$v.err; exit 1; }
  python3 -c "
import json,sys
d=json.load(open('$O/bench_$v.json'))
print('$var=$v', d['ms_per_step'], d['roofline']['stage_ms'], d['matches_per_step'], d['extra_configs'][0]['ms_per_step'], d['extra_configs'][0]['matches_per_step'])"
done
