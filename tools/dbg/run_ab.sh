#!/bin/bash
# GPU call: full parity suite, then the bench line's per-stage times (one line per ORBX_LIB given as arguments; none = the built library)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ab; mkdir -p $O
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest_gpu.log
  [ $rc -ne 0 ] && exit $rc
fi
libs="$@"; [ -z "$libs" ] && libs=default
for lib in $libs; do
  if [ "$lib" != default ]; then export ORBX_LIB=$GRAFT_REPO_ROOT/my-slam_amd/lib/variants/liborbx_$lib.so; else unset ORBX_LIB; fi
  timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-host-api --no-pipelined > $O/bench_$lib.json 2> $O/bench_$lib.err || { echo "bench $lib failed"; tail -5 $O/bench_$lib.err; exit 1; }
  python3 -c "
import json,sys
d=json.load(open('$O/bench_$lib.json'))
print('$lib', d['ms_per_step'], d['roofline']['stage_ms'], d['matches_per_step'], d['extra_configs'][0]['ms_per_step'], d['extra_configs'][0]['matches_per_step'])"
done
