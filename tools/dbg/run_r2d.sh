#!/bin/bash
# GPU call: MFMA dense matcher -- parity (matcher + extractor + frame grid + vocabulary + harness suites), then A/B timing vs the popcount kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2d; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
for mode in mfma popcount; do
  ORBM_DENSE=$mode timeout -k 10 200 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-host-api --no-pipelined > $O/bench_$mode.json 2> $O/bench_$mode.err; echo "bench $mode rc=$?"
  python3 -c "
import json,sys
d=json.load(open('$O/bench_$mode.json'))
print('$mode', d['ms_per_step'], d['roofline']['stage_ms'], d['matches_per_step'], d['extra_configs'][0]['ms_per_step'], d['extra_configs'][0]['matches_per_step'])"
done
