#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2g; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_matcher_gpu.py tests/test_extractor_gpu.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
ORBX_BATCH_TRACE=1 python3 - 2>&1 <<'PY' | tail -12
import sys, os, time
sys.path.insert(0, "tests"); import conftest
import numpy as np
import my_slam_amd as M, my_slam_amd.synth as synth
fr = synth.stream(4, 640, 480, 64)
for chunk in (16, 32):
    ex = M.ORBextractor(1000, max_width=640, max_height=480, max_batch=64)
    ex.set_batch_chunk(chunk)
    for _ in range(5): ex.extract_batch_raw(fr)
PY
for S in 1 3; do
  ORBM_MFMA_SPLITS=$S rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o s$S -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs > /dev/null 2> $O/bench_s$S.err
  python3 - $S <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/r2g/prof/s%s_kernel_stats.csv' % sys.argv[1])[0]
for r in csv.DictReader(open(f)):
    if r['Name'].startswith(('k_best2_mfma','k_accept_rot','k_expand','k_merge')): print("S=%s %-16s avg_us %8.2f" % (sys.argv[1], r['Name'][:14], float(r['AverageNs'])/1e3))
PY
done
