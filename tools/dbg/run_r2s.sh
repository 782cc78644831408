#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2s; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_matcher_gpu.py tests/test_extractor_gpu.py tests/test_frame_grid.py tests/test_vocabulary.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o s -- python3 tools/dbg/ab_match.py > $O/s.log 2>&1
python3 - <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/r2s/prof/s_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if r['Name'].startswith(('k_best2_mfma','k_accept_rot','k_expand','k_merge')): print("%-16s calls %s avg_us %8.2f" % (r['Name'][:14], r['Calls'], float(r['AverageNs'])/1e3))
PY
