#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2t; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_matcher_gpu.py tests/test_extractor_gpu.py tests/test_frame_grid.py tests/test_vocabulary.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
for S in 0 2 3 4 5; do
  if [ $S = 0 ]; then unset ORBM_MFMA_SPLITS; else export ORBM_MFMA_SPLITS=$S; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o s$S -- python3 tools/dbg/ab_match.py > $O/s$S.log 2>&1
  python3 - $S <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/r2t/prof/s%s_kernel_stats.csv' % sys.argv[1])[0]
for r in csv.DictReader(open(f)):
    if r['Name'].startswith(('k_best2_mfma','k_accept_rot','k_merge')): print("S=%s %-16s calls %s avg_us %8.2f" % (sys.argv[1], r['Name'][:14], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
