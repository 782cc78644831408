#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2l; mkdir -p $O
for e in 0 1 2 3 4; do
  ORBX_LIB=$PWD/my-slam_amd/lib/variants/liborbx_mfexp$e.so rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o e$e -- python3 tools/dbg/ab_match.py > $O/e$e.log 2>&1
  python3 - $e <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/r2l/prof/e%s_kernel_stats.csv' % sys.argv[1])[0]
for r in csv.DictReader(open(f)):
    if r['Name'].startswith(('k_best2_mfma',)): print("EXP=%s %-16s calls %s avg_us %8.2f" % (sys.argv[1], r['Name'][:14], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
