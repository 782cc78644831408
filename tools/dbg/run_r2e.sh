#!/bin/bash
# GPU call: rocprofv3 per-kernel stats of the bench step (MFMA matcher)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2e; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o mfma -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs > $O/bench.json 2> $O/bench.err; echo "rc=$?"
ls $O/prof | head; python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r2e/prof/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    print("%-60s calls %6s avg_us %9.2f total_ms %8.2f" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
