#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3md; mkdir -p $O
for D in ${DBGS:-0 1 2 4}; do
  ORBM_MFMA_SP=0 ORBM_MFMA_DBG=$D rocprofv3 --kernel-trace --stats --output-format csv -d $O/statsD$D -o bench -- python3 bench.py --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs --steps 30 > $O/benchD$D.json 2> $O/benchD$D.err
  echo "DBG=$D rc=$?"; python3 - <<PY
import csv,glob
f=glob.glob("$O/statsD$D/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "best2" in n:
        print("  %-40s calls %4s avg %8.1f us min %.1f"%(n[:40],r["Calls"],float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3))
PY
done
