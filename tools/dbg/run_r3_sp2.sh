#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3sp; mkdir -p $O
for S in ${SPLITS:-1 2 3 4}; do
  ORBM_MFMA_SP=1 ORBM_MFMA_SPLITS=$S rocprofv3 --kernel-trace --stats --output-format csv -d $O/statsS$S -o bench -- python3 bench.py --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs --steps 30 > $O/benchS$S.json 2> $O/benchS$S.err
  echo "S=$S rc=$?"; python3 - <<PY
import csv,glob,json
f=glob.glob("$O/statsS$S/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "best2" in n or "accept" in n:
        print("  %-40s calls %4s avg %8.1f us min %.1f"%(n[:40],r["Calls"],float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3))
PY
done
