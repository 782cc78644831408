#!/bin/bash
# round 3: shared-row resize kernel -- parity, then per-kernel times with and without it
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3rs; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_extractor_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for v in 1 0; do
  ORBX_RESIZE6=$v rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$v -o bench -- python3 bench.py --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs > $O/bench$v.json 2> $O/bench$v.err
  echo "RESIZE6=$v rc=$?"; python3 - <<PY
import csv,glob,json
f=glob.glob("$O/stats$v/**/*kernel_stats.csv",recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if n.startswith("k_") or n.startswith("void k_"):
        print("  %-40s calls %4s avg %8.1f us"%(n[:40],r["Calls"],float(r["AverageNs"])/1e3))
print(open("$O/bench$v.json").read()[:400])
PY
done
