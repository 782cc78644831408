#!/bin/bash
# round 3: acceptance fused into the tail of the match launch -- parity, then kernel times with and without
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3fa; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_matcher_gpu.py tests/test_frame_grid.py tests/test_bench_gpu.py tests/test_kf_matchers.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for v in 1 0; do
  ORBM_FUSE_ACCEPT=$v rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$v -o bench -- python3 bench.py --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs > $O/bench$v.json 2> $O/bench$v.err
  echo "FUSE=$v rc=$?"; python3 - <<PY
import csv,glob,json
f=glob.glob("$O/stats$v/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "best2" in n or "accept" in n:
        print("  %-40s calls %4s avg %8.1f us min %.1f"%(n[:40],r["Calls"],float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3))
j=json.loads(open("$O/bench$v.json").read().strip().splitlines()[-1]); print("  ms_per_step", j["ms_per_step"], "matches_per_step", j["config"].get("matches_per_step"))
PY
done
