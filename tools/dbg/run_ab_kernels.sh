#!/bin/bash
# A/B of a build (on the GPU box): extractor parity, then rocprofv3 per-kernel averages and ms_per_step of the bench step, twice
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3ab; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_extractor_gpu.py tests/test_stereo_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st$rep -o bench -- python3 bench.py --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs > $O/bench$rep.json 2> $O/bench$rep.err
python3 - <<PY
import csv,glob,json
f=glob.glob("$O/st$rep/**/*kernel_stats.csv",recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if n.startswith(("k_","void k_")) and int(r["Calls"])>=100:
        print("  %-40s calls %4s avg %8.1f us"%(n[:40],r["Calls"],float(r["AverageNs"])/1e3)); tot+=float(r["TotalDurationNs"])/120e3
j=json.loads(open("$O/bench$rep.json").read().strip().splitlines()[-1]); print("  kernels per step %.1f us; ms_per_step %.4f"%(tot,j["ms_per_step"]))
PY
done
