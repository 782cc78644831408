#!/bin/bash
# A/B of a build (on the GPU box): matcher parity, then rocprofv3 averages of the match kernels in the bench step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/abm; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_matcher_gpu.py tests/test_frame_grid.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for S in ${SPLITS:-default}; do
  if [ "$S" = default ]; then unset ORBM_MFMA_SPLITS; else export ORBM_MFMA_SPLITS=$S; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$S -o bench -- python3 bench.py --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs > $O/bench_$S.json 2> $O/bench_$S.err
  python3 - <<PY
import csv,glob,json
f=glob.glob("$O/st_$S/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "best2" in r["Name"] or "accept" in r["Name"]: print("  S=$S %-30s calls %4s avg %8.1f us min %.1f"%(r["Name"][:30],r["Calls"],float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3))
j=json.loads(open("$O/bench_$S.json").read().strip().splitlines()[-1]); print("  ms_per_step %.4f"%j["ms_per_step"])
PY
done
