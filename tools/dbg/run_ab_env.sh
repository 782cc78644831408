#!/bin/bash
# A/B over values of one environment variable (on the GPU box): ms_per_step and the per-kernel averages of the bench step.
# usage: VAR=ORBX_PYRAMID_TILES VALS="0 4,32,32 5,32,32" tools/dbg/run_ab_env.sh [kernel-name filter]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/abe; mkdir -p $O
FILT=${1:-k_}
for v in $VALS; do
  export $VAR=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$v -o bench -- python3 bench.py --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs > $O/bench_$v.json 2> $O/bench_$v.err
  echo "$VAR=$v rc=$?"; python3 - <<PY
import csv,glob,json
f=glob.glob("$O/st_$v/**/*kernel_stats.csv",recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "$FILT" in n and int(r["Calls"])>=100:
        print("  %-44s calls %4s avg %8.1f us total/step %.1f"%(n[:44],r["Calls"],float(r["AverageNs"])/1e3,float(r["TotalDurationNs"])/120e3)); tot+=float(r["TotalDurationNs"])/120e3
j=json.loads(open("$O/bench_$v.json").read().strip().splitlines()[-1]); print("  filtered kernels per step %.1f us; ms_per_step %.4f"%(tot,j["ms_per_step"]))
PY
done
