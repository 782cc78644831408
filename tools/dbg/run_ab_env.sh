#!/bin/bash
# A/B of library variants with environment settings (on the GPU box): per "variant[@VAR=value[,VAR=value]]" the rocprofv3 per-kernel
# averages and ms_per_step of the bench step, twice.  TESTS=1 runs the extractor parity tests on every variant first.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3abe; mkdir -p $O
one() { # spec rep
  local spec=$1 lib=${1%%@*} envs=""; [ "$spec" != "$lib" ] && envs=${spec#*@}
  ( if [ "$lib" != default ]; then export ORBX_LIB=$GRAFT_REPO_ROOT/my-slam_amd/lib/variants/liborbx_$lib.so; fi
    IFS=','; for kv in $envs; do export "$kv"; done; unset IFS
    tag=$(echo "$spec" | tr '@=,' '___')
    if [ "${TESTS:-0}" = "1" ] && [ "$2" = "1" ]; then timeout -k 10 600 python3 -m pytest tests/test_extractor_gpu.py -m gpu -x -q 2>&1 | tail -1; fi
    rm -rf $O/st_$tag
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$tag -o bench -- python3 bench.py --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs > $O/bench_$tag.json 2> $O/bench_$tag.err || { echo "$spec failed"; tail -3 $O/bench_$tag.err; exit 1; }
    python3 - <<PY
import csv,glob,json
f=glob.glob("$O/st_$tag/**/*kernel_stats.csv",recursive=True)[0]
tot=0; parts=[]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if n.startswith(("k_","void k_")) and int(r["Calls"])>=100:
        parts.append("%s %.1f"%(n.replace("void ","").split("(")[0][:14],float(r["AverageNs"])/1e3)); tot+=float(r["TotalDurationNs"])/120e3
j=json.loads(open("$O/bench_$tag.json").read().strip().splitlines()[-1]); print("%-28s ms %.4f kern %.1f | "%("$spec",j["ms_per_step"],tot)+"; ".join(parts))
PY
  ) || exit 1
}
for rep in 1 2; do for spec in "$@"; do one "$spec" $rep || exit 1; done; done
