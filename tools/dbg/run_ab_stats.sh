#!/bin/bash
# A/B of library variants (on the GPU box): extractor parity on the built library, then per variant the rocprofv3 per-kernel averages and
# ms_per_step of the bench step.  usage: run_ab_stats.sh <variant|default> ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3ab; mkdir -p $O
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout -k 10 600 python3 -m pytest tests/test_extractor_gpu.py tests/test_stereo_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
fi
for rep in 1 2; do
for lib in "$@"; do
if [ "$lib" != default ]; then export ORBX_LIB=$GRAFT_REPO_ROOT/my-slam_amd/lib/variants/liborbx_$lib.so; else unset ORBX_LIB; fi
rm -rf $O/st_$lib
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$lib -o bench -- python3 bench.py --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs > $O/bench_$lib.json 2> $O/bench_$lib.err || { echo "$lib failed"; tail -3 $O/bench_$lib.err; exit 1; }
python3 - <<PY
import csv,glob,json
f=glob.glob("$O/st_$lib/**/*kernel_stats.csv",recursive=True)[0]
tot=0; parts=[]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if n.startswith(("k_","void k_")) and int(r["Calls"])>=100:
        parts.append("%s %.1f"%(n.replace("void ","").split("(")[0][:22],float(r["AverageNs"])/1e3)); tot+=float(r["TotalDurationNs"])/120e3
j=json.loads(open("$O/bench_$lib.json").read().strip().splitlines()[-1]); print("%-8s ms_per_step %.4f kernels %.1f us | "%("$lib",j["ms_per_step"],tot)+"; ".join(parts))
PY
done
done
