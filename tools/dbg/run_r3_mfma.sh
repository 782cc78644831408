#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3mfma
for V in "2 3" "4 2" "3 2"; do
  set -- $V
  (cd my-slam_amd && rm -f build/orbm_mfma.o && make -s HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DMF_QB=$1 -DMF_OCC=$2" 2>&1 | grep -E "error" )
  for S in 0 2 3 4 6; do
    if [ $S = 0 ]; then unset ORBM_MFMA_SPLITS; else export ORBM_MFMA_SPLITS=$S; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3mfma -o q$1s$S -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs > gpurun_out/r3mfma/bench.log 2>&1
    python3 - <<PY
import csv, glob, json
for f in glob.glob("gpurun_out/r3mfma/**/q$1s${S}_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "best2" in r["Name"]:
            print("QB=$1 OCC=$2 S=${S}", r["Name"][:20], r["AverageNs"])
for ln in open("gpurun_out/r3mfma/bench.log"):
    if ln.startswith("{"):
        j = json.loads(ln); print("   ms_per_step", j["ms_per_step"], "matches", j["matches_per_step"])
PY
  done
done
