#!/bin/bash
# round 3: tile-fused upper pyramid -- parity with the tiles forced on for every batch size, then per-kernel times per variant
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3tl; mkdir -p $O
for cfg in "2,32,32,1" "1,32,32,1" "3,24,24,1"; do
  ORBX_PYRAMID_TILES=$cfg timeout -k 10 600 python3 -m pytest tests/test_extractor_gpu.py -m gpu -x -q > $O/pytest_$cfg.log 2>&1; rc=$?; echo "tiles=$cfg"; tail -3 $O/pytest_$cfg.log
  [ $rc -ne 0 ] && exit $rc
done
for cfg in ${CFGS:-0 2,32,32 1,32,32 3,32,32 2,24,24 2,32,16 2,48,32}; do
  ORBX_PYRAMID_TILES=$cfg rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$cfg -o bench -- python3 bench.py --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs > $O/bench_$cfg.json 2> $O/bench_$cfg.err
  echo "TILES=$cfg rc=$?"; python3 - <<PY
import csv,glob,json
f=glob.glob("$O/stats_$cfg/**/*kernel_stats.csv",recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "resize" in n:
        print("  %-40s calls %4s avg %8.1f us  total/step %.1f"%(n[:40],r["Calls"],float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/120e3)); tot+=float(r["TotalDurationNs"])/120e3
j=json.loads(open("$O/bench_$cfg.json").read().strip().splitlines()[-1])
print("  pyramid kernels per step %.1f us; ms_per_step %.4f"%(tot, j["ms_per_step"]))
PY
done
