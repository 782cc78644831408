#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3td; mkdir -p $O
for d in 0 1 2 3; do
  ORBX_TILE_DBG=$d ORBX_PYRAMID_TILES=${CFG:-2,32,32} rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$d -o bench -- python3 bench.py --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs --steps 30 > $O/bench_$d.json 2> $O/bench_$d.err
  echo "DBG=$d rc=$?"; grep "k_resize_tiles" $O/stats_$d/bench_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
done
