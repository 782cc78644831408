#!/bin/bash
# usage: fast_variant.sh "<extra -D flags>" ...   -- rebuilds orbx_fast.o with each flag set and prints the bench stage times
cd "$GRAFT_REPO_ROOT/my-slam_amd"
for a in "$@"; do
  rm -f build/orbx_fast.o
  make -s HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $a" > /dev/null 2>&1
  echo "flags: $a"; (cd .. && python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-match 2>/dev/null | grep -o '"stage_ms[^}]*}')
done
