#!/bin/bash
# usage: build_variant.sh <object basename> "<extra -D flags>" ... -- rebuilds one object with each flag set and prints bench stage times
cd "$GRAFT_REPO_ROOT/my-slam_amd"
obj=$1; shift
for a in "$@"; do
  rm -f build/$obj.o
  make -s HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $a" > /dev/null 2>&1
  echo "flags: $a"; (cd .. && python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-match 2>/dev/null | grep -o '"stage_ms[^}]*}'; python tests/tools/bench_extra.py 2>/dev/null | grep -A6 '"device_640x480_n1000_b1"' | tr -d "\n ")
  echo
done
