#!/bin/bash
# Build A/B variants of liborbx.so HERE (hipcc cross-compiles; no GPU time spent compiling):
#   tools/dbg/ab_build.sh <name> "<object basename>" "<extra -D flags>"   ->  my-slam_amd/lib/variants/liborbx_<name>.so
# Only the named object is rebuilt with the flags; all others are taken from my-slam_amd/build.
set -e
cd "$(dirname "$0")/../../my-slam_amd"
name=$1; obj=$2; flags=$3
make -s
mkdir -p build/var_$name lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -c csrc/$obj.hip -o build/var_$name/$obj.o
objs=""
for src in csrc/*.hip csrc/*.cc; do b=$(basename $src); case $b in *.hip) b=${b%.hip}.o;; *.cc) b=${b%.cc}.host.o;; esac; o=build/$b; if [ "$b" = "$obj.o" ]; then objs="$objs build/var_$name/$obj.o"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/variants/liborbx_$name.so $objs
echo "built lib/variants/liborbx_$name.so ($flags)"
