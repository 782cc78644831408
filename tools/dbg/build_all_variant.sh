#!/bin/bash
# Build a variant of liborbx.so with extra flags on EVERY device object: tools/dbg/build_all_variant.sh <name> "<flags>"
set -e
cd "$(dirname "$0")/../../my-slam_amd"
name=$1; flags=$2
make -s
mkdir -p build/var_$name lib/variants
objs=""
for src in csrc/*.hip; do b=$(basename $src .hip); ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -c $src -o build/var_$name/$b.o ) & objs="$objs build/var_$name/$b.o"; while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.5; done; done
wait
for src in csrc/*.cc; do objs="$objs build/$(basename $src .cc).host.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/variants/liborbx_$name.so $objs
echo "built lib/variants/liborbx_$name.so ($flags)"
