#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3tt; mkdir -p $O
cd my-slam_amd && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -Wno-unused-value -DORBX_TRACE -c csrc/orbx_pyramid.hip -o build/orbx_pyramid.o 2> ../$O/build.err && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/liborbx.so build/*.o && cd .. || exit 1
for cfg in ${CFGS:-2,32,32 3,32,32}; do echo "== $cfg"; ORBX_PYRAMID_TILES=$cfg python3 tools/dbg/tile_trace.py 640 480 1000 64; done
