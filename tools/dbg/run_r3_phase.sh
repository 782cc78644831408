#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3ph; mkdir -p $O
cd my-slam_amd && for f in orbx_fast orbx_describe; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -Wno-unused-value -DORBX_TRACE -c csrc/$f.hip -o build/$f.o 2>> ../$O/build.err; done; /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/liborbx.so build/*.o && cd .. || exit 1
python3 tools/dbg/phase_trace.py 640 480 1000 64
