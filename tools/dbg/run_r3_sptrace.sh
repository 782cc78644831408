#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3st; mkdir -p $O
cd my-slam_amd && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -Wno-unused-value -DSP_TRACE -c csrc/orbm_mfma.hip -o build/orbm_mfma.o 2> ../$O/build.err && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/liborbx.so build/*.o && cd .. || exit 1
for S in ${SPLITS:-1 2 3}; do echo "== S=$S"; ORBM_MFMA_SP=${SP:-1} ORBM_MFMA_SPLITS=$S rocprofv3 --kernel-trace --stats --output-format csv -d $O/st$S -o t -- python3 tools/dbg/sp_trace.py; grep "best2" $O/st$S/t_kernel_stats.csv | cut -d, -f1-8 | cut -c1-40,150-; done
