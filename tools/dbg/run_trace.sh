#!/bin/bash
# In-kernel timing builds (on the GPU box; the box's copy of the tree is scratch): rebuild one file with its trace macro and run the
# matching reader.  usage: tools/dbg/run_trace.sh phase|tiles|mfma_sp [reader args]
#   phase    -DORBX_TRACE  k_fast_cells / k_describe phase split            (phase_trace.py W H nfeatures batch)
#   tiles    -DORBX_TRACE  k_resize_tiles per-level split + wave residency  (tile_trace.py W H nfeatures batch; ORBX_PYRAMID_TILES selects the plan)
#   mfma_sp  -DSP_TRACE    k_best2_mfma_sp phase split + workgroup residency (sp_trace.py; ORBM_MFMA_SP=1, ORBM_MFMA_SPLITS)
set -e
which=${1:?phase|tiles|mfma_sp}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -Wno-unused-value"
cd my-slam_amd
case $which in
  phase)   for f in orbx_fast orbx_describe; do /opt/rocm/bin/hipcc $FL -DORBX_TRACE -c csrc/$f.hip -o build/$f.o; done ;;
  tiles)   /opt/rocm/bin/hipcc $FL -DORBX_TRACE -c csrc/orbx_pyramid.hip -o build/orbx_pyramid.o ;;
  mfma_sp) /opt/rocm/bin/hipcc $FL -DSP_TRACE -c csrc/orbm_mfma.hip -o build/orbm_mfma.o ;;
esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/liborbx.so build/*.o
cd ..
case $which in
  phase)   python3 tools/dbg/phase_trace.py ${@:-640 480 1000 64} ;;
  tiles)   ORBX_PYRAMID_TILES=${ORBX_PYRAMID_TILES:-2,32,32} python3 tools/dbg/tile_trace.py ${@:-640 480 1000 64} ;;
  mfma_sp) ORBM_MFMA_SP=1 python3 tools/dbg/sp_trace.py ;;
esac
