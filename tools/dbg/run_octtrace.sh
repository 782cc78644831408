#!/bin/bash
# GPU call: parity suite + bench stage times with the built library, then k_octree's section timeline (-DOCT_TRACE variant)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/dbg/run_ab.sh || exit 1
export ORBX_LIB=$GRAFT_REPO_ROOT/my-slam_amd/lib/variants/liborbx_octtrace.so
timeout -k 10 120 python3 tools/dbg/oct_trace.py 640 480 1000 | tail -3
timeout -k 10 120 python3 tools/dbg/oct_trace.py 1920 1080 4000 | tail -3
