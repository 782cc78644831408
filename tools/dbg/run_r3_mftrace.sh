#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for v in mftrace mft_stg2 mft_stg1; do
for S in 3; do echo "== $v S=$S"; ORBX_LIB=$PWD/my-slam_amd/lib/variants/liborbx_$v.so ORBM_MFMA_SPLITS=$S python tools/dbg/mf_trace.py 2>&1 | tail -4; done; done
