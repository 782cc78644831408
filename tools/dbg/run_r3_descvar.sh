#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT/my-slam_amd"
for a in "-DDESC_RAW_STRIDE=48 -DDESC_NO_B128 -DDESC_NO_B64" "-DDESC_RAW_STRIDE=44 -DDESC_NO_B128 -DDESC_NO_B64" "-DDESC_RAW_STRIDE=48 -DDESC_NO_B64" "-DDESC_RAW_STRIDE=48 -DDESC_NO_B128"; do
  rm -f build/orbx_describe.o
  make -s HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $a" 2>&1 | grep -E "error|Error" | head -5
  echo "flags: $a  lib $(md5sum lib/liborbx.so | cut -c1-8)"
  (cd .. && python -m pytest tests/test_extractor_gpu.py -m gpu -x -q -k "matches_oracle" 2>&1 | tail -1)
done
