#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3sg; mkdir -p $O
for G in 4 2; do
cd my-slam_amd && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -Wno-unused-value -DSP_STG=$G -c csrc/orbm_mfma.hip -o build/orbm_mfma.o 2> ../$O/build.err && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/liborbx.so build/*.o && cd .. || exit 1
for S in 2 3; do
  ORBM_MFMA_SP=1 ORBM_MFMA_SPLITS=$S rocprofv3 --kernel-trace --stats --output-format csv -d $O/stG${G}S$S -o bench -- python3 bench.py --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs --steps 30 > $O/bench.json 2> $O/bench.err
  echo "STG=$G S=$S"; grep "best2" $O/stG${G}S$S/bench_kernel_stats.csv | awk -F'",' '{print $NF}' | cut -d, -f1-3
done; done
