#!/bin/bash
# Counter calibration + traffic of the bench kernels.  Separate --pmc passes, kernel-trace only.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/traffic
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/traffic -o cal_fetch -- python3 tools/calibrate_traffic.py > gpurun_out/traffic/cal_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/traffic -o cal_write -- python3 tools/calibrate_traffic.py > gpurun_out/traffic/cal_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/traffic -o bench_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs > gpurun_out/traffic/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/traffic -o bench_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs > gpurun_out/traffic/bench_write.log 2>&1
python3 tools/traffic_summary.py gpurun_out/traffic
