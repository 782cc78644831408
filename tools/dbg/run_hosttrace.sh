#!/bin/bash
# GPU call: host-buffer batch path: parity of the page-locked upload, then its time split (ORBX_BATCH_TRACE)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python3 -m pytest tests/test_extractor_gpu.py -m gpu -x -q -k "batch" 2>&1 | tail -3
ORBX_BATCH_TRACE=1 timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pipelined --no-extra-configs 2> gpurun_out/hosttrace.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['host_api'])"
grep orbx_extract_batch gpurun_out/hosttrace.err | sed -n '10,12p;24,27p'
