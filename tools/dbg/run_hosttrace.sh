#!/bin/bash
# GPU call: host-buffer batch path with its host-side time split (ORBX_BATCH_TRACE): compute streams x chunk sizes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "2 16" "3 16" "3 8" "1 16" "3 12"; do
  set -- $cfg
  echo "== streams $1 chunk $2"
  ORBX_BATCH_STREAMS=$1 ORBX_BATCH_TRACE=1 ORBX_BENCH_CHUNK=$2 timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pipelined --no-extra-configs 2> gpurun_out/hosttrace.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['host_api']['ms_per_batch'])"
  grep orbx_extract_batch gpurun_out/hosttrace.err | tail -2
done
