#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "0 0" "1 0" "1 2" "0 2" "1 3" "1 4"; do
  set -- $cfg
  python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs --step-graph $1 --subbatches $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('graph=$1 nsub=$2', d['ms_per_step'], d['ms_per_step_gpu'], d['value'], d['matches_per_step'])"
done
