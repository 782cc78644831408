#!/bin/bash
# GPU call: whole GPU suite on the new bench/shard/cand_cap code + microbenchmark (extended op list) + bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2b; mkdir -p $O
tools/ubench/valu_issue 2000 > $O/valu_issue.json 2> $O/valu_issue.err; echo "ubench rc=$?"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
timeout -k 10 300 python3 bench.py --steps 50 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 3000 $O/bench.json
