#!/bin/bash
# GPU suite on the tree as built, then the bench step with the kernel arguments in host memory (0) / device memory (1)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3k; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -3 $O/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
B="--no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs --steps 200"
for v in 0 1 0 1; do
  HIP_FORCE_DEV_KERNARG=$v timeout -k 10 200 python3 bench.py $B > $O/bench_ka$v.json 2> $O/bench_ka$v.err || exit 1
  python3 - <<PY
import json
d=json.loads(open("$O/bench_ka$v.json").read().strip().splitlines()[-1])
print("KERNARG=$v", d["ms_per_step"], d["ms_per_step_gpu"], d["roofline"]["stage_ms"])
PY
done
timeout -k 10 200 python3 bench.py $B > $O/bench_default.json 2> $O/bench_default.err
python3 -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('default', d['ms_per_step'], d['ms_per_step_gpu'], d['roofline']['stage_ms'])"
bash tools/prof_mem.sh r3mem > $O/mem.log 2>&1; python3 tools/pmc_summary.py r3mem > $O/mem_summary.txt 2>&1; grep -c mean $O/mem_summary.txt
