#!/bin/bash
# round 3: k_describe after the LDS layout changes -- parity, kernel stats, LDS counters
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3desc
python -m pytest tests/test_extractor_gpu.py tests/test_golden_cpu.py -m gpu -x -q > gpurun_out/r3desc/pytest.log 2>&1 || { tail -30 gpurun_out/r3desc/pytest.log; exit 1; }
tail -2 gpurun_out/r3desc/pytest.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3desc -o stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs > gpurun_out/r3desc/bench_stats.log 2>&1
tail -1 gpurun_out/r3desc/bench_stats.log | cut -c1-400
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VALU --output-format csv -d gpurun_out/r3desc -o lds -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs > gpurun_out/r3desc/bench_lds.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/r3desc/**/stats_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:40], r["Calls"], r["AverageNs"], r["Percentage"])
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/r3desc/**/lds_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:24]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k, v in acc.items():
    if n[k]: print(k, {c: round(x / n[k]) for c, x in v.items()})
PY
