#!/bin/bash
# GPU call: chunked host batches (parity + timing), MFMA split sweep
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2f; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_extractor_gpu.py tests/test_matcher_gpu.py tests/test_threads_gpu.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.log
python3 - > $O/host_api.txt 2>&1 <<'PY'
import sys, os, time, json
sys.path.insert(0, "tests"); import conftest
import numpy as np, torch
import my_slam_amd as M, my_slam_amd.synth as synth
fr = synth.stream(4, 640, 480, 64)
pin = torch.from_numpy(fr.copy()).pin_memory().numpy()
for chunk in (0, 8, 16, 32):
    ex = M.ORBextractor(1000, max_width=640, max_height=480, max_batch=64)
    ex.set_batch_chunk(chunk)
    for name, a in (("pageable", fr), ("pinned", pin)):
        for _ in range(4): ex.extract_batch_raw(a)
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); ex.extract_batch_raw(a); ts.append(time.perf_counter() - t0)
        print("chunk %2d %-8s median %.3f ms  min %.3f ms" % (chunk, name, 1e3 * float(np.median(ts)), 1e3 * min(ts)))
PY
cat $O/host_api.txt
for S in 1 2 3 4 6; do
  ORBM_MFMA_SPLITS=$S rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o s$S -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs > /dev/null 2> $O/bench_s$S.err
  python3 - $S <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/r2f/prof/s%s_kernel_stats.csv' % sys.argv[1])[0]
for r in csv.DictReader(open(f)):
    if r['Name'].startswith(('k_best2_mfma','k_accept_rot','k_expand','k_merge')): print("S=%s %-16s avg_us %8.2f" % (sys.argv[1], r['Name'][:14], float(r['AverageNs'])/1e3))
PY
done
