#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2h; mkdir -p $O
cat > /tmp/hb.py <<'PY'
import sys, os, time
sys.path.insert(0, "tests"); import conftest
import numpy as np
import my_slam_amd as M, my_slam_amd.synth as synth
fr = synth.stream(4, 640, 480, 64)
ex = M.ORBextractor(1000, max_width=640, max_height=480, max_batch=64)
ex.set_batch_chunk(int(sys.argv[1]))
for _ in range(6): ex.extract_batch_raw(fr)
PY
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/prof -o c16 -- python3 /tmp/hb.py 16 > /dev/null 2> $O/c16.err
ls $O/prof
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob('gpurun_out/r2h/prof/c16_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:28], r.get('Stream_Id', r.get('Queue_Id', '?'))))
for f in glob.glob('gpurun_out/r2h/prof/c16_memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', r.get('Name', ''))[:24], '-'))
rows.sort()
# last call = last 4*10 kernels; print the tail of the timeline relative to its first H2D
tail = rows[-70:]
t0 = tail[0][0]
for s, e, n, q in tail:
    print("%9.1f %9.1f us  %-30s q=%s" % ((s - t0) / 1e3, (e - t0) / 1e3, n, q))
PY
