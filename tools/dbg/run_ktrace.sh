#!/bin/bash
# GPU call: per-dispatch kernel trace of the bench step; prints mean duration per (kernel, grid) -- the resize levels separately
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/kt; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-api --no-pipelined --no-extra-configs > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 - <<PY
import csv,glob,collections
f=glob.glob("$O/**/*kernel_trace.csv",recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"].split("(")[0][:40]
    d[(n,r["Grid_Size_X"],r["Grid_Size_Z"])].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    if len(v)>=20: print("%-42s grid %8s z %3s  n=%4d  mean %8.2f us  min %8.2f"%(k[0],k[1],k[2],len(v),sum(v)/len(v)/1e3,min(v)/1e3))
PY
rm -f $O/*/*.csv $O/*.csv 2>/dev/null; true
