#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (gpurun_out/pmc/<tag>_*_counter_collection.csv): per kernel, mean per dispatch."""
import csv, glob, sys, collections, os
tag = sys.argv[1] if len(sys.argv) > 1 else "pmc"
d = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/pmc"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, tag + "_*counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:40]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    if not k.startswith("k_") and "k_" not in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-24s n=%-4d mean=%14.1f" % (c, len(v), sum(v) / len(v)))
