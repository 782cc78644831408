#!/usr/bin/env python3
"""Turn the rocprofv3 FETCH_SIZE / WRITE_SIZE passes into profiles/traffic.json (bytes per launch per bench stage)."""
import csv, glob, json, os, sys, collections
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/traffic"
def load(tag):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, tag + "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            acc[(row["Kernel_Name"].split("(")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
    return acc
cal = {}
for tag, ctr in (("cal_fetch", "FETCH_SIZE"), ("cal_write", "WRITE_SIZE")):
    for (k, c), v in load(tag).items():
        if "k_dbg" in k and c == ctr:
            cal[k + ":" + c] = sum(v) / len(v)
GiB_kb = (1 << 30) / 1024.0
f4 = next((v for k, v in cal.items() if "k_dbg_read<unsigned int>" in k and "FETCH" in k), None)
f16 = next((v for k, v in cal.items() if "k_dbg_read<HIP_vector" in k and "FETCH" in k), None) or next((v for k, v in cal.items() if "k_dbg_read<" in k and "unsigned int" not in k and "FETCH" in k), None)
w4 = next((v for k, v in cal.items() if "k_dbg_write" in k and "WRITE" in k), None)
out = {"calibration": {"bytes_streamed": 1 << 30, "FETCH_SIZE_4B_loads_KB": f4, "FETCH_SIZE_16B_loads_KB": f16, "WRITE_SIZE_4B_stores_KB": w4,
                       "fetch_factor_4B": (GiB_kb / f4) if f4 else None, "fetch_factor_16B": (GiB_kb / f16) if f16 else None,
                       "write_factor_4B": (GiB_kb / w4) if w4 else None}}
ff = out["calibration"]["fetch_factor_4B"] or 1.0
wf = out["calibration"]["write_factor_4B"] or 1.0
fetch, write = load("bench_fetch"), load("bench_write")
stage = {"pyramid": "k_resize", "fast": "k_fast_cells", "quadtree": "k_octree", "describe": "k_describe", "match": "k_best2_"}
for st, pat in stage.items():
    fs = [sum(v) / len(v) * (1 if "resize" not in k else 1) for (k, c), v in fetch.items() if pat in k and c == "FETCH_SIZE"]
    ws = [sum(v) / len(v) for (k, c), v in write.items() if pat in k and c == "WRITE_SIZE"]
    ncalls = {k: len(v) for (k, c), v in fetch.items() if pat in k}
    if st == "pyramid":   # several launches per step: sum over the distinct kernels weighted by calls per step
        tot_f = sum(sum(v) for (k, c), v in fetch.items() if pat in k and c == "FETCH_SIZE")
        tot_w = sum(sum(v) for (k, c), v in write.items() if pat in k and c == "WRITE_SIZE")
        steps = max(1, max(len(v) for (k, c), v in fetch.items() if "k_fast_cells" in k))
        f_kb, w_kb = tot_f / steps, tot_w / steps
    else:
        f_kb, w_kb = (sum(fs) if fs else 0.0), (sum(ws) if ws else 0.0)
    out[st] = int((f_kb * ff + w_kb * wf) * 1024)
    out[st + "_detail"] = {"FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb, "fetch_factor": ff, "write_factor": wf}
json.dump(out, open(os.path.join(d, "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
