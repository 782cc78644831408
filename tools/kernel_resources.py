#!/usr/bin/env python3
"""Print VGPR/SGPR/scratch/LDS/occupancy per kernel of the HIP sources (hipcc -Rpass-analysis)."""
import re, subprocess, sys, glob, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
srcs = sys.argv[1:] or sorted(glob.glob(os.path.join(root, "my-slam_amd/csrc/*.hip")))
for src in srcs:
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                          "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                         capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
        cur[k] = v
        if k.startswith("LDS"):
            print("%-60s sgpr %-4s vgpr %-4s scratch %-4s occ %-3s lds %s" % (
                cur["name"][:60], cur.get("TotalSGPRs"), cur.get("VGPRs"), cur.get("ScratchSize [bytes/lane]"),
                cur.get("Occupancy [waves/SIMD]"), v))
