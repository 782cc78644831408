#!/usr/bin/env python3
"""Exhaustive check of the device cos/sin routine (sincos_cr in csrc/orbx_describe.hip) against the
canonical definition "correctly rounded fp32 cos/sin of the fp32 angle" (DESIGN.md §2), evaluated on
the host with the x87 80-bit cosl/sinl of the oracle library, over EVERY fp32 value in [0, 2*pi]
(the domain of kpt.angle * factorPI, src/ORBextractor.cc:114).  Run on the GPU box:
    python tests/tools/verify_sincos.py [--stride N] [--out profiles/sincos_exhaustive.json]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: F401,E402
import oracle_lib as O  # noqa: E402
import my_slam_amd as M  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stride", type=int, default=1, help="check every N-th bit pattern (1 = exhaustive)")
    ap.add_argument("--chunk", type=int, default=1 << 24)
    ap.add_argument("--threads", type=int, default=min(16, os.cpu_count() or 1))
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    L = O.lib()
    L.oro_sincos_rad_array.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong]
    L.oro_sincos_rad_array.restype = None
    top = int(np.float32(2 * np.pi).view(np.uint32)) + 2      # a little past RN(360 * factorPI)
    t0 = time.time()
    total = bad_c = bad_s = 0
    examples = []
    pool = ThreadPoolExecutor(args.threads)

    def ref(theta):
        c = np.empty_like(theta); s = np.empty_like(theta)
        n = len(theta)
        parts = np.linspace(0, n, args.threads + 1).astype(np.int64)
        futs = [pool.submit(L.oro_sincos_rad_array, theta[a:b].ctypes.data, c[a:b].ctypes.data, s[a:b].ctypes.data, int(b - a))
                for a, b in zip(parts[:-1], parts[1:]) if b > a]
        for f in futs:
            f.result()
        return c, s

    for lo in range(0, top, args.chunk * args.stride):
        bits = np.arange(lo, min(top, lo + args.chunk * args.stride), args.stride, dtype=np.uint32)
        theta = bits.view(np.float32)
        gc, gs = M.debug_sincos(theta)
        rc, rs = ref(theta)
        mc = np.nonzero(gc.view(np.uint32) != rc.view(np.uint32))[0]
        ms = np.nonzero(gs.view(np.uint32) != rs.view(np.uint32))[0]
        total += len(bits); bad_c += len(mc); bad_s += len(ms)
        for i in list(mc[:20]) + list(ms[:20]):
            if len(examples) < 64:
                examples.append({"theta_bits": int(bits[i]), "theta": float(theta[i]), "gpu_cos": float(gc[i]), "ref_cos": float(rc[i]),
                                 "gpu_sin": float(gs[i]), "ref_sin": float(rs[i])})
        if (lo // (args.chunk * args.stride)) % 8 == 0:
            print("  %.1f %%  checked %d  cos mismatches %d  sin mismatches %d  (%.0f s)" % (100.0 * lo / top, total, bad_c, bad_s, time.time() - t0), flush=True)
    res = {"checked": total, "stride": args.stride, "domain": "fp32 bit patterns 0 .. %d (theta in [0, 2pi])" % top,
           "cos_mismatches": bad_c, "sin_mismatches": bad_s, "examples": examples, "seconds": round(time.time() - t0, 1)}
    print(json.dumps({k: v for k, v in res.items() if k != "examples"}))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
