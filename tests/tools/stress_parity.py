#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep (run on the GPU box): random sizes, feature counts, thresholds, scale factors and
seeds; every keypoint field and descriptor byte must agree.  usage: stress_parity.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import conftest  # noqa
import oracle_lib as O
import my_slam_amd as M
import my_slam_amd.synth as synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); n_ok = 0; n_rej = 0; reasons = {}
while time.time() - t0 < budget:
    W = int(rng.integers(64, 1400)); H = int(rng.integers(64, 1000))
    nf = int(rng.choice([1, 37, 200, 500, 1000, 2000, 4000]))
    sf = float(rng.choice([1.1, 1.2, 1.2, 1.2, 1.33, 1.5, 2.0])); nl = int(rng.integers(1, 10))
    ini = int(rng.integers(5, 40)); mn = int(rng.integers(2, 25))
    seed = int(rng.integers(1, 1 << 30))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        img = synth.texture(seed, W, H)
    elif kind == 1:   # uniform noise
        img = np.random.default_rng(seed).integers(0, 256, (H, W), dtype=np.uint8)
    elif kind == 2:   # low-contrast texture (threshold fallback everywhere)
        img = (synth.texture(seed, W, H).astype(np.int32) // 6 + 100).astype(np.uint8)
    else:             # texture with a strided (non-contiguous) view
        big = synth.texture(seed, W + 7, H)
        img = big[:, 3:3 + W]
    try:
        ex = M.ORBextractor(nf, sf, nl, ini, mn, max_width=W, max_height=H)
    except Exception as e:     # shapes the library rejects at create time (e.g. LDS limit)
        n_rej += 1; k = str(e)[:60]; reasons[k] = reasons.get(k, 0) + 1; continue
    try:
        kps, desc = ex(img)
    except Exception as e:
        # the oracle must reject the same shape (portrait levels etc.)
        n_rej += 1; k = str(e).split(":")[-1][:70]; reasons[k] = reasons.get(k, 0) + 1; del ex; continue
    okps, odesc, _ = O.Extractor(nf, sf, nl, ini, mn).extract(np.ascontiguousarray(img))
    tag = "W=%d H=%d nf=%d sf=%g nl=%d th=%d/%d seed=%d kind=%d" % (W, H, nf, sf, nl, ini, mn, seed, kind)
    assert len(kps) == len(okps), "count %d vs %d: %s" % (len(kps), len(okps), tag)
    for name in ("x", "y", "size", "angle", "response", "octave"):
        a, b = kps[name], okps[name]
        assert np.array_equal(a.view(np.uint32) if a.dtype.kind == "f" else a, b.view(np.uint32) if b.dtype.kind == "f" else b), "%s: %s" % (name, tag)
    assert np.array_equal(desc, odesc), "descriptors: " + tag
    if n_ok % 3 == 0:     # the same handle again: the second and third call of a shape capture and replay the HIP graph
        for rep in range(2):
            k2, d2 = ex(img)
            assert k2.tobytes() == kps.tobytes() and np.array_equal(d2, desc), "graph replay %d: %s" % (rep, tag)
    if n_ok % 7 == 0:     # the batched path (other grids, XCD-aware order over several frames) must give the same frames
        B = int(rng.integers(2, 6))
        stack = np.stack([np.ascontiguousarray(img), np.ascontiguousarray(img[::-1, ::-1]), np.ascontiguousarray(img)] + [np.ascontiguousarray(img)] * (B - 3))[:B]
        exb = M.ORBextractor(nf, sf, nl, ini, mn, max_width=W, max_height=H, max_batch=B)
        res = exb.extract_batch(stack)
        for b in [0] + ([B - 1] if B > 2 else []):       # frames 0, 2, 3.. are the image itself; frame 1 is flipped
            assert len(res[b][0]) == len(kps) and np.array_equal(res[b][1], desc), "batch frame %d: %s" % (b, tag)
            assert np.array_equal(res[b][0]["x"], kps["x"]) and np.array_equal(res[b][0]["angle"].view(np.uint32), kps["angle"].view(np.uint32)), "batch kps: " + tag
        if B > 1:
            ok2, od2, _ = O.Extractor(nf, sf, nl, ini, mn).extract(stack[1])
            assert len(res[1][0]) == len(ok2) and np.array_equal(res[1][1], od2), "batch frame 1 (flipped): " + tag
        del exb
    n_ok += 1; del ex
    if n_ok % 20 == 0:
        print("%d cases ok (%d rejected shapes), %.0f s" % (n_ok, n_rej, time.time() - t0), flush=True)
print("PASS: %d random cases bit-exact, %d shapes rejected by the library" % (n_ok, n_rej))
for k, v in sorted(reasons.items(), key=lambda kv: -kv[1]):
    print("   %5d  %s" % (v, k))
