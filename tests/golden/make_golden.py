#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the oracle (CPU restatement) on seeded synthetic frames.

PARITY UNPINNED: the reference ships no fixtures and cannot be built here (no OpenCV 3.1.0), so these
vectors freeze the ORACLE's behaviour (regression pin + GPU-box fixture), not the reference's.
Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: F401  (registers my_slam_amd)
import oracle_lib as O
import my_slam_amd.synth as synth

CASES = [  # name, seed, W, H, nfeatures, blur_mode
    ("c1_640x480_n1000", 1, 640, 480, 1000, 0),
    ("c5_1241x376_n2000", 5, 1241, 376, 2000, 0),
    ("small_322x241_n500", 9, 322, 241, 500, 0),
    ("c1_640x480_n1000_x86blur", 1, 640, 480, 1000, 1),
]


def main():
    for name, seed, W, H, n, blur in CASES:
        img = synth.texture(seed, W, H)
        ex = O.Extractor(n, blur_mode=blur)
        kps, desc, npl, levels = ex.extract(img, want_levels=True)
        sha = np.array([hashlib.sha256(l.tobytes()).hexdigest() for l in levels])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), seed=seed, W=W, H=H, nfeatures=n, blur_mode=blur,
                            image_sha256=hashlib.sha256(img.tobytes()).hexdigest(),
                            keypoints=kps, descriptors=desc, n_per_level=np.array(npl), level_sha256=sha)
        print(name, len(kps), npl)
    # matcher vector: config-3-like pair at reduced size
    f0, f1 = synth.frame_pair(2, 960, 540)
    ex = O.Extractor(2000)
    k0, d0, _ = ex.extract(f0)
    k1, d1, _ = ex.extract(f1)
    nm, m12 = O.match_dense(d1, k1["angle"], d0, k0["angle"], 50, 0.9, True)
    bi, bd, sd = O.best2(d1, d0)
    np.savez_compressed(os.path.join(HERE, "match_960x540_n2000.npz"), q=d1, t=d0, angle_q=k1["angle"], angle_t=k0["angle"],
                        best_idx=bi, best_d=bd, second_d=sd, match12=m12, nmatches=nm)
    print("match", nm)


if __name__ == "__main__":
    main()
