#!/usr/bin/env python3
"""Randomised GPU-vs-oracle sweep of ORBmatcher::SearchByBoW (k_bow_select): random vocabularies, node granularity, feature
counts, ratios, MapPoint masks and image pairs; match table and return value must be identical.  usage: stress_bow.py [seconds] [seed]"""
import os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import conftest  # noqa
import oracle_lib as O
import my_slam_amd as M
import my_slam_amd.synth as synth
from test_vocabulary import make_vocabulary

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
tmp = tempfile.mkdtemp()
t0 = time.time(); n_ok = 0; nm_tot = 0; big = 0
while time.time() - t0 < budget:
    k = int(rng.integers(2, 11)); L = int(rng.integers(1, 5)); levelsup = int(rng.integers(0, L + 2))
    path = os.path.join(tmp, "v.txt")
    make_vocabulary(path, k, L, seed=int(rng.integers(1, 1 << 30)))
    v = M.ORBVocabulary(path)
    W = int(rng.integers(200, 1300)); H = int(rng.integers(160, 720)); nf = int(rng.choice([50, 300, 1000, 2000, 4000]))
    shift = (int(rng.integers(0, 12)), int(rng.integers(0, 8)))
    f0, f1 = synth.frame_pair(int(rng.integers(1, 1 << 30)), W, H, shift=shift)
    try:
        ex = M.ORBextractor(nf, max_width=W, max_height=H)
        k0, d0 = ex(f0); k1, d1 = ex(f1)
    except M.OrbxError:
        continue
    if len(d0) == 0 or len(d1) == 0:
        continue
    _, fv0 = v.transform(d0, levelsup); _, fv1 = v.transform(d1, levelsup)
    ratio = float(rng.choice([0.6, 0.7, 0.75, 0.9, 1.0])); ori = bool(rng.integers(0, 2))
    valid = (rng.random(len(d0)) < rng.uniform(0.2, 1.0)).astype(np.uint8) if rng.integers(0, 2) else None
    m = M.ORBmatcher(ratio, ori, max_queries=8192, max_train=8192, max_pairs=1 << 22)
    mf, nm = m.SearchByBoW(k0, d0, fv0, k1, d1, fv1, valid)
    omf, onm = O.search_by_bow(d0, k0["angle"], fv0, d1, k1["angle"], fv1, ratio, ori, valid)
    if nm != onm or not np.array_equal(mf, omf):
        print("MISMATCH", dict(k=k, L=L, levelsup=levelsup, W=W, H=H, nf=nf, ratio=ratio, ori=ori, valid=valid is not None, nm=nm, onm=onm))
        sys.exit(1)
    n_ok += 1; nm_tot += nm
    big += int(max(np.diff(fv1[1])) > 256) if len(fv1[1]) > 1 else 0
print("stress_bow: %d random cases identical to the oracle in %.0f s (%d matches in total, %d cases with a node of more than 256 frame features)"
      % (n_ok, time.time() - t0, nm_tot, big))
