"""N2: DBoW2 vocabulary tree.  The reference ships no vocabulary (ORBvoc.txt is a missing large
blob), so a synthetic one is written in the text format of TemplatedVocabulary::loadFromTextFile.
CPU: oracle restatement by hand.  GPU: loader + descent + BowVector/FeatureVector == oracle; SearchByBoW-style
bucket matching through orbm_best2 == brute force per shared node."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O


def make_vocabulary(path, k=8, depth=3, seed=0, scoring=0, weighting=0):
    """Random complete k-ary tree; returns the arrays the oracle needs (parsed independently of the product loader)."""
    rng = np.random.default_rng(seed)
    parent, leaf, desc, weight = [], [], [], []
    frontier = [0]
    nid = 1
    lines = ["%d %d %d %d" % (k, depth, scoring, weighting)]
    for level in range(1, depth + 1):
        nxt = []
        for p in frontier:
            base = rng.integers(0, 256, 32, dtype=np.uint8)
            for c in range(k):
                d = base.copy()
                flips = rng.integers(0, 256, 40 // level)
                for f in flips:
                    d[f >> 3] ^= np.uint8(1 << (f & 7))
                is_leaf = level == depth
                w = float(np.round(rng.uniform(0.5, 9.0), 6)) if is_leaf else 0.0
                if is_leaf and rng.uniform() < 0.03:
                    w = 0.0                                   # a stopped word
                lines.append("%d %d %s %.6f" % (p, 1 if is_leaf else 0, " ".join(str(int(x)) for x in d), w))
                parent.append(p); leaf.append(is_leaf); desc.append(d); weight.append(w)
                nxt.append(nid); nid += 1
        frontier = nxt
    open(path, "w").write("\n".join(lines) + "\n")
    nn = nid
    children = [[] for _ in range(nn)]
    for i, p in enumerate(parent):
        children[p].append(i + 1)
    off = np.zeros(nn + 1, np.int32)
    for i in range(nn):
        off[i + 1] = off[i] + len(children[i])
    ids = np.array([c for ch in children for c in ch], np.int32)
    D = np.zeros((nn, 32), np.uint8); D[1:] = np.array(desc)
    W = np.zeros(nn, np.float64); W[1:] = weight
    word_of = np.zeros(nn, np.int32)
    wid = 0
    for i in range(1, nn):
        if leaf[i - 1]:
            word_of[i] = wid; wid += 1
    return dict(k=k, L=depth, scoring=scoring, weighting=weighting, nnodes=nn, child_off=off, child_ids=ids, desc=D, word_of=word_of,
                weight=W, nwords=wid)


class OroVoc(C.Structure):
    _fields_ = [("k", C.c_int), ("L", C.c_int), ("scoring", C.c_int), ("weighting", C.c_int), ("nnodes", C.c_int),
                ("child_off", C.c_void_p), ("child_ids", C.c_void_p), ("desc", C.c_void_p), ("word_of", C.c_void_p), ("weight", C.c_void_p)]


def oracle_transform(voc, feat, levelsup):
    L = O.lib()
    L.oro_voc_transform_features.restype = None
    L.oro_voc_score_l1.restype = C.c_double
    v = OroVoc(voc["k"], voc["L"], voc["scoring"], voc["weighting"], voc["nnodes"], voc["child_off"].ctypes.data,
               voc["child_ids"].ctypes.data, voc["desc"].ctypes.data, voc["word_of"].ctypes.data, voc["weight"].ctypes.data)
    feat = np.ascontiguousarray(feat, np.uint8)
    n = len(feat)
    w = np.zeros(n, np.int32); nd = np.zeros(n, np.int32); wt = np.zeros(n, np.float64)
    L.oro_voc_transform_features(C.byref(v), C.c_void_p(feat.ctypes.data), n, levelsup, C.c_void_p(w.ctypes.data), C.c_void_p(nd.ctypes.data), C.c_void_p(wt.ctypes.data))
    ids = np.zeros(max(n, 1), np.int32); vals = np.zeros(max(n, 1), np.float64)
    nb = L.oro_voc_bow_vector(C.byref(v), C.c_void_p(w.ctypes.data), C.c_void_p(wt.ctypes.data), n, C.c_void_p(ids.ctypes.data), C.c_void_p(vals.ctypes.data))
    nids = np.zeros(max(n, 1), np.int32); off = np.zeros(n + 1, np.int32); idx = np.zeros(max(n, 1), np.int32)
    nn = L.oro_voc_feature_vector(C.c_void_p(nd.ctypes.data), C.c_void_p(wt.ctypes.data), n, C.c_void_p(nids.ctypes.data), C.c_void_p(off.ctypes.data), C.c_void_p(idx.ctypes.data))
    return (w, nd, wt), (ids[:nb].copy(), vals[:nb].copy()), (nids[:nn].copy(), off[:nn + 1].copy(), idx[:off[nn]].copy())


def test_oracle_descent_by_hand(tmp_path):
    voc = make_vocabulary(str(tmp_path / "v.txt"), k=3, depth=2, seed=1)
    # a feature equal to a leaf descriptor lands in that leaf (distance 0 at the last level) if its parent is the
    # nearest level-1 node; the node id at levelsup=1 is that parent
    leaf = 3 + 1 + 4                       # some level-2 node
    parent = int(np.searchsorted(voc["child_off"], np.nonzero(voc["child_ids"] == leaf)[0][0], side="right") - 1)
    feat = voc["desc"][leaf:leaf + 1]
    (w, nd, wt), bow, fv = oracle_transform(voc, feat, 1)
    d_par = [int(np.unpackbits(feat[0] ^ voc["desc"][c]).sum()) for c in (1, 2, 3)]
    if int(np.argmin(d_par)) + 1 == parent:
        assert w[0] == voc["word_of"][leaf] and nd[0] == parent and wt[0] == voc["weight"][leaf]
    (w0, nd0, _), _, _ = oracle_transform(voc, feat, 5)
    assert nd0[0] == 0                      # levelsup >= L: the root
    # BowVector is L1-normalised, FeatureVector lists features per node in order
    feats = np.repeat(voc["desc"][4:10], 2, axis=0)
    (w, nd, wt), (ids, vals), (nids, off, idx) = oracle_transform(voc, feats, 1)
    assert abs(vals.sum() - 1.0) < 1e-12 and (np.diff(ids) > 0).all() and (np.diff(nids) > 0).all()
    for j in range(len(nids)):
        seg = idx[off[j]:off[j + 1]]
        assert (np.diff(seg) > 0).all() and (nd[seg] == nids[j]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("k,depth,scoring,weighting,levelsup", [(10, 3, 0, 0, 2), (8, 4, 0, 0, 4), (6, 3, 1, 1, 1), (5, 3, 0, 3, 2), (4, 2, 5, 1, 1)])
def test_hip_vocabulary_equals_oracle(orbx, synth, tmp_path, k, depth, scoring, weighting, levelsup):
    path = str(tmp_path / "voc.txt")
    voc = make_vocabulary(path, k, depth, seed=k * 10 + depth, scoring=scoring, weighting=weighting)
    v = orbx.ORBVocabulary(path)
    assert (v.k, v.depth, v.nnodes, v.nwords, v.scoring, v.weighting) == (k, depth, voc["nnodes"], voc["nwords"], scoring, weighting)
    img = synth.texture(3, 640, 480)
    kps, desc = orbx.ORBextractor(1000, max_width=640, max_height=480)(img)
    w, nd, wt = v.transform_features(desc, levelsup)
    (ow, ond, owt), obow, ofv = oracle_transform(voc, desc, levelsup)
    assert np.array_equal(w, ow) and np.array_equal(nd, ond) and np.array_equal(wt.view(np.uint64), owt.view(np.uint64))
    bow, fv = v.transform(desc, levelsup)
    assert np.array_equal(bow[0], obow[0]) and np.array_equal(bow[1].view(np.uint64), obow[1].view(np.uint64))
    assert all(np.array_equal(a, b) for a, b in zip(fv, ofv))
    O.lib().oro_voc_score_l1.restype = C.c_double
    assert abs(v.score(bow, bow) - 1.0) < 1e-12 or scoring != 0


@pytest.mark.gpu
def test_search_by_bow_buckets(orbx, synth, tmp_path):
    """SearchByBoW's merge-join (src/ORBmatcher.cc:180-264) on two frames: per shared node, best/second-best of every
    feature of frame A over frame B's features in that node == orbm_best2 on the CSR lists."""
    path = str(tmp_path / "voc.txt")
    make_vocabulary(path, 10, 3, seed=5)
    v = orbx.ORBVocabulary(path)
    f0, f1 = synth.frame_pair(2, 640, 480)
    ex = orbx.ORBextractor(1000, max_width=640, max_height=480)
    k0, d0 = ex(f0); k1, d1 = ex(f1)
    _, (n0, o0, i0) = v.transform(d0, 2)
    _, (n1, o1, i1) = v.transform(d1, 2)
    q_idx, off, idx = [], [0], []
    a = b = 0
    while a < len(n0) and b < len(n1):            # merge-join on node id
        if n0[a] == n1[b]:
            for qa in i0[o0[a]:o0[a + 1]]:
                q_idx.append(int(qa)); idx.extend(i1[o1[b]:o1[b + 1]].tolist()); off.append(len(idx))
            a += 1; b += 1
        elif n0[a] < n1[b]:
            a += 1
        else:
            b += 1
    assert len(q_idx) > 300
    m = orbx.ORBmatcher(0.7, True, max_queries=4096, max_train=4096, max_pairs=1 << 20)
    bi, bd, sd = m.best2(d0[q_idx], d1, np.array(off, np.int32), np.array(idx, np.int32))
    obi, obd, osd = O.best2(d0[q_idx], d1, np.array(off, np.int32), np.array(idx, np.int32))
    assert np.array_equal(bi, obi) and np.array_equal(bd, obd) and np.array_equal(sd, osd)
    assert ((bd <= 50) & (bd < 0.7 * sd)).sum() > 50


@pytest.mark.gpu
@pytest.mark.parametrize("nnratio,check_ori,with_valid", [(0.7, True, False), (0.9, False, True), (0.6, True, True)])
def test_search_by_bow_equals_oracle(orbx, synth, tmp_path, nnratio, check_ori, with_valid):
    """orbm_search_by_bow == ORBmatcher::SearchByBoW restated (order-dependent skip of already matched frame features,
    TH_LOW, ratio test, rotation histogram): same match table and the same return value."""
    path = str(tmp_path / "voc.txt")
    make_vocabulary(path, 10, 3, seed=9)
    v = orbx.ORBVocabulary(path)
    f0, f1 = synth.frame_pair(3, 640, 480)
    ex = orbx.ORBextractor(1000, max_width=640, max_height=480)
    k0, d0 = ex(f0); k1, d1 = ex(f1)
    _, fv0 = v.transform(d0, 2)
    _, fv1 = v.transform(d1, 2)
    valid = None
    if with_valid:
        valid = (np.random.default_rng(1).random(len(d0)) < 0.7).astype(np.uint8)   # some key-frame features without a MapPoint
    m = orbx.ORBmatcher(nnratio, check_ori, max_queries=4096, max_train=4096, max_pairs=1 << 20)
    mf, nm = m.SearchByBoW(k0, d0, fv0, k1, d1, fv1, valid)
    omf, onm = O.search_by_bow(d0, k0["angle"], fv0, d1, k1["angle"], fv1, nnratio, check_ori, valid)
    assert nm == onm and np.array_equal(mf, omf)
    assert nm > 40 and nm == int((mf >= 0).sum())
    if with_valid:
        assert valid[mf[mf >= 0]].all()
    # degenerate inputs
    e = (np.zeros(0, np.int32), np.zeros(1, np.int32), np.zeros(0, np.int32))
    mf2, nm2 = m.SearchByBoW(k0, d0, e, k1, d1, fv1)
    assert nm2 == 0 and (mf2 == -1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("levelsup", [0, 1, 3])
def test_search_by_bow_node_sizes(orbx, synth, tmp_path, levelsup):
    """Node granularity from single words (a handful of features per node) to the root (every feature in one node: 16
    chunks of 64 lanes in k_bow_select, and every key-frame feature competes for every frame feature)."""
    path = str(tmp_path / "voc.txt")
    make_vocabulary(path, 10, 3, seed=9)
    v = orbx.ORBVocabulary(path)
    f0, f1 = synth.frame_pair(3, 640, 480)
    ex = orbx.ORBextractor(1000, max_width=640, max_height=480)
    k0, d0 = ex(f0); k1, d1 = ex(f1)
    _, fv0 = v.transform(d0, levelsup)
    _, fv1 = v.transform(d1, levelsup)
    m = orbx.ORBmatcher(0.75, True, max_queries=4096, max_train=4096, max_pairs=1 << 21)
    mf, nm = m.SearchByBoW(k0, d0, fv0, k1, d1, fv1)
    omf, onm = O.search_by_bow(d0, k0["angle"], fv0, d1, k1["angle"], fv1, 0.75, True, None)
    assert nm == onm and np.array_equal(mf, omf)
    assert nm > 20


@pytest.mark.gpu
def test_search_by_bow_host_selection_fallback():
    """The general path of orbm_search_by_bow (all node-mate distances on the GPU, selection scan on the host) is what
    nodes with more than 4096 frame features take; ORBM_BOW_HOST_SELECT=1 forces it.  Same tests, in a child process
    because the switch is read once per process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ORBM_BOW_HOST_SELECT="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-k", "search_by_bow_equals_oracle or search_by_bow_node_sizes",
                        os.path.join(root, "tests", "test_vocabulary.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "6 passed" in r.stdout
