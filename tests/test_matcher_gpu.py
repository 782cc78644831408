"""GPU parity for the Hamming primitives: HIP (through the C ABI) == oracle, exactly."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _rand_desc(rng, n, clusters=None):
    d = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    if clusters is not None:   # near-duplicates so that ties and small distances occur
        base = rng.integers(0, 256, size=(clusters, 32), dtype=np.uint8)
        d = base[rng.integers(0, clusters, n)].copy()
        flips = rng.integers(0, 256, size=(n, 3))
        for i in range(n):
            for b in flips[i]:
                d[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return d


@pytest.mark.parametrize("nq,nt", [(1, 1), (7, 300), (1000, 1003), (4000, 4001)])
def test_best2_dense(orbx, nq, nt):
    rng = np.random.default_rng(nq * 7 + nt)
    q, t = _rand_desc(rng, nq, 40), _rand_desc(rng, nt, 40)
    m = orbx.ORBmatcher()
    bi, bd, sd = m.best2(q, t)
    obi, obd, osd = O.best2(q, t)
    assert np.array_equal(bd, obd) and np.array_equal(sd, osd) and np.array_equal(bi, obi)


def test_best2_csr_and_distances(orbx):
    rng = np.random.default_rng(5)
    nq, nt = 777, 900
    q, t = _rand_desc(rng, nq, 30), _rand_desc(rng, nt, 30)
    lens = rng.integers(0, 150, nq)
    lens[::50] = 0                       # empty candidate lists
    off = np.zeros(nq + 1, np.int32)
    off[1:] = np.cumsum(lens)
    idx = rng.integers(0, nt, int(off[-1])).astype(np.int32)
    m = orbx.ORBmatcher()
    bi, bd, sd = m.best2(q, t, off, idx)
    obi, obd, osd = O.best2(q, t, off, idx)
    assert np.array_equal(bd, obd) and np.array_equal(sd, osd) and np.array_equal(bi, obi)
    d = m.distances(q, t, off, idx)
    L = O.lib()
    ref = np.array([L.oro_descriptor_distance(q[i].ctypes.data, t[idx[c]].ctypes.data)
                    for i in range(nq) for c in range(off[i], off[i + 1])], np.int32)
    assert np.array_equal(d, ref)
    dd = m.distances(q[:50], t[:60])
    ref = np.array([[L.oro_descriptor_distance(q[i].ctypes.data, t[j].ctypes.data) for j in range(60)] for i in range(50)], np.int32)
    assert np.array_equal(dd.reshape(50, 60), ref)


def test_descriptor_distance_host(orbx):
    rng = np.random.default_rng(1)
    a, b = _rand_desc(rng, 64), _rand_desc(rng, 64)
    L = O.lib()
    for i in range(64):
        assert orbx.ORBmatcher.DescriptorDistance(a[i], b[i]) == L.oro_descriptor_distance(a[i].ctypes.data, b[i].ctypes.data)
    assert orbx.ORBmatcher.DescriptorDistance(a[0], a[0]) == 0
    assert orbx.ORBmatcher.DescriptorDistance(np.zeros(32, np.uint8), np.full(32, 255, np.uint8)) == 256


def test_extract_then_match_frame_pair(orbx, synth):
    """BASELINE config 3 shape (reduced): extract two shifted frames, dense match with rotation filter."""
    W, H, n = 960, 540, 2000
    f0, f1 = synth.frame_pair(2, W, H)
    ex = orbx.ORBextractor(n, max_width=W, max_height=H)
    k0, d0 = ex(f0)
    k1, d1 = ex(f1)
    m = orbx.ORBmatcher(0.9, True)
    nm, m12 = m.match_dense(d1, k1, d0, k0)
    onm, om12 = O.match_dense(d1, k1["angle"], d0, k0["angle"], 50, 0.9, True)
    assert nm == onm and np.array_equal(m12, om12)
    assert nm > 300     # the shifted frame really matches


@pytest.mark.parametrize("cap,nqs,nts", [(1100, [1000, 0, 1037], [990, 1100, 1]), (5000, [4500, 4999], [4800, 4097]),
                                         # 40 pairs: <= 16 train splits, k_accept_rot merges the partials itself; the two
                                         # cases above have 64 splits and go through the grid-wide k_merge_keys first
                                         (1100, [1000] * 38 + [0, 1037], [990] * 38 + [1100, 1])])
def test_match_batch_device(orbx, cap, nqs, nts):
    """Device-resident batched dense match (bench path): acceptance + rotation filter == oracle,
    including the > 4096-query sweep of k_accept_rot and empty / single-descriptor frames."""
    import torch
    rng = np.random.default_rng(cap)
    nb = len(nqs)
    q = np.zeros((nb, cap, 32), np.uint8); t = np.zeros((nb, cap, 32), np.uint8)
    kq = np.zeros((nb, cap), orbx.KP_DTYPE); kt = np.zeros((nb, cap), orbx.KP_DTYPE)
    for b in range(nb):
        base = _rand_desc(rng, 400, 60)
        t[b, :nts[b]] = base[rng.integers(0, 400, nts[b])]
        q[b, :nqs[b]] = base[rng.integers(0, 400, nqs[b])]
        flips = rng.integers(0, 256, (nqs[b], 2))
        for i in range(nqs[b]):
            for f in flips[i]:
                q[b, i, f >> 3] ^= np.uint8(1 << (f & 7))
        kt[b]["angle"] = rng.uniform(0, 360, cap).astype(np.float32)
        kq[b]["angle"] = (kt[b]["angle"][rng.integers(0, max(nts[b], 1), cap)] + np.float32(20.0)) % np.float32(360.0)
    dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
    dkq = torch.from_numpy(kq.view(np.float32).reshape(nb, cap, 7)).cuda()
    dkt = torch.from_numpy(kt.view(np.float32).reshape(nb, cap, 7)).cuda()
    dnq = torch.tensor(nqs, dtype=torch.int32).cuda(); dnt = torch.tensor(nts, dtype=torch.int32).cuda()
    m12 = torch.zeros((nb, cap), dtype=torch.int32).cuda(); nm = torch.zeros(nb, dtype=torch.int32).cuda()
    m = orbx.ORBmatcher(0.9, True, max_queries=cap, max_train=cap, max_pairs=1)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        m.match_batch_device(dq.data_ptr(), dkq.data_ptr(), dnq.data_ptr(), dt.data_ptr(), dkt.data_ptr(), dnt.data_ptr(),
                             cap, nb, m12.data_ptr(), nm.data_ptr(), stream=st.cuda_stream)
    st.synchronize()
    m12h, nmh = m12.cpu().numpy(), nm.cpu().numpy()
    for b in range(nb):
        on, om = O.match_dense(q[b, :nqs[b]], kq[b]["angle"][:nqs[b]], t[b, :nts[b]], kt[b]["angle"][:nts[b]], 50, 0.9, True) \
            if nqs[b] > 0 else (0, np.zeros(0, np.int32))
        assert nmh[b] == on
        assert np.array_equal(m12h[b, :nqs[b]], om)
        assert (m12h[b, nqs[b]:] == -1).all()


def test_mfma_operand_prefetch_variant_in_fresh_process():
    """ORBM_MFMA_SP=1 selects k_best2_mfma_sp (a whole tile's A operands prefetched, running best / second kept relative to the
    tile in hand; slower at the bench shape).  The switch is read once per process, so the dense and batched
    parity cases of this file run again in a process of their own with it set."""
    import os, subprocess, sys
    env = dict(os.environ, ORBM_MFMA_SP="1")
    here = os.path.abspath(__file__)
    p = subprocess.run([sys.executable, "-m", "pytest", here, "-q", "-x", "-m", "gpu", "-k", "test_best2_dense or test_match_batch_device or test_extract_then_match"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, cwd=os.path.dirname(os.path.dirname(here)))
    assert p.returncode == 0, p.stdout[-2000:]
    assert " passed" in p.stdout and "failed" not in p.stdout


@pytest.mark.parametrize("counts", [
    {0: 5, 3: 5, 7: 5, 9: 5},                 # four-way tie: the three lowest bins win (strict '>' in index order)
    {2: 100, 5: 9, 8: 9},                     # second < 10 % of the first: only the first bin survives
    {2: 100, 5: 10, 8: 9},                    # second exactly 10 %: kept; third below: dropped
    {11: 7, 1: 7, 6: 3, 4: 3, 12: 1},         # ties at two ranks
    {0: 1},                                   # one match only
    {4: 40, 5: 40, 6: 40, 7: 39, 8: 41},      # the maximum comes last in index order
])
def test_rotation_histogram_ties_and_ten_percent_rule(orbx, counts):
    """ComputeThreeMaxima (src/ORBmatcher.cc:1601-1642) inside k_accept_rot runs as three wave-wide maxima: rotation histograms with
    ties and with bins on both sides of the 10 % rule, built from perfect matches whose angle differences fall in chosen bins."""
    import torch
    rng = np.random.default_rng(sum(k * v for k, v in counts.items()) + 17)
    n = sum(counts.values())
    cap = max(n, 64)
    desc = np.zeros((cap, 32), np.uint8)
    desc[:n] = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    for i in range(n):                                                     # far apart: every query's best is its own copy, the second best is far
        desc[i, :4] = np.frombuffer(np.uint32(i * 2654435761 % (1 << 32)).tobytes(), np.uint8)
    bins = np.concatenate([np.full(v, k) for k, v in counts.items()]).astype(np.int32)
    rng.shuffle(bins)
    kq = np.zeros((1, cap), orbx.KP_DTYPE); kt = np.zeros((1, cap), orbx.KP_DTYPE)
    kt[0]["angle"][:n] = rng.uniform(0, 25, n).astype(np.float32)
    kq[0]["angle"][:n] = (kt[0]["angle"][:n] + np.float32(30.0) * bins.astype(np.float32)) % np.float32(360.0)   # rot = 30 * bin (bins 0..12)
    q = desc[None].copy(); t = desc[None].copy()
    dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
    dkq = torch.from_numpy(kq.view(np.float32).reshape(1, cap, 7)).cuda(); dkt = torch.from_numpy(kt.view(np.float32).reshape(1, cap, 7)).cuda()
    dn = torch.tensor([n], dtype=torch.int32).cuda()
    m12 = torch.zeros((1, cap), dtype=torch.int32).cuda(); nm = torch.zeros(1, dtype=torch.int32).cuda()
    m = orbx.ORBmatcher(0.9, True, max_queries=cap, max_train=cap, max_pairs=1)
    m.match_batch_device(dq.data_ptr(), dkq.data_ptr(), dn.data_ptr(), dt.data_ptr(), dkt.data_ptr(), dn.data_ptr(), cap, 1, m12.data_ptr(), nm.data_ptr())
    torch.cuda.synchronize()
    on, om = O.match_dense(q[0, :n], kq[0]["angle"][:n], t[0, :n], kt[0]["angle"][:n], 50, 0.9, True)
    assert on > 0 and int(nm.cpu()[0]) == on
    assert np.array_equal(m12.cpu().numpy()[0, :n], om)
