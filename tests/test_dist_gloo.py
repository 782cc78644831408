"""Multi-rank path on CPU: world_size-2 (and 3) gloo.  The ranks run my_slam_amd.shard.run_step -- the function
bench.py runs on the GPUs -- over their blocks of ONE stream, with the oracle standing in for the HIP extractor and
matcher (this test is about the sharding, the boundary exchange and the flat-block gather, not about the kernels), and
rank 0 must end up with exactly what one process gets for the whole stream: keypoints, descriptors and
the match of every frame against its predecessor, including the pairs that straddle two ranks."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NFRAMES, W, H, NF = 7, 200, 160, 200
CAP = NF + 64


def _fill(ex, O, frames, kps, desc, counts):
    for j in range(len(frames)):
        k, d, _ = ex.extract(frames[j])
        n = len(k)
        kps[j, :n] = torch.from_numpy(k.view(np.float32).reshape(n, 7).copy())
        desc[j, :n] = torch.from_numpy(d)
        counts[j] = n


def _match(O, layout, buf, first, npairs):
    kps, desc, m12, counts, nmatch = layout.views(buf)
    for s in range(first, first + npairs):
        nq, nt = int(counts[s]), int(counts[s - 1])
        n, m = O.match_dense(desc[s, :nq].numpy(), kps[s, :nq, 3].numpy(), desc[s - 1, :nt].numpy(), kps[s - 1, :nt, 3].numpy(), 50, 0.9, True)
        m12[s, :nq] = torch.from_numpy(m)
        nmatch[s] = n


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import conftest  # noqa: F401
    import oracle_lib as O
    import my_slam_amd.shard as shard
    import my_slam_amd.synth as synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(NFRAMES, world, rank)
    frames = synth.stream(4, W, H, NFRAMES, first=lo, count=hi - lo)      # this rank's block of the ONE stream
    layout = shard.FlatLayout(shard.max_shard(NFRAMES, world), CAP)
    buf = layout.alloc()
    ex = O.Extractor(NF)
    works = shard.run_step(layout, buf, rank, world, hi - lo,
                           lambda k, d, c: _fill(ex, O, frames, k, d, c),
                           lambda first, npairs: _match(O, layout, buf, first, npairs))
    for w in works:
        w.wait()
    gb = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    shard.gather_flat(buf, gb)
    if rank == 0:
        res = shard.assemble_flat(layout, gb, [shard.shard_range(NFRAMES, world, r) for r in range(world)])
        q.put([(a.numpy().tobytes(), b_.numpy().tobytes(), c.numpy().tobytes()) for a, b_, c in res])
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_cover_everything():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import conftest  # noqa: F401
    import my_slam_amd.shard as shard
    for n in (1, 5, 64, 67):
        for w in (1, 2, 4, 8):
            r = [shard.shard_range(n, w, k) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in r) == shard.max_shard(n, w)


def test_stream_shards_are_the_stream(synth):
    whole = synth.stream(4, 96, 64, 9)
    for lo, hi in ((0, 4), (4, 9), (8, 9)):
        assert np.array_equal(synth.stream(4, 96, 64, 9, first=lo, count=hi - lo), whole[lo:hi])


@pytest.mark.parametrize("world", [2, 3])
def test_n_rank_step_equals_single_process(synth, world):
    import oracle_lib as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    frames = synth.stream(4, W, H, NFRAMES)
    ex = O.Extractor(NF)
    assert len(got) == NFRAMES
    prev = None
    for f in range(NFRAMES):
        k, d, _ = ex.extract(frames[f])
        assert got[f][0] == k.tobytes() and got[f][1] == d.tobytes()
        if prev is None:
            m = np.full(len(k), -1, np.int32)          # the stream's first frame has no predecessor
        else:
            _, m = O.match_dense(d, k["angle"], prev[1], prev[0]["angle"], 50, 0.9, True)
        assert got[f][2] == m.astype(np.int32).tobytes(), "frame %d: match table differs" % f
        prev = (k, d)
