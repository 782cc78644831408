"""Multi-rank path on CPU: world_size-2 gloo.  Each rank extracts its block of frames (with the
oracle standing in for the GPU extractor -- this test is about the sharding + gather plumbing that
bench.py and a batched caller use) and rank 0 must end up with exactly the single-process result."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NFRAMES, W, H, NF = 5, 200, 160, 200


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import conftest  # noqa: F401
    import oracle_lib as O
    import my_slam_amd.shard as shard
    import my_slam_amd.synth as synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = synth.stream(4, W, H, NFRAMES)
    lo, hi = shard.shard_range(NFRAMES, world, rank)
    b = shard.max_shard(NFRAMES, world)
    ex = O.Extractor(NF)
    cap = NF + 64
    kps = torch.zeros((b, cap, 7), dtype=torch.float32)
    desc = torch.zeros((b, cap, 32), dtype=torch.uint8)
    counts = torch.zeros(b, dtype=torch.int32)
    for j, f in enumerate(range(lo, hi)):
        k, d, _ = ex.extract(frames[f])
        n = len(k)
        kps[j, :n] = torch.from_numpy(k.view(np.float32).reshape(n, 7).copy())
        desc[j, :n] = torch.from_numpy(d)
        counts[j] = n
    g = shard.gather_results(kps, desc, counts, dst=0)
    if rank == 0:
        res = shard.assemble(NFRAMES, world, g)
        q.put([(a.numpy().tobytes(), b_.numpy().tobytes()) for a, b_ in res])
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


def test_two_rank_gather_equals_single_process(synth):
    import oracle_lib as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    frames = synth.stream(4, W, H, NFRAMES)
    ex = O.Extractor(NF)
    assert len(got) == NFRAMES
    for f in range(NFRAMES):
        k, d, _ = ex.extract(frames[f])
        assert got[f][0] == k.tobytes() and got[f][1] == d.tobytes()
