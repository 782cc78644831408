"""Frame-level data parallelism: frames are independent units, so a batch is cut
into contiguous per-rank blocks, every rank extracts its block with no data-path collective, and one
gather per buffer brings the fixed-size padded results (count, cap*28 B keypoints, cap*32 B
descriptors per frame) back to rank 0 -- RCCL over xGMI when the backend is "nccl", gloo on CPU.
Backend-agnostic: works on whatever device the tensors live on."""
import torch
import torch.distributed as dist


def shard_range(nframes, world, rank):
    """Contiguous block [lo, hi) of `nframes` owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(nframes, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_shard(nframes, world):
    return (nframes + world - 1) // world


def gather_results(kps, desc, counts, dst=0, group=None):
    """kps [b,cap,7] f32, desc [b,cap,32] u8, counts [b] i32 with identical b on every rank (pad the
    last shard).  Returns (kps, desc, counts) lists per rank on `dst`, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if rank == dst:
        bufs = [[torch.empty_like(t) for _ in range(world)] for t in (kps, desc, counts)]
    else:
        bufs = [None, None, None]
    dist.gather(kps, bufs[0], dst=dst, group=group)
    dist.gather(desc, bufs[1], dst=dst, group=group)
    dist.gather(counts, bufs[2], dst=dst, group=group)
    return tuple(bufs) if rank == dst else None


def assemble(nframes, world, gathered):
    """Undo the sharding on rank 0: per-frame (keypoints, descriptors) in original frame order."""
    gk, gd, gc = gathered
    out = []
    for r in range(world):
        lo, hi = shard_range(nframes, world, r)
        for j in range(hi - lo):
            n = int(gc[r][j])
            out.append((gk[r][j, :n], gd[r][j, :n]))
    return out
