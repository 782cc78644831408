"""Frame-level data parallelism: frames are independent units, so a stream is cut into
contiguous per-rank blocks and every rank extracts its block with no data-path collective.  Two things cross
GPUs, both tiny and both here:

  * the boundary exchange: frame k is matched against frame k-1, so the first frame of a rank's block needs the
    keypoints/descriptors of the previous rank's LAST frame (<= 62 KB, one send/recv per rank and step);
  * one gather per step of the flat result block to rank 0 (RCCL over xGMI when the backend is "nccl").

Everything a rank produces in a step lives in ONE flat uint8 block (FlatLayout) so that one gather moves it.
Slot 0 of the block is the previous rank's last frame (the train side of the first match pair), slots 1..b are the
rank's own frames: the matcher then sees b uniform (query = slot i, train = slot i-1) pairs.

Backend-agnostic: tensors may live on the GPU (nccl) or the CPU (gloo: the CPU tests and rehearsals).
bench.py and tests/test_dist_gloo.py run exactly these functions."""
import torch
import torch.distributed as dist


def shard_range(nframes, world, rank):
    """Contiguous block [lo, hi) of `nframes` owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(nframes, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_shard(nframes, world):
    return (nframes + world - 1) // world


class FlatLayout:
    """[b+1][cap] 28-B keypoints | [b+1][cap][32] descriptors | [b+1][cap] int32 match (index into the previous
    frame, -1 = none) | [b+1] int32 keypoint counts | [b+1] int32 match counts.  Slot 0 = the previous rank's last
    frame (keypoints, descriptors and count only)."""

    def __init__(self, b, cap):
        self.b, self.cap = int(b), int(cap)
        s = self.b + 1
        self.nb_k, self.nb_d, self.nb_m = s * cap * 28, s * cap * 32, s * cap * 4
        self.off_k = 0
        self.off_d = self.off_k + self.nb_k
        self.off_m = self.off_d + self.nb_d
        self.off_c = self.off_m + self.nb_m
        self.off_n = self.off_c + s * 4
        self.nbytes = self.off_n + s * 4

    def alloc(self, device="cpu"):
        buf = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        self.views(buf)[2].fill_(-1)
        return buf

    def views(self, buf):
        """(kps [b+1,cap,7] f32, desc [b+1,cap,32] u8, match12 [b+1,cap] i32, counts [b+1] i32, nmatch [b+1] i32)"""
        s, cap = self.b + 1, self.cap
        return (buf[self.off_k:self.off_d].view(torch.float32).view(s, cap, 7),
                buf[self.off_d:self.off_m].view(s, cap, 32),
                buf[self.off_m:self.off_c].view(torch.int32).view(s, cap),
                buf[self.off_c:self.off_n].view(torch.int32),
                buf[self.off_n:self.nbytes].view(torch.int32))


def boundary_exchange(layout, buf, rank, world, nown, group=None):
    """Rank r sends its last own frame (slot `nown`) to rank r+1, which receives it into slot 0.  Returns the list
    of work handles (wait on them before the match of slot 1 is enqueued / before `buf` is reused).  Rank 0's slot 0
    stays empty (count 0): the stream's first frame has no predecessor."""
    if world == 1:
        return []
    kps, desc, _, counts, _ = layout.views(buf)
    ops = []
    if rank + 1 < world:
        ops += [dist.P2POp(dist.isend, kps[nown], rank + 1, group), dist.P2POp(dist.isend, desc[nown], rank + 1, group),
                dist.P2POp(dist.isend, counts[nown:nown + 1], rank + 1, group)]
    if rank > 0:
        ops += [dist.P2POp(dist.irecv, kps[0], rank - 1, group), dist.P2POp(dist.irecv, desc[0], rank - 1, group),
                dist.P2POp(dist.irecv, counts[0:1], rank - 1, group)]
    return dist.batch_isend_irecv(ops) if ops else []


def gather_flat(buf, gather_bufs, dst=0, group=None, async_op=False):
    """One gather of the whole flat block to `dst` (gather_bufs: list of `world` tensors like buf on dst, else None)."""
    return dist.gather(buf, gather_bufs if dist.get_rank(group) == dst else None, dst=dst, group=group, async_op=async_op)


def assemble_flat(layout, gathered, ranges):
    """Undo the sharding on rank 0: per frame, in stream order, (keypoints [n,7] f32, descriptors [n,32] u8,
    match12 [n] i32).  `ranges[r]` = (lo, hi) of rank r; `gathered[r]` = rank r's flat block."""
    out = []
    for r, (lo, hi) in enumerate(ranges):
        kps, desc, m12, counts, _ = layout.views(gathered[r])
        for j in range(1, hi - lo + 1):
            n = int(counts[j])
            out.append((kps[j, :n].cpu(), desc[j, :n].cpu(), m12[j, :n].cpu()))
    return out


def run_step(layout, buf, rank, world, nown, extract_fn, match_fn, group=None, comm_buf=None, stage_out=None, stage_in=None):
    """One step of the sharded hot path on this rank: extract the rank's `nown` frames into slots 1..nown, exchange the
    boundary frame, match every own frame against its predecessor.  extract_fn(kps, desc, counts) fills the views of
    slots 1..nown; match_fn(first_slot, npairs) matches slots first_slot..first_slot+npairs-1 against the slot before
    each.  The boundary receive is waited for only before the one pair that needs it, after the block's own pairs
    have been enqueued.
    comm_buf/stage_out/stage_in: when the communication backend cannot reach `buf` (gloo with `buf` in HBM), the exchange
    runs on the mirror `comm_buf`: stage_out() copies buf -> comm_buf after the extraction, stage_in() copies slot 0 of
    comm_buf back into buf after the receive."""
    kps, desc, m12, counts, nmatch = layout.views(buf)
    extract_fn(kps[1:nown + 1], desc[1:nown + 1], counts[1:nown + 1])
    if world > 1 and stage_out is not None:
        stage_out()
    works = boundary_exchange(layout, buf if comm_buf is None else comm_buf, rank, world, nown, group)
    if nown > 1:
        match_fn(2, nown - 1)              # pairs inside the block: no other rank involved
    if rank > 0:                           # rank 0 holds the stream's first frame, which has no predecessor
        for w in works:
            w.wait()
        works = []
        if stage_in is not None:
            stage_in()
        match_fn(1, 1)                     # the block's first frame against the previous rank's last
    return works                           # pending sends (ranks 0 .. world-2): wait before the slot is rewritten
