"""Multi-GPU plumbing: one process per GPU, clips sharded over ranks, outputs gathered over RCCL.

The path has no exchange step inside the models: clips are independent, weights are
replicated, so the only collective is the gather of generated poses.  `torch.distributed` backend
"nccl" is RCCL over xGMI on ROCm; the CPU tests run the same code on "gloo".
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) share of n_items for `rank` (first n_items % world ranks get one more)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_batch(t, rank=None, world=None, dim=0):
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_range(t.shape[dim], rank, world)
    return t.narrow(dim, lo, hi - lo)


def gather_batch(t, n_total=None):
    """Inverse of shard_batch along dim 0 on every rank (ragged last shards allowed)."""
    world = dist.get_world_size()
    if world == 1:
        return t
    n_total = n_total if n_total is not None else t.shape[0] * world
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = t
    if t.shape[0] < mx:
        pad = torch.cat([t, t.new_zeros((mx - t.shape[0],) + tuple(t.shape[1:]))])
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous())
    return torch.cat([b[:hi - lo] for b, (lo, hi) in zip(bufs, sizes)])


class OutputGatherer:
    """all_gather of equal-sized per-rank outputs on a side stream so it overlaps the next step's compute
    (bucket = one step's output, 28.9 MB per rank at 64 x 376 x 300 fp32: large enough to run xGMI
    point-to-point links at rate, small enough to hide behind a ~60 ms step)."""

    def __init__(self, world, shape, device):
        self.world = world
        self.cuda = torch.device(device).type == "cuda"
        self.bufs = [torch.empty((world * shape[0],) + tuple(shape[1:]), device=device) for _ in range(2)]
        self.i = 0
        self.stream = torch.cuda.Stream(device=device) if self.cuda else None

    def submit(self, out):
        buf = self.bufs[self.i % 2]
        self.i += 1
        if not self.cuda:
            dist.all_gather_into_tensor(buf, out.contiguous()) if dist.get_backend() != "gloo" else \
                dist.all_gather(list(buf.chunk(self.world)), out.contiguous())
            return buf
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            dist.all_gather_into_tensor(buf, out)
        out.record_stream(self.stream)
        return buf

    def wait(self):
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.stream)

    def last(self):
        return self.bufs[(self.i - 1) % 2]
