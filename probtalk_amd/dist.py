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
    """Inverse of shard_batch along dim 0 on every rank.  Ragged shards (n_total not a multiple of the world size) are
    allowed; without n_total the per-rank lengths are exchanged first, so ranks holding q and q+1 items agree on the
    sizes (deriving n_total from the local length would not)."""
    world = dist.get_world_size()
    if world == 1:
        return t
    if n_total is None:
        lens = torch.zeros(world, dtype=torch.int64, device=t.device if dist.get_backend() != "gloo" else "cpu")
        lens[dist.get_rank()] = t.shape[0]
        dist.all_reduce(lens)
        sizes = [int(v) for v in lens.tolist()]
    else:
        sizes = [hi - lo for lo, hi in (shard_range(n_total, r, world) for r in range(world))]
        if sizes[dist.get_rank()] != t.shape[0]:
            raise ValueError(f"gather_batch: rank {dist.get_rank()} holds {t.shape[0]} items, its share of {n_total} is "
                             f"{sizes[dist.get_rank()]}")
    mx = max(sizes)
    pad = t
    if t.shape[0] < mx:
        pad = torch.cat([t, t.new_zeros((mx - t.shape[0],) + tuple(t.shape[1:]))])
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous())
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)])


class OutputGatherer:
    """Gather of equal-sized per-rank outputs to rank `dst` on a side stream, so it overlaps the next step's compute.

    north_star asks for a gather, not an all-gather: only rank 0 needs the 4096 x 376 x 300 fp32 result (1.85 GB at
    configs[3]).  Each peer sends its 231 MB once over its own point-to-point xGMI link to rank 0 (7 links in parallel,
    ~1.5 ms at 153 GB/s per link); an all_gather would move 8x the bytes over the fabric and hold 2 x 1.85 GB of buffers
    on every rank.  Only `dst` allocates the double-buffered result."""

    def __init__(self, world, shape, device, dst=0):
        self.world, self.dst = world, dst
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.cuda = torch.device(device).type == "cuda"
        self.bufs = [torch.empty((world * shape[0],) + tuple(shape[1:]), device=device) for _ in range(2)] \
            if self.rank == dst else [None, None]
        self.i = 0
        self.stream = torch.cuda.Stream(device=device) if self.cuda else None

    def _gather(self, out, buf):
        host = self.cuda and dist.get_backend() == "gloo"       # rehearsal on one GPU: gloo gets host copies
        src = out.cpu() if host else out.contiguous()
        if self.rank == self.dst:
            parts = [torch.empty(src.shape, dtype=src.dtype) for _ in range(self.world)] if host else list(buf.chunk(self.world))
            dist.gather(src, parts, dst=self.dst)
            if host:
                buf.copy_(torch.cat(parts))
        else:
            dist.gather(src, None, dst=self.dst)

    def submit(self, out):
        """Start gathering `out`; returns the destination buffer on rank dst (valid after wait()), None elsewhere."""
        buf = self.bufs[self.i % 2]
        self.i += 1
        if not self.cuda:
            self._gather(out, buf)
            return buf
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            self._gather(out, buf)
        out.record_stream(self.stream)
        return buf

    def wait(self):
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.stream)

    def last(self):
        return self.bufs[(self.i - 1) % 2]


# tools/micro/dist_train_debug.py sets this to hand gloo the device tensors themselves (the configuration that hung)
GLOO_DEVICE_TENSORS = False


class GradientReducer:
    """Data-parallel training: averages the gradients of `params` over all ranks in flat buckets.

    The reference trains under nn.DataParallel (predictor.py:101-110 `to_parallel`): the batch is split over devices,
    BatchNorm uses per-replica batch statistics, the loss is the mean over the whole batch, so the gradient is the mean
    of the per-replica gradients and device 0's BatchNorm running statistics are the ones kept.  One process per GPU
    reproduces exactly that with one all-reduce per bucket (RCCL ring over xGMI: per-link bound, so a few large
    messages — 32 MB buckets by default, ~6 for the predictor's 46 M parameters — rather than one per parameter),
    launched asynchronously DURING backward (a bucket goes out as soon as its last gradient has been accumulated; buckets
    are filled last-parameter-first, the order backward produces gradients in), plus a broadcast of rank 0's BatchNorm
    buffers.
    """

    def __init__(self, params, buffers=(), bucket_bytes=32 << 20, overlap=True):
        self.params = [p for p in params if p.requires_grad]
        self.buffers = [b for b in buffers if torch.is_floating_point(b)]
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel() * 4
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self._flat = {}
        # overlap with backward: a bucket is all-reduced as soon as autograd has accumulated the gradient of its last
        # parameter (post-accumulate-grad hooks), while the earlier layers' gradients are still being computed;
        # reduce() launches whatever is left (buckets holding a parameter without gradient) and waits for all of them
        self._bucket_of = {id(p): i for i, bk in enumerate(self.buckets) for p in bk}
        self._ready = [0] * len(self.buckets)
        # gradients a bucket waits for before it leaves: all of its parameters at first; after a step, the ones that
        # received a gradient in that step (parameters the graph never reaches would otherwise hold every bucket back
        # until reduce(); the graph is the same on every rank and every step)
        self._expect = [len(bk) for bk in self.buckets]
        self._pending = {}
        self.launched_in_backward = 0
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params] if overlap else []

    def _flat_for(self, i, n, like):
        f = self._flat.get(i)
        if f is None or f.numel() != n or f.device != like.device:
            f = self._flat[i] = torch.empty(n, dtype=torch.float32, device=like.device)
        return f

    def _launch(self, i):
        live = [p for p in self.buckets[i] if p.grad is not None]
        if not live:
            return
        flat = self._flat_for(i, sum(p.numel() for p in live), live[0].grad)
        torch.cat([p.grad.reshape(-1) for p in live], out=flat)
        # gloo (CPU rehearsals, or several ranks on one GPU in tests) gets a host copy.  Handed the device buffers, the
        # predictor's six buckets (7.1-9.9 M floats each; one launched from the autograd thread, five from reduce(), same
        # order on both ranks) are all in flight as async all_reduce and the first work.wait() never returns on either
        # rank (profiles/r2/gloo_device_tensor_hang.txt; three 64 K-float all_reduce launched the same way complete,
        # profiles/r2/gloo_cuda_probe_small.txt).  RCCL reduces the device buffer in place; a stuck collective there ends
        # the job through init_process_group's timeout (bench.py)
        msg = flat.cpu() if flat.is_cuda and dist.get_backend() == "gloo" and not GLOO_DEVICE_TENSORS else flat
        self._pending[i] = (dist.all_reduce(msg, op=dist.ReduceOp.SUM, async_op=True), flat, live, msg)

    def _on_grad(self, p):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        i = self._bucket_of[id(p)]
        self._ready[i] += 1
        if self._ready[i] == self._expect[i] and i not in self._pending:
            self._launch(i)
            self.launched_in_backward += 1

    def reduce(self):
        """Call after backward, before clipping / the optimizer step.  Parameters whose gradient is None on this rank
        (unused on every rank: the graph is the same everywhere) are skipped.  Buckets completed during backward are
        already in flight; the rest start here."""
        world = dist.get_world_size()
        if world == 1:
            return
        for i in range(len(self.buckets)):
            if i not in self._pending:
                self._launch(i)
        for i in sorted(self._pending):
            work, flat, live, msg = self._pending[i]
            self._expect[i] = len(live)
            work.wait()
            if msg is not flat:
                flat.copy_(msg)
            flat.mul_(1.0 / world)
            off = 0
            for p in live:
                p.grad.copy_(flat[off:off + p.numel()].view_as(p.grad))
                off += p.numel()
        self._pending = {}
        self._ready = [0] * len(self.buckets)

    def agree(self, *scalars):
        """Rank 0's values of small device tensors (gradient norms, clip coefficients) replace everyone's, in place.
        The norm kernels sum with floating-point atomics, so two ranks holding the same averaged gradient can differ in
        the last bit of the norm; a clip coefficient that differs between ranks would let the replicas drift apart."""
        if dist.get_world_size() == 1 or not scalars:
            return
        flat = torch.cat([t.detach().reshape(-1).float() for t in scalars])
        msg = flat.cpu() if flat.is_cuda and dist.get_backend() == "gloo" else flat
        dist.broadcast(msg, src=0)
        if msg is not flat:
            flat.copy_(msg)
        off = 0
        for t in scalars:
            t.detach().copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()

    def sync_buffers(self):
        """Rank 0's BatchNorm running statistics win, as device 0's do under nn.DataParallel."""
        if dist.get_world_size() == 1:
            return
        if not self.buffers:
            return
        # one flat message per dtype instead of one per buffer (a BatchNorm layer has two 256-float buffers)
        for dt in {b.dtype for b in self.buffers}:
            bs = [b for b in self.buffers if b.dtype == dt]
            flat = torch.cat([b.reshape(-1) for b in bs])
            if flat.is_cuda and dist.get_backend() == "gloo":
                host = flat.cpu()
                dist.broadcast(host, src=0)
                flat.copy_(host)
            else:
                dist.broadcast(flat, src=0)
            off = 0
            for b in bs:
                b.copy_(flat[off:off + b.numel()].view_as(b))
                off += b.numel()
