"""world_size-2 gloo tests of the batch sharding / output gather plumbing (no GPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from probtalk_amd.dist import OutputGatherer, gather_batch, shard_batch, shard_range


def test_shard_range_partitions():
    for n in (0, 1, 7, 64, 4096):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_clips):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(n_clips * 6, dtype=torch.float32).reshape(n_clips, 2, 3)
        mine = shard_batch(full)
        lo, hi = shard_range(n_clips, rank, world)
        assert torch.equal(mine, full[lo:hi])
        out = mine * 2 + 1                      # stand-in for the per-rank pipeline
        back = gather_batch(out, n_total=n_clips)
        assert torch.equal(back, full * 2 + 1)
        back = gather_batch(out)                # ragged shards without n_total: lengths are exchanged first
        assert torch.equal(back, full * 2 + 1)
        if n_clips % world == 0:
            g = OutputGatherer(world, tuple(out.shape), "cpu")      # gather to rank 0 only
            for rep in range(3):                                    # double-buffered: three steps reuse buffer 0
                got = g.submit(out + rep)
                g.wait()
                if rank == 0:
                    assert torch.equal(got, full * 2 + 1 + rep) and got is g.last()
                else:
                    assert got is None
    finally:
        dist.destroy_process_group()


def test_shard_gather_world2_even_and_ragged():
    for n in (8, 7):
        mp.spawn(_worker, args=(2, _free_port(), n), nprocs=2, join=True)


def _grad_worker(rank, world, port):
    from probtalk_amd.dist import GradientReducer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        shapes = [(300, 7), (5,), (1000,), (64, 64), (3,), (17, 2)]
        params = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
        frozen = torch.nn.Parameter(torch.randn(4), requires_grad=False)
        grads = [[torch.randn(s, generator=g) for s in shapes] for _ in range(world)]      # same on every rank
        for p, gr in zip(params, grads[rank]):
            p.grad = gr.clone()
        params[4].grad = None                                   # unused parameter: no gradient on any rank
        bn = torch.full((8,), float(rank + 1))
        red = GradientReducer(params + [frozen], [bn, torch.tensor(3)], bucket_bytes=4096)
        assert len(red.buckets) >= 2 and sum(len(b) for b in red.buckets) == len(params)
        for _ in range(2):                                      # twice: flat buffers are reused
            for p, gr in zip(params, grads[rank]):
                if p.grad is not None:
                    p.grad.copy_(gr)
            red.reduce()
            for i, p in enumerate(params):
                if i == 4:
                    assert p.grad is None
                else:
                    want = sum(grads[r][i] for r in range(world)) / world
                    assert torch.allclose(p.grad, want, atol=1e-6)
        red.sync_buffers()
        assert torch.equal(bn, torch.full((8,), 1.0))
    finally:
        dist.destroy_process_group()


def test_gradient_reducer_world2():
    mp.spawn(_grad_worker, args=(2, _free_port()), nprocs=2, join=True)


def _overlap_worker(rank, world, port):
    """Gradients produced by a real backward: buckets whose parameters are all used go out from the autograd hooks
    (before reduce() is called), the bucket holding an unused parameter starts inside reduce(); result = mean over ranks."""
    from probtalk_amd.dist import GradientReducer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        w = [torch.nn.Parameter(torch.randn(64, 64, generator=g)) for _ in range(4)]
        unused = torch.nn.Parameter(torch.randn(8, generator=g))
        red = GradientReducer(w[:2] + [unused] + w[2:], bucket_bytes=64 * 64 * 4)
        assert [len(b) for b in red.buckets] == [1, 1, 2, 1]           # [w3] [w2] [unused, w1] [w0]
        xs = [torch.randn(16, 64, generator=g) for _ in range(world)]                       # same on every rank
        for step in range(2):
            for p in w:
                p.grad = None
            h = xs[rank]
            for p in w:
                h = torch.tanh(h @ p)
            h.pow(2).sum().backward()
            # step 0: the three complete buckets go out during backward; from then on the reducer knows that `unused`
            # never receives a gradient and its bucket leaves during backward as well
            assert red.launched_in_backward == 3 + 4 * step
            red.reduce()
            want = []
            for r in range(world):
                ps = [p.detach().clone().requires_grad_(True) for p in w]
                h = xs[r]
                for p in ps:
                    h = torch.tanh(h @ p)
                h.pow(2).sum().backward()
                want.append([p.grad for p in ps])
            for i, p in enumerate(w):
                assert torch.allclose(p.grad, sum(wr[i] for wr in want) / world, atol=1e-5)
            assert unused.grad is None
        # rank 0's small tensors win (clip coefficients must be identical on every rank)
        a, b = torch.tensor([1.0 + rank]), torch.tensor([[10.0 * (rank + 1), 3.0]])
        red.agree(a, b)
        assert a.tolist() == [1.0] and b.tolist() == [[10.0, 3.0]]
    finally:
        dist.destroy_process_group()


def test_gradient_reducer_overlaps_backward_world2():
    mp.spawn(_overlap_worker, args=(2, _free_port()), nprocs=2, join=True)


# ---- bench.py's own launcher (probtalk_amd/launch.py): `bench.py --gpus N` without torchrun
def _child_script(tmp_path, body):
    p = tmp_path / "child.py"
    p.write_text("import os, sys, time\nrank = int(os.environ['RANK'])\n" + body)
    return [os.sys.executable, str(p)]


def test_launcher_all_ranks_ok(tmp_path):
    from probtalk_amd.launch import spawn_ranks
    cmd = _child_script(tmp_path, "assert os.environ['WORLD_SIZE'] == '3' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                                  "open(os.path.join(os.path.dirname(__file__), f'ok{rank}'), 'w').write(os.environ['MASTER_PORT'])\n")
    assert spawn_ranks(cmd, 3, 30) == 0
    ports = {(tmp_path / f"ok{r}").read_text() for r in range(3)}
    assert len(ports) == 1


def test_launcher_failing_rank_stops_the_job(tmp_path):
    """A rank that exits 1 while the others sit in a (here: simulated) collective: the parent kills them and returns
    non-zero well within the limit."""
    import time
    from probtalk_amd.launch import spawn_ranks
    cmd = _child_script(tmp_path, "open(os.path.join(os.path.dirname(__file__), f'pid{rank}'), 'w').write(str(os.getpid()))\n"
                                  "if rank == 1:\n    time.sleep(0.5); sys.exit(1)\ntime.sleep(120)\n")
    t0 = time.monotonic()
    rc = spawn_ranks(cmd, 2, 60)
    assert rc == 1 and time.monotonic() - t0 < 20
    pid0 = int((tmp_path / "pid0").read_text())
    try:
        os.kill(pid0, 0)
        alive = True
    except OSError:
        alive = False
    assert not alive, "rank 0 was left running"


def test_launcher_wall_clock_limit(tmp_path):
    import time
    from probtalk_amd.launch import spawn_ranks
    cmd = _child_script(tmp_path, "time.sleep(120)\n")
    t0 = time.monotonic()
    assert spawn_ranks(cmd, 2, 1.5) == 124
    assert time.monotonic() - t0 < 15


def test_bench_parent_exits_nonzero_when_a_rank_fails():
    """`bench.py --gpus 2` as the driver runs it (no torchrun, no WORLD_SIZE): the parent spawns two ranks before touching
    the GPU; PT_BENCH_FAIL_RANK makes one of them exit 1 at start-up, the parent must follow with a non-zero code."""
    import subprocess
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(PT_BENCH_FAIL_RANK="1", PT_BENCH_BACKEND="gloo", PT_BENCH_LAUNCH_LIMIT="60")
    t0 = time.monotonic()
    r = subprocess.run([os.sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and time.monotonic() - t0 < 90
    assert "[launch] rank" in r.stderr
