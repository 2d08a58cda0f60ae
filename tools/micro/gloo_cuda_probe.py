"""Probe: does gloo move CUDA tensors on this build (2 ranks on one GPU)?  sync all_reduce, async all_reduce, async from an
autograd hook."""
import os, sys, faulthandler
import torch, torch.distributed as dist, torch.multiprocessing as mp


def w(rank, port):
    faulthandler.dump_traceback_later(40, exit=True)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    t = torch.full((1 << 20,), float(rank + 1), device="cuda")
    dist.all_reduce(t)
    print(rank, "sync ok", float(t[0]), flush=True)
    t = torch.full((1 << 20,), float(rank + 1), device="cuda")
    wk = dist.all_reduce(t, async_op=True)
    wk.wait()
    print(rank, "async ok", float(t[0]), flush=True)
    ws = [torch.nn.Parameter(torch.randn(256, 256, device="cuda")) for _ in range(3)]
    pend = []
    for p in ws:
        p.register_post_accumulate_grad_hook(lambda q: pend.append(dist.all_reduce(q.grad, async_op=True)))
    x = torch.randn(8, 256, device="cuda")
    for p in ws:
        x = x @ p
    x.sum().backward()
    for wk in pend:
        wk.wait()
    print(rank, "hook async ok", float(ws[0].grad.sum()), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(w, args=(int(sys.argv[1]),), nprocs=2, join=True)
