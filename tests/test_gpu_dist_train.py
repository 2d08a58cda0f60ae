"""Two ranks on ONE GPU over gloo (gloo moves CUDA tensors through the host): the predictor TrainWrapper's data-parallel
step on the real HIP training kernels: bucketed all-reduce launched from backward hooks, averaged gradients, rank 0's
BatchNorm buffers.  (RCCL needs one GPU per rank; the collective calls are the same, only the backend differs.)"""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from probtalk_amd import synthetic
    from probtalk_amd.weights import make_input
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import faulthandler
    dbg = os.environ.get("PT_DIST_TEST_TRACE")
    if dbg:
        faulthandler.dump_traceback_later(60, file=open(f"{dbg}.{rank}", "w"), exit=True)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pred, _ = synthetic.build_pipeline(seed=0)
        pred.train_dropout = 0.0
        B, T_ = 2, 88
        bat = {"aud_feat": make_input(f"dt.aud{rank}", (B, 768, T_)), "poses": 0.3 * make_input(f"dt.poses{rank}", (B, 330, T_)),
               "expression": 0.3 * make_input(f"dt.expr{rank}", (B, 100, T_)), "speaker": torch.tensor([rank, 3 - rank]),
               "epoch": 10, "_mask": torch.zeros(B, 1, T_),
               "_keep_scores": torch.rand(B, 44, generator=torch.Generator().manual_seed(rank))}
        before = torch.cat([p.detach().reshape(-1) for p in pred.Predictor.parameters()]).cpu().clone()
        out = []
        for step in range(2):
            _, ld = pred(bat)
            red = pred._reducer
            # step 0 learns which parameters the graph reaches (every bucket holds a few it does not); from step 1 on the
            # buckets leave while backward is still running
            assert step == 0 or red.launched_in_backward >= len(red.buckets) - 1, "buckets did not leave during backward"
            out.append((ld["loss"], ld["grad_p"], ld["grad_ae"]))
        after = torch.cat([p.detach().reshape(-1) for p in pred.Predictor.parameters()]).cpu()
        bufs = torch.cat([b.detach().float().reshape(-1) for b in pred.AudEnc.buffers() if torch.is_floating_point(b)]).cpu()
        assert not torch.equal(before, after)
        q.put((rank, out, float(after.double().sum()), float(after.double().abs().sum()), float(bufs.double().sum())))
    finally:
        dist.destroy_process_group()


def test_predictor_train_step_two_ranks_gloo_on_one_gpu():
    ctx = torch.multiprocessing.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q), daemon=True) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(2):
            r = q.get(timeout=100)
            res[r[0]] = r[1:]
        for p in procs:
            p.join(30)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    (o0, s0, a0, b0), (o1, s1, a1, b1) = res[0], res[1]
    # different data per rank: different losses; the SAME averaged gradient (norms) and the same parameters afterwards
    assert o0[0][0] != o1[0][0]
    for (l0, gp0, ga0), (l1, gp1, ga1) in zip(o0, o1):
        assert gp0 == gp1 and ga0 == ga1, "ranks clipped different gradients"
    assert s0 == s1 and a0 == a1, "parameters diverged between ranks"
    assert b0 == b1, "BatchNorm buffers were not taken from rank 0"
