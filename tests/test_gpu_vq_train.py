"""GPU parity of the PQ-VAE training path (s2g_body_vqt): EMA codebook update with dead-code re-seeding, commitment loss
+ straight-through estimator, part-weighted smooth-L1 / velocity losses, the whole train forward/backward against the
oracle and the reference-generated fixture, and TrainWrapper.__call__ (Adam) against the oracle's optimizer restatement."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import probtalk_oracle_train as OT
from probtalk_amd import train_ops as T
from probtalk_amd import training
from tests.helpers import VQ_TRAIN_GRAD_KEYS, ref_state_dict, rel_err
from tests.test_oracle_vq_train_golden import load_gold, vq_train_case

pytestmark = pytest.mark.gpu


def _quant_sd(G=4, K=128, D=128, seed=0, warm=True):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for i in range(G):
        sd[f"vq_layer.embeddings_{i}"] = torch.randn(K, D, generator=g)
        sd[f"vq_layer.ema_dw.{i}.hidden"] = torch.randn(K, D, generator=g) * (0.5 if warm else 0.0)
        sd[f"vq_layer.ema_cluster_size.{i}.hidden"] = torch.rand(K, generator=g) * (3.0 if warm else 0.0)
    return sd


@pytest.mark.parametrize("rows,counter,warm", [(132, 1, False), (2816, 7, True), (143, 300, True)])
def test_pq_train_quantize_ema_and_commitment(rows, counter, warm):
    G, K, D = 4, 128, 128
    sd = _quant_sd(seed=rows, warm=warm)
    g = torch.Generator().manual_seed(1)
    W = 11
    z = torch.randn(rows // W, G * D, W, generator=g) * 1.3            # (B, C, W)
    dq = torch.randn(rows // W, G * D, W, generator=g)
    zr = z.clone().requires_grad_(True)
    q_ref, loss_ref, idx_ref, new = OT.pq_train(sd, "vq_layer.", zr, G, counter=counter)
    (q_ref * dq).sum().add(loss_ref * 3.0).backward()
    z_rows = z.permute(0, 2, 1).reshape(-1, G * D).contiguous().cuda().requires_grad_(True)
    cb = torch.stack([sd[f"vq_layer.embeddings_{i}"] for i in range(G)]).cuda()
    q, loss, idx = T.PQQuantizeTrain.apply(z_rows, cb, 0.25)
    (q * dq.permute(0, 2, 1).reshape(-1, G * D).cuda()).sum().add(loss * 3.0).backward()
    assert torch.equal(idx.cpu(), idx_ref)
    assert rel_err(q.detach().cpu(), q_ref.detach().permute(0, 2, 1).reshape(-1, G * D)) < 1e-6
    assert abs(float(loss) - float(loss_ref)) < 1e-5 * float(loss_ref)
    assert rel_err(z_rows.grad.cpu(), zr.grad.permute(0, 2, 1).reshape(-1, G * D)) < 1e-5
    cs = torch.stack([sd[f"vq_layer.ema_cluster_size.{i}.hidden"] for i in range(G)]).cuda()
    dw = torch.stack([sd[f"vq_layer.ema_dw.{i}.hidden"] for i in range(G)]).cuda()
    T.pq_ema_update(z_rows.detach(), idx, cs, dw, cb, 0.99, 1e-5, counter)
    dead = 0
    for i in range(G):
        assert rel_err(cb[i].cpu(), new[f"vq_layer.embeddings_{i}"]) < 1e-5
        assert rel_err(cs[i].cpu(), new[f"vq_layer.ema_cluster_size.{i}.hidden"]) < 1e-5
        assert rel_err(dw[i].cpu(), new[f"vq_layer.ema_dw.{i}.hidden"]) < 1e-5
        dead += int((new[f"vq_layer.ema_cluster_size.{i}.hidden"] == 1.0).sum())
    if rows < 1000:
        assert dead > 0                       # few rows: some codes starve and are re-seeded from the first K latent rows


def test_pq_ema_short_batches_are_tiled():
    """Fewer latent rows than codes: `_tile` (vqvae_modules.py:571-580) re-seeds dead codes from the repeated batch plus
    noise (parity: tests/test_gpu_pq_tile.py).  All-zero rows here: every re-seeded code is pure noise * 0.01 / sqrt(D)."""
    cb = torch.zeros(4, 128, 128, device="cuda")
    noise = torch.randn(4, 128, 128)
    T.pq_ema_update(torch.zeros(64, 512, device="cuda"), torch.zeros(64, 4, dtype=torch.int64, device="cuda"),
                    torch.zeros(4, 128, device="cuda"), torch.zeros(4, 128, 128, device="cuda"), cb, 0.99, 1e-5, 1, tile_noise=noise)
    std = torch.tensor(0.01 / 128 ** 0.5, dtype=torch.float32)
    assert torch.equal(cb[:, 1:].cpu(), (noise * std)[:, 1:])          # code 0 took all 64 rows, codes 1.. are dead


@pytest.mark.parametrize("B,Tn", [(3, 16), (2, 88)])
def test_vq_losses(B, Tn):
    g = torch.Generator().manual_seed(Tn)
    gt = torch.randn(B, Tn, 376, generator=g)
    pred = gt + torch.randn(B, Tn, 376, generator=g) * torch.tensor([0.003, 0.5])[torch.randint(0, 2, (B, Tn, 376), generator=g)]
    pr = pred.double().requires_grad_(True)
    tot_ref, parts = OT.vq_losses(pr, gt.double())
    (tot_ref * 1.3).backward()
    pad = lambda t: torch.cat([t.reshape(B * Tn, 376), torch.zeros(B * Tn, 8)], 1).contiguous()
    pd = pad(pred).cuda().requires_grad_(True)
    tot, sums = T.VQLosses.apply(pd, pad(gt).cuda(), B, Tn, [0, 6, 96, 276, 376], [0.5 - 0.4, 0.4, 0.4, 0.5 - 0.4], 276, 0.01)
    (tot * 1.3).backward()
    assert abs(float(tot) - float(tot_ref)) < 1e-5 * float(tot_ref)
    for i, k in enumerate(("face_loss", "body_loss", "hand_loss", "exp_loss", "velocity_loss")):
        assert abs(float(sums[i]) - float(parts[k])) < 1e-5 * float(parts[k]), k
    assert rel_err(pd.grad[:, :376].cpu(), pr.grad.reshape(B * Tn, 376)) < 1e-5
    assert float(pd.grad[:, 376:].abs().max()) == 0.0


def _cuda_params(sd):
    out = {}
    for k, v in sd.items():
        t = v.detach().clone().cuda()
        if t.dtype == torch.float32 and "running" not in k and "embeddings_" not in k and "hidden" not in k:
            t.requires_grad_(True)
        out[k] = t
    return out


def test_vq_train_step_matches_oracle_and_fixture():
    sd = ref_state_dict("VQ")
    x = vq_train_case()
    B, _, Tn = x.shape
    from oracle import probtalk_oracle as O
    pre, o_leaky, o_relu = [], O._leaky, torch.relu     # the oracle's (Leaky)ReLU inputs in call order
    def leaky(v, slope):
        pre.append(v.detach())
        return o_leaky(v, slope)
    def relu(v):
        pre.append(v.detach())
        return o_relu(v)
    O._leaky, torch.relu = leaky, relu
    try:
        loss_o, parts_o, grads_o, new_o, pred_o, idx_o = OT.vq_train_grads(sd, x)
    finally:
        O._leaky, torch.relu = o_leaky, o_relu
    P = _cuda_params(sd)
    cb = torch.stack([P[f"vq_layer.embeddings_{i}"] for i in range(4)]).contiguous()
    x_rows = T.ncl_to_rows(x.cuda())
    training.relu_trace = []
    try:
        z, e_q, idx, pred, Tout = training.vqvae_train_forward(P, cb, x_rows, B, Tn)
        trace = training.relu_trace
    finally:
        training.relu_trace = None
    rec, sums = T.VQLosses.apply(pred, x_rows, B, Tn, [0, 6, 96, 276, 376], [0.5 - 0.4, 0.4, 0.4, 0.5 - 0.4], 276, 0.01)
    loss = rec + e_q
    loss.backward()
    # (Leaky)ReLU units within rounding of 0 may fall on either side in two correct fp32 implementations; the unit's
    # gradient then differs by its whole slope change, and batch-statistic BatchNorm spreads that over every upstream
    # layer.  Count them, PROVE each one (|pre-activation| < 1e-5 in the oracle), and only then widen the tolerance.
    assert len(pre) == len(trace) == 31
    flips = 0
    for p_o, y in zip(pre, trace):
        Bc, C, L = p_o.shape
        differ = (y.cpu().reshape(Bc, L, -1)[:, :, :C].permute(0, 2, 1) > 0) != (p_o > 0)
        if bool(differ.any()):
            assert float(p_o[differ].abs().max()) < 1e-5
            flips += int(differ.sum())
    assert flips <= 12, flips
    assert Tout == Tn
    # codes are integers: bit-exact unless a near-tie (counted; tests/test_gpu_vqvae.py explains the margin rule)
    mism = int((idx.cpu() != idx_o).sum())
    assert mism <= 2, mism
    assert abs(float(loss) - float(loss_o)) < 2e-5 * float(loss_o)
    assert abs(float(e_q) - float(parts_o["e_q_loss"])) < 2e-5 * float(parts_o["e_q_loss"])
    pred_ncl = pred.detach().cpu().reshape(B, Tn, -1)[:, :, :376].permute(0, 2, 1)
    assert rel_err(pred_ncl, pred_o) < 1e-4
    # gradients: strict when no unit flipped; otherwise all gradients as one vector to 5e-3, every tensor to 5 % of
    # its largest entry (the operator-level tests in this file and tests/test_gpu_train.py hold every kernel to 1e-4)
    num = den = 0.0
    worst = ("", 0.0)
    gmax = max(float(v.abs().max()) for v in grads_o.values())
    for k, gref in grads_o.items():
        got = P[k].grad
        if float(gref.abs().max()) < 1e-6 * gmax:
            assert got is None or float(got.abs().max()) < 1e-5 * gmax, k      # conv bias in front of batch-stat BN
            continue
        assert got is not None, k
        num += float((got.cpu().double() - gref.double()).pow(2).sum())
        den += float(gref.double().pow(2).sum())
        e = rel_err(got.cpu(), gref)
        if e > worst[1]:
            worst = (k, e)
    assert math.sqrt(num / den) < (1e-4 if flips == 0 else 5e-3), math.sqrt(num / den)
    assert worst[1] < (5e-4 if flips == 0 else 5e-2), worst
    # fixture from the imported reference
    gold = load_gold()
    assert abs(float(loss) - float(gold["loss"])) < 2e-5 * float(gold["loss"])
    assert rel_err(pred_ncl[:, :, :16], gold["pred"]) < 1e-4
    for i, k in enumerate(VQ_TRAIN_GRAD_KEYS):
        got = P[k].grad.cpu()
        got = got[:4] if got.numel() > 20000 else got
        if float(np.abs(gold[f"grad_{i}"]).max()) < 1e-6 * gmax:    # conv bias in front of batch-stat BN: exact gradient
            assert float(got.abs().max()) < 1e-5 * gmax, k          # is 0, both sides hold rounding noise only
            continue
        assert rel_err(got, gold[f"grad_{i}"]) < (5e-4 if flips == 0 else 5e-2), k
    # EMA side effects and BatchNorm running statistics
    cs = torch.stack([P[f"vq_layer.ema_cluster_size.{i}.hidden"] for i in range(4)]).contiguous()
    dw = torch.stack([P[f"vq_layer.ema_dw.{i}.hidden"] for i in range(4)]).contiguous()
    T.pq_ema_update(z.detach(), idx, cs, dw, cb, 0.99, 1e-5, 1)
    if mism == 0:
        for i in range(4):
            assert rel_err(cb[i].cpu(), gold[f"embeddings_{i}"]) < 1e-4
            assert rel_err(cs[i].cpu(), gold[f"ema_cluster_size_{i}"]) < 1e-5
            assert rel_err(dw[i].cpu(), gold[f"ema_dw_{i}"]) < 1e-4
    assert rel_err(P["encoder.project.norm.running_mean"].cpu(), gold["enc_project_running_mean"]) < 1e-5
    assert rel_err(P["decoder._up_4.norm.running_var"].cpu(), gold["dec_up4_running_var"]) < 1e-4


def _bat(B, T_, seed):
    from probtalk_amd.weights import make_input
    return {"aud_feat": make_input(f"vt.aud{seed}", (B, 64, T_)), "poses": 0.5 * make_input(f"vt.poses{seed}", (B, 330, T_)),
            "expression": 0.5 * make_input(f"vt.expr{seed}", (B, 100, T_)), "speaker": torch.zeros(B, dtype=torch.long),
            "epoch": 0}


def test_vq_teacher_wrapper_trains():
    from probtalk_amd import nets, synthetic
    from probtalk_amd.config import default_transformer_config
    from probtalk_amd.weights import fill_module
    cfg = default_transformer_config(vq_path=None)
    w = nets.init_model("s2g_body_vqt", synthetic.args(infer=False), cfg)
    fill_module(w.VQ, 0, "VQ.")
    bat = _bat(16, 88, 0)
    e0 = w.VQ.vq_layer.embeddings_0.clone()
    losses = []
    for _ in range(8):
        _, ld = w(bat)
        total = 0.1 * (ld["face_loss"] + ld["exp_loss"]) + 0.4 * (ld["body_loss"] + ld["hand_loss"]) + ld["velocity_loss"] + ld["e_q_loss"]
        assert math.isfinite(total)
        losses.append(total)
    assert ld["grad"] == 0.0 and set(ld) >= {"face_loss", "body_loss", "hand_loss", "exp_loss", "velocity_loss", "e_q_loss"}
    assert min(losses[-2:]) < losses[0], losses
    assert w.VQ.vq_layer.ema_dw[0].counter == 8
    assert not torch.equal(w.VQ.vq_layer.embeddings_0.cpu(), e0.cpu())
    sd = w.state_dict()
    assert set(sd) == {"VQ", "generator_optim"} and "vq_layer.ema_dw.0.hidden" in sd["VQ"]
    # eval path sees the trained weights and codebooks
    w.args.infer = True
    out, _ = w.infer_on_audio(bat["poses"].new_zeros(1, 376, 88).cuda(), None, None, None, 1)
    assert out.shape == (1, 376, 88) and torch.isfinite(out).all()
