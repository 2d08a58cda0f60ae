"""GPU parity of the refiner training path (s2g_body_refiner): masked part-L1 + velocity loss kernel, RTransformer train
forward/backward against the oracle and the reference-generated fixture, TrainWrapper.__call__."""
import math

import pytest
import torch

from oracle import probtalk_oracle_train as OT
from probtalk_amd import train_ops as T
from probtalk_amd import training
from tests.helpers import REFINER_TRAIN_GRAD_KEYS, ref_state_dict, refiner_train_case, rel_err
from tests.test_oracle_refiner_train_golden import load_gold

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,Tn", [(3, 16), (2, 64)])
def test_masked_l1_losses(B, Tn):
    g = torch.Generator().manual_seed(Tn)
    gt = torch.randn(B, 376, Tn, generator=g)
    pred = gt + torch.randn(B, 376, Tn, generator=g)
    mask = torch.zeros(B, 1, Tn)
    mask[0, :, 3:9] = 1
    pr = pred.double().requires_grad_(True)
    tot_ref, parts = OT.refiner_losses(pr, gt.double(), mask.double())
    (tot_ref * 0.7).backward()
    rows = lambda t: torch.cat([t.permute(0, 2, 1).reshape(B * Tn, 376), torch.zeros(B * Tn, 8)], 1).contiguous()
    pd = rows(pred).cuda().requires_grad_(True)
    nmask = (1 - mask).reshape(-1).cuda()
    tot, sums = T.MaskedL1Losses.apply(pd, rows(gt).cuda(), nmask, float(nmask.sum()), B, Tn, [0, 6, 96, 276, 376],
                                       [0.25] * 4, 276)
    (tot * 0.7).backward()
    assert abs(float(tot) - float(tot_ref)) < 1e-5 * float(tot_ref)
    for i, k in enumerate(("face_loss", "body_loss", "hand_loss", "exp_loss", "velocity_loss")):
        assert abs(float(sums[i]) - float(parts[k])) < 1e-5 * float(parts[k]), k
    assert rel_err(pd.grad[:, :376].cpu(), pr.grad.permute(0, 2, 1).reshape(B * Tn, 376)) < 1e-5
    with pytest.raises(RuntimeError, match="no unmasked frame"):
        T.MaskedL1Losses.apply(pd, rows(gt).cuda(), torch.zeros_like(nmask), 0.0, B, Tn, [0, 6, 96, 276, 376], [0.25] * 4, 276)


def _cuda_params(sd):
    out = {}
    for k, v in sd.items():
        t = v.detach().clone().cuda()
        if t.dtype == torch.float32 and "position_encoding" not in k:
            t.requires_grad_(True)
        out[k] = t
    return out


def test_refiner_train_step_matches_oracle_and_fixture():
    sd = ref_state_dict("HFTransformer")
    aud, inp, gt, mask, label = refiner_train_case()
    pre, o_relu = [], torch.relu
    def relu(v):
        pre.append(v.detach())
        return o_relu(v)
    torch.relu = relu
    try:
        loss_o, parts_o, grads_o, pred_o = OT.refiner_train_grads(sd, aud, inp, gt, mask, label)
    finally:
        torch.relu = o_relu
    P = _cuda_params(sd)
    training.relu_trace = []
    try:
        loss, sums, pred = training.refiner_loss(P, aud.cuda(), inp.cuda(), gt.cuda(), mask.cuda(), label.cuda())
        trace = training.relu_trace
    finally:
        training.relu_trace = None
    loss.backward()
    B, _, Tn = aud.shape
    assert abs(float(loss) - float(loss_o)) < 1e-5 * float(loss_o)
    assert rel_err(pred.detach().cpu().reshape(B, Tn, -1)[:, :, :376].permute(0, 2, 1), pred_o) < 1e-4
    # FFN ReLU units that land on the other side of 0 than in the oracle: each must be proven to sit at 0
    flips = 0
    for l in range(6):
        po = pre[l].reshape(-1, 2048)
        differ = (trace[l].cpu() > 0) != (po > 0)
        if bool(differ.any()):
            assert float(po[differ].abs().max()) < 1e-5, l
            flips += int(differ.sum())
    assert flips <= 4
    tol = 5e-4 if flips == 0 else 3e-2
    num = den = 0.0
    worst = ("", 0.0)
    for k, gref in grads_o.items():
        if "position_encoding" in k:
            continue
        got = P[k].grad
        assert got is not None, k
        num += float((got.cpu().double() - gref.double()).pow(2).sum())
        den += float(gref.double().pow(2).sum())
        e = rel_err(got.cpu(), gref)
        if e > worst[1]:
            worst = (k, e)
    assert worst[1] < tol, worst
    assert math.sqrt(num / den) < (1e-4 if flips == 0 else 5e-3)
    gold = load_gold()
    assert abs(float(loss) - float(gold["loss"])) < 1e-5 * float(gold["loss"])
    for i, k in enumerate(REFINER_TRAIN_GRAD_KEYS):
        got = P[k].grad.cpu()
        got = got[:8] if got.numel() > 20000 else got
        assert rel_err(got, gold[f"grad_{i}"]) < tol, k


def _bat(B, T_, seed):
    from probtalk_amd.weights import make_input
    return {"aud_feat": make_input(f"rw.aud{seed}", (B, 768, T_)), "poses": 0.3 * make_input(f"rw.poses{seed}", (B, 330, T_)),
            "expression": 0.3 * make_input(f"rw.expr{seed}", (B, 100, T_)), "speaker": torch.tensor([1, 3, 0, 2][:B]),
            "epoch": 10}


def test_refiner_wrapper_first_step_and_training():
    from oracle import probtalk_oracle as O
    from probtalk_amd import synthetic
    _, ref = synthetic.build_pipeline(seed=0)
    ref.train_dropout = 0.0
    B, T_ = 2, 64
    bat = _bat(B, T_, 0)
    mask = torch.zeros(B, 1, T_)
    mask[0, :, 20:50] = 1
    bat["_mask"], bat["_noise"] = mask, torch.randn(B, 376, T_, generator=torch.Generator().manual_seed(3))
    sd = {k: v.detach().cpu().clone() for k, v in ref.RTransformer.state_dict().items()}
    _, ld = ref(bat)
    gt = torch.cat([bat["poses"][:, :6], bat["poses"][:, ref.c_index], bat["expression"]], 1)
    with torch.no_grad():
        vq = ref.VQ(gt_poses=gt.cuda()).cpu()              # the frozen tokenizer's own (parity-tested) reconstruction
    inp = gt * mask + (vq + 0.05 * bat["_noise"]) * (1 - mask)
    loss_o, parts_o, grads_o, _ = OT.refiner_train_grads(sd, bat["aud_feat"], inp, gt, mask, bat["speaker"])
    assert abs(ld["loss"] - float(loss_o)) < 2e-5 * float(loss_o)
    assert abs(ld["rec_loss"] - float(parts_o["rec_loss"])) < 2e-5 * float(parts_o["rec_loss"])
    n, coef = OT.clip_coef(list(grads_o.values()), 5.0)
    assert abs(ld["grad_HF"] - float(n)) < 5e-3 * float(n)
    new = dict(ref.RTransformer.named_parameters())
    for k in ("output_conv.2.weight", "output_conv.0.bias", "transformer_decoder.layers.5.norm3.weight"):
        gr = grads_o[k]
        want, _, _ = OT.adamw_step(sd[k].double(), gr.double() * coef.double(), 0.0, 0.0, 1, 1e-4)
        sure = gr.abs() > 1e-3 * gr.abs().max()
        move_ref, move = want - sd[k].double(), new[k].detach().cpu().double() - sd[k].double()
        assert float((move - move_ref)[sure].abs().max()) < 0.02 * 1e-4 + 1e-7, k
    # a few more steps with dropout and fresh random masks: finite and decreasing
    ref.train_dropout = 0.1
    bat2 = _bat(4, 180, 1)
    losses = []
    for _ in range(6):
        _, ld = ref(bat2)
        assert math.isfinite(ld["loss"]) and math.isfinite(ld["grad_HF"])
        losses.append(ld["loss"])
    assert min(losses[-2:]) < losses[0], losses
    assert set(ref.state_dict()) == {"HFTransformer", "AudEnc", "generator_optim", "dropout_seeds"}
