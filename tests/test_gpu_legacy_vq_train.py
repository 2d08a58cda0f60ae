"""GPU parity of the legacy VQ-VAE training path (s2g_body_vq, the wrapper the north star names): single-codebook EMA update
without re-seeding, the whole train forward/backward against the oracle and the reference-generated fixture, and
TrainWrapper.__call__ (AdamW) against the oracle's optimizer restatement."""
import math

import pytest
import torch

from oracle import probtalk_oracle_train as OT
from probtalk_amd import train_ops as T
from probtalk_amd import training
from tests.helpers import LEGACY_VQ_TRAIN_GRAD_KEYS, ref_state_dict, rel_err
from tests.test_oracle_legacy_vq_train_golden import legacy_vq_train_case, load_gold

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,K,counter,warm", [(132, 128, 1, False), (2816, 128, 7, True), (4400, 2048, 3, True)])
def test_vq_ema_update(rows, K, counter, warm):
    """pt_train_vq_ema against VectorQuantizerEMA's update (no re-seeding; K = 2048 is config/body_vq.json's code_num)."""
    D = 64
    g = torch.Generator().manual_seed(rows)
    sd = {"vq_layer.embeddings": torch.randn(K, D, generator=g),
          "vq_layer.ema_dw.hidden": torch.randn(K, D, generator=g) * (0.5 if warm else 0.0),
          "vq_layer.ema_cluster_size.hidden": torch.rand(K, generator=g) * (3.0 if warm else 0.0)}
    W = 11
    z = torch.randn(rows // W, D, W, generator=g) * 1.3
    _, _, idx_ref, new = OT.vq_ema_train(sd, "vq_layer.", z, counter=counter)
    z_rows = z.permute(0, 2, 1).reshape(-1, D).contiguous().cuda()
    cb = sd["vq_layer.embeddings"].cuda()
    q, loss, idx = T.PQQuantizeTrain.apply(z_rows, cb.unsqueeze(0), 0.25)
    assert torch.equal(idx.cpu().reshape(-1), idx_ref)
    cs, dw = sd["vq_layer.ema_cluster_size.hidden"].cuda(), sd["vq_layer.ema_dw.hidden"].cuda()
    T.vq_ema_update(z_rows, idx, cs, dw, cb, 0.99, 1e-5, counter)
    assert rel_err(cs.cpu(), new["vq_layer.ema_cluster_size.hidden"]) < 1e-5
    assert rel_err(dw.cpu(), new["vq_layer.ema_dw.hidden"]) < 1e-5
    ref = new["vq_layer.embeddings"]
    fin = torch.isfinite(ref)                       # a code never used so far divides 0 by the smoothed 0: both sides agree on it
    assert torch.equal(torch.isfinite(cb.cpu()), fin)
    assert rel_err(cb.cpu()[fin], ref[fin]) < 1e-5


def _cuda_params(sd):
    out = {}
    for k, v in sd.items():
        t = v.detach().clone().cuda()
        if t.dtype == torch.float32 and "running" not in k and "embeddings" not in k and "hidden" not in k:
            t.requires_grad_(True)
        out[k] = t
    return out


def test_legacy_vq_train_step_matches_oracle_and_fixture(tap_kernel):
    sd = ref_state_dict("g")
    x = legacy_vq_train_case()
    B, _, Tn = x.shape
    loss_o, parts_o, grads_o, new_o, pred_o, idx_o = OT.legacy_vq_train_grads(sd, x)
    P = _cuda_params(sd)
    x_rows = T.ncl_to_rows(x.cuda())
    z, e_q, idx, pred, Tout = training.legacy_vq_train_forward(P, P["vq_layer.embeddings"], x_rows, B, Tn)
    ones = torch.ones(B * Tn, device="cuda")
    rec, sums = T.MaskedL1Losses.apply(pred, x_rows, ones, float(B * Tn), B, Tn, [0, 270], [1.0], 270)
    loss = rec + e_q
    loss.backward()
    assert Tout == Tn
    mism = int((idx.cpu().reshape(-1) != idx_o).sum())
    assert mism <= 1, mism                       # codes are integers: bit-exact unless a near-tie (tests/test_gpu_vqvae.py)
    assert abs(float(loss) - float(loss_o)) < 2e-5 * float(loss_o)
    assert abs(float(sums[0]) - float(parts_o["rec_loss"])) < 2e-5 * float(parts_o["rec_loss"])
    assert abs(float(sums[1]) - float(parts_o["velocity_loss"])) < 2e-5 * float(parts_o["velocity_loss"])
    assert abs(float(e_q) - float(parts_o["e_q_loss"])) < 2e-5 * float(parts_o["e_q_loss"])
    pred_ncl = pred.detach().cpu().reshape(B, Tn, -1)[:, :, :270].permute(0, 2, 1)
    assert rel_err(pred_ncl, pred_o) < 1e-4
    # gradients: all tensors as one vector, and each tensor on its own.  A (Leaky)ReLU unit within rounding of 0 may fall on
    # either side in two correct fp32 implementations (tests/test_gpu_vq_train.py proves that case unit by unit for the
    # PQ-VAE); the bounds here are the ones that test uses when a unit flipped.
    num = den = 0.0
    gmax = max(float(v.abs().max()) for v in grads_o.values())
    worst = ("", 0.0)
    for k, gref in grads_o.items():
        got = P[k].grad
        if float(gref.abs().max()) < 1e-6 * gmax:
            assert got is None or float(got.abs().max()) < 1e-5 * gmax, k      # conv bias in front of batch-stat BN
            continue
        assert got is not None, k
        num += float((got.cpu().double() - gref.double()).pow(2).sum())
        den += float(gref.double().pow(2).sum())
        e = rel_err(got.cpu(), gref)
        worst = (k, e) if e > worst[1] else worst
    assert math.sqrt(num / den) < tap_kernel.bound(5e-3, 1e-2), math.sqrt(num / den)
    assert worst[1] < tap_kernel.bound(5e-2, 1e-1), worst          # one flipped unit owns one row of a conv's gradient
    gold = load_gold()
    assert abs(float(loss) - float(gold["loss"])) < 2e-5 * float(gold["loss"])
    assert rel_err(pred_ncl[:, :16], gold["pred"]) < 1e-4
    for i, k in enumerate(LEGACY_VQ_TRAIN_GRAD_KEYS):
        got = P[k].grad.cpu()
        got = got[:4] if got.numel() > 20000 else got
        assert rel_err(got, gold[f"grad_{i}"]) < 5e-2, k
    cs, dw, cb = P["vq_layer.ema_cluster_size.hidden"], P["vq_layer.ema_dw.hidden"], P["vq_layer.embeddings"]
    T.vq_ema_update(z.detach(), idx, cs, dw, cb, 0.99, 1e-5, 1)
    if mism == 0:
        assert rel_err(cs.cpu(), gold["ema_cluster_size"]) < 1e-5
        assert rel_err(dw.cpu(), gold["ema_dw"]) < 1e-4
        fin = torch.isfinite(gold["embeddings"])
        assert rel_err(cb.cpu()[fin], gold["embeddings"][fin]) < 1e-4
    assert rel_err(P["encoder.project.norm.running_mean"].cpu(), gold["enc_project_running_mean"]) < 1e-5
    assert rel_err(P["decoder._up_3.norm.running_var"].cpu(), gold["dec_up3_running_var"]) < 1e-4


def test_legacy_vq_wrapper_step_matches_oracle_adamw():
    """s2g_body_vq.__call__ (non-composition, config/vq.json): one step = the oracle's gradients through AdamW
    (lr 1e-4, betas 0.9/0.999, weight decay 1e-2), then the loss falls over a few steps."""
    from probtalk_amd import nets, synthetic
    from probtalk_amd.config import default_transformer_config
    from probtalk_amd.weights import fill_module, make_input
    cfg = default_transformer_config(vq_path=None)
    cfg.Model.composition = False
    w = nets.init_model("s2g_body_vq", synthetic.args(infer=False), cfg)
    fill_module(w.g, 0, "g.")
    sd0 = {k: v.detach().cpu().clone() for k, v in w.g.state_dict().items()}
    B, Tn = 6, 88
    poses = 0.5 * make_input("lvt.poses", (B, 330, Tn))
    bat = {"aud_feat": torch.zeros(B, 64, Tn), "poses": poses, "expression": torch.zeros(B, 100, Tn),
           "speaker": torch.zeros(B, dtype=torch.long), "epoch": 0}
    x = poses[:, w.c_index, :]
    assert x.shape[1] == 270
    loss_o, parts_o, grads_o, new_o, _, _ = OT.legacy_vq_train_grads(sd0, x)
    _, ld = w(bat)
    assert set(ld) == {"grec_loss", "gvelocity_loss", "ge_q_loss"}
    assert abs(ld["grec_loss"] - float(parts_o["rec_loss"])) < 2e-5 * float(parts_o["rec_loss"])
    assert abs(ld["ge_q_loss"] - float(parts_o["e_q_loss"])) < 2e-5 * float(parts_o["e_q_loss"])
    sd1 = w.g.state_dict()
    for k in ("decoder.project.weight", "encoder.pre_vq_conv.bias", "decoder._dec_2.norm.weight"):
        p_ref, _, _ = OT.adamw_step(sd0[k], grads_o[k], torch.zeros_like(sd0[k]), torch.zeros_like(sd0[k]), 1, 1e-4,
                                    betas=(0.9, 0.999), wd=1e-2)
        # one AdamW step moves every weight by ~lr whatever the gradient's size: compare the MOVE, not the weight
        assert rel_err(sd1[k].cpu() - sd0[k], p_ref - sd0[k]) < 2e-2, k
    assert w.g.vq_layer.ema_dw.counter == 1
    assert rel_err(sd1["vq_layer.ema_cluster_size.hidden"].cpu(), new_o["vq_layer.ema_cluster_size.hidden"]) < 1e-5
    first = ld["grec_loss"] + ld["gvelocity_loss"] + ld["ge_q_loss"]
    for _ in range(6):
        _, ld = w(bat)
    last = ld["grec_loss"] + ld["gvelocity_loss"] + ld["ge_q_loss"]
    assert math.isfinite(last) and last < first, (first, last)
    assert set(w.state_dict()) == {"g", "g_optim", "discriminator", "discriminator_optim"}


def test_legacy_vq_wrapper_composition_trains_two_models():
    """config/body_vq.json style (`composition` true): separate body (90-d) and hand (180-d) models, one step each."""
    from probtalk_amd import nets, synthetic
    from probtalk_amd.config import default_transformer_config
    from probtalk_amd.weights import fill_module, make_input
    cfg = default_transformer_config(vq_path=None)
    cfg.Model.composition = True
    w = nets.init_model("s2g_body_vq", synthetic.args(infer=False), cfg)
    fill_module(w.g_body, 1, "g.")
    fill_module(w.g_hand, 2, "g.")
    B, Tn = 8, 64
    bat = {"aud_feat": torch.zeros(B, 64, Tn), "poses": 0.5 * make_input("lvt.cposes", (B, 330, Tn)),
           "expression": torch.zeros(B, 100, Tn), "speaker": torch.zeros(B, dtype=torch.long), "epoch": 0}
    first = None
    for _ in range(5):
        _, ld = w(bat)
        tot = sum(ld.values())
        assert math.isfinite(tot)
        first = tot if first is None else first
    assert set(ld) == {"brec_loss", "bvelocity_loss", "be_q_loss", "hrec_loss", "hvelocity_loss", "he_q_loss"}
    assert tot < first
    assert w.g_body.vq_layer.ema_dw.counter == 5 and w.g_hand.vq_layer.ema_dw.counter == 5
