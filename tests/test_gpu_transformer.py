"""GPU parity: predictor (motion-context encoder, tokens_to_logits, MaskGIT sampler) and refiner HIP
paths vs the CPU oracle and the reference's golden fixtures, through the C ABI.

Tolerance (BASELINE.json north_star): predictor logits / decoded motion within 1e-4 relative error in fp32.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import probtalk_oracle as O
from probtalk_amd.weights import make_input
from tests.helpers import golden, ref_state_dict, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _predictor():
    from probtalk_amd.nets.transformer_modules import PredictorNet
    m = PredictorNet(False, 376, 128, 512, 10, 6, 4, groups=4, identity=True, maskgit=True, maskgit_T=8,
                     transformer=True, text=None, audio="wav2vec", motion_context=True).eval()
    m.load_state_dict(ref_state_dict("Predictor"))
    return m


def _refiner():
    from probtalk_amd.nets.transformer_modules import RTransformer
    m = RTransformer(376, 768, 512, 6, use_label=True, motion_context=True).eval()
    m.load_state_dict(ref_state_dict("HFTransformer"))
    return m


@pytest.mark.parametrize("B,Nq,Nk", [(2, 44, 44), (3, 148, 37), (1, 300, 300), (2, 33, 65), (1, 1, 1)])
def test_attention_op(B, Nq, Nk):
    from probtalk_amd import _lib
    g = torch.Generator().manual_seed(Nq * 7 + Nk)
    q = torch.randn(B * Nq, 512, generator=g)
    kv = torch.randn(B * Nk, 1024, generator=g) * 1.5
    qh = q.reshape(B, Nq, 8, 64).transpose(1, 2)
    kh = kv[:, :512].reshape(B, Nk, 8, 64).transpose(1, 2)
    vh = kv[:, 512:].reshape(B, Nk, 8, 64).transpose(1, 2)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) / 8.0, -1) @ vh).transpose(1, 2).reshape(B * Nq, 512)
    qd, kvd = q.cuda(), kv.cuda()
    out = torch.full((B * Nq, 512), float("nan"), device="cuda")
    _lib.check(_lib.lib().pt_attention_cl(_lib.ptr(qd), 512, _lib.ptr(kvd), 1024,
                                          _lib.c_fp(kvd.data_ptr() + 512 * 4), 1024, _lib.ptr(out), 512, B, Nq, Nk, 8,
                                          _lib.stream()))
    assert rel_err(out.cpu(), ref) < 2e-6


def _pred_inputs():
    gt = make_input("pred.gt", (2, 376, 88))
    mask = torch.zeros(2, 1, 88)
    mask[:, :, :30] = 1
    return gt, mask, torch.tensor([1, 3])


def test_predictor_condition_and_logits_golden():
    g = golden("predictor_b2_t88.npz")
    pn = _predictor()
    gt, mask, label = _pred_inputs()
    cond = pn.condition((gt * mask).cuda(), mask.cuda(), g["audio"].cuda())
    assert rel_err(cond.cpu(), g["condition"]) < TOL
    ids = make_input("pred.ids", (2, 11, 4), "int", low=0, high=129)
    logits = pn.decoder.tokens_to_logits(ids.cuda(), label.cuda(), g["condition"].cuda())
    assert tuple(logits.shape) == (2, 11, 4, 128)
    assert rel_err(logits.cpu(), g["logits"]) < TOL


def test_predictor_sample_good_golden_codes():
    """whole MaskGIT loop with the reference's injected noise reproduces the reference's codes."""
    g = golden("predictor_b2_t88.npz")
    pn = _predictor()
    _, _, label = _pred_inputs()
    noise_q = make_input("pred.noise_q", (8, 2, 44, 128), "exp")
    noise_g = make_input("pred.noise_g", (8, 2, 44), "gumbel")
    codes, logits = pn.decoder.sample_good(label.cuda(), g["condition"].cuda(), None, T=8,
                                           noise=(noise_q.cuda(), noise_g.cuda()), return_logits=True)
    assert codes.dtype == torch.int32 and tuple(codes.shape) == (2, 11, 4)
    ids_ref, logits_ref = O.sample_good(ref_state_dict("Predictor"), label, g["condition"], noise_q, noise_g,
                                        return_logits=True)
    for t in range(8):
        assert rel_err(logits[t].cpu(), logits_ref[t]) < TOL, f"pass {t}"
    assert torch.equal(codes.cpu(), ids_ref)
    assert torch.equal(codes.cpu().long(), g["codes"].long())
    # end to end through PredictorNet.forward
    gt, mask, _ = _pred_inputs()
    codes2 = pn((gt * mask).cuda(), None, mask.cuda(), label.cuda(), g["audio"].cuda(), None, 1,
                noise=(noise_q.cuda(), noise_g.cuda()))
    assert torch.equal(codes2, codes)


def test_predictor_t300_vs_oracle():
    """config-3 clip geometry (T=300 -> N=148 tokens, memory 37) at a batch the oracle finishes quickly."""
    pn = _predictor()
    sd = ref_state_dict("Predictor")
    B, T = 2, 300
    gt = make_input("p300.gt", (B, 376, T))
    mask = torch.zeros(B, 1, T)
    mask[:, :, :30] = 1
    audio = make_input("p300.audio", (B, 256, 37))
    label = torch.tensor([0, 2])
    cond_ref = O.predictor_condition(sd, gt * mask, mask, audio)
    cond = pn.condition((gt * mask).cuda(), mask.cuda(), audio.cuda())
    assert tuple(cond.shape) == (B, 768, 37)
    assert rel_err(cond.cpu(), cond_ref) < TOL
    ids = make_input("p300.ids", (B, 37, 4), "int", low=0, high=129)
    logits = pn.decoder.tokens_to_logits(ids.cuda(), label.cuda(), cond_ref.cuda())
    assert rel_err(logits.cpu(), O.tokens_to_logits(sd, ids, label, cond_ref)) < TOL


def test_sample_good_default_noise_properties():
    """with on-device noise: every token resolved, codes in range, different draws differ."""
    g = golden("predictor_b2_t88.npz")
    pn = _predictor()
    label = torch.tensor([1, 3]).cuda()
    torch.manual_seed(0)
    a = pn.decoder.sample_good(label, g["condition"].cuda(), None, T=8)
    b = pn.decoder.sample_good(label, g["condition"].cuda(), None, T=8)
    assert int(a.min()) >= 0 and int(a.max()) < 128 and int(b.max()) < 128
    assert not torch.equal(a, b)


def test_refiner_golden_and_t300():
    g = golden("refiner_b2_t64.npz")
    rt = _refiner()
    x = make_input("ref.x", (2, 376, 64))
    aud = make_input("ref.aud", (2, 768, 64))
    mask = torch.zeros(2, 1, 64)
    mask[:, :, :30] = 1
    out = rt(x.cuda(), aud.cuda(), mask.cuda(), torch.tensor([0, 2]).cuda())
    assert rel_err(out.cpu(), g["out"]) < TOL
    sd = ref_state_dict("HFTransformer")
    B, T = 2, 300
    x3 = make_input("ref.x300", (B, 376, T))
    a3 = make_input("ref.a300", (B, 768, T))
    m3 = torch.zeros(B, 1, T)
    m3[:, :, 100:130] = 1
    lab = torch.tensor([3])                       # a single label broadcasts over the batch
    out3 = rt(x3.cuda(), a3.cuda(), m3.cuda(), lab.cuda())
    assert rel_err(out3.cpu(), O.rtransformer(sd, x3, a3, m3, lab)) < TOL


@pytest.mark.parametrize("Tin,Tout", [(296, 300), (88, 88), (176, 180), (8, 30), (300, 296)])
def test_interp_blend(Tin, Tout):
    from probtalk_amd.nets.transformer_modules import interp_blend
    pred = make_input("ib.pred", (3, 376, Tin))
    gt = make_input("ib.gt", (3, 376, Tout))
    mask = (make_input("ib.mask", (3, 1, Tout), "uniform") > 0.6).float()
    ref = F.interpolate(pred, size=Tout, mode="linear")
    assert rel_err(interp_blend(pred.cuda(), Tout).cpu(), ref) < 2e-6
    out = interp_blend(pred.cuda(), Tout, gt.cuda(), mask.cuda(), premask=True)
    assert rel_err(out.cpu(), (gt * mask) * mask + ref * (1 - mask)) < 2e-6
    out = interp_blend(pred.cuda(), Tout, gt.cuda(), mask.cuda(), premask=False)
    assert rel_err(out.cpu(), gt * mask + ref * (1 - mask)) < 2e-6


@pytest.mark.parametrize("T", [300, 30, 7])
def test_interp_blend_streaming_and_odd_lengths(T):
    """T_in == T_out and a blend without prediction: the 16-byte streaming kernel (T % 4 == 0) and the per-frame kernel
    (odd lengths) give the exact blend."""
    from probtalk_amd.nets.transformer_modules import interp_blend
    pred = make_input("ib2.pred", (5, 376, T))
    gt = make_input("ib2.gt", (5, 376, T))
    mask = (make_input("ib2.mask", (5, 1, T), "uniform") > 0.5).float()
    assert torch.equal(interp_blend(pred.cuda(), T).cpu(), pred)
    assert torch.equal(interp_blend(pred.cuda(), T, gt.cuda(), mask.cuda(), premask=False).cpu(), gt * mask + pred * (1 - mask))
    assert torch.equal(interp_blend(None, T, gt.cuda(), mask.cuda(), premask=False).cpu(), gt * mask)
    assert torch.equal(interp_blend(pred.cuda(), T, gt.cuda(), mask.cuda(), premask=True).cpu(), (gt * mask) * mask + pred * (1 - mask))


def test_maskgit_schedule_edge_cases():
    """one-pass decoding (T=1) resolves everything; ties in confidence keep reference semantics."""
    g = golden("predictor_b2_t88.npz")
    pn = _predictor()
    label = torch.tensor([1, 3])
    nq = make_input("edge.q", (1, 2, 44, 128), "exp")
    ng = make_input("edge.g", (1, 2, 44), "gumbel")
    codes = pn.decoder.sample_good(label.cuda(), g["condition"].cuda(), None, T=1, noise=(nq.cuda(), ng.cuda()))
    ref = O.sample_good(ref_state_dict("Predictor"), label, g["condition"], nq, ng, T=1)
    assert torch.equal(codes.cpu(), ref)
    assert int(codes.max()) < 128
