"""The CPU oracle against fixtures produced by the imported reference (tools/make_golden.py)."""
import hashlib

import torch

from oracle import probtalk_oracle as O
from probtalk_amd.weights import make_input
from tests.helpers import golden, manifest, ref_state_dict, rel_err

TOL = 2e-5  # oracle vs reference: same torch CPU primitives, only association order differs


def test_pe_table_matches_reference():
    pe = O.sinusoid_table(512, 1800)
    assert pe.dtype == torch.float64 and tuple(pe.shape) == (1801, 512)
    assert hashlib.sha256(pe.numpy().tobytes()).hexdigest() == manifest()["_pe_sha256"]


def test_vqvae_encode_decode():
    g = golden("vq_b2_t88.npz")
    sd = ref_state_dict("VQ")
    x = make_input("vq.x", (2, 376, 88))
    z, q, idx, recon = O.vqvae_forward(sd, x)
    assert rel_err(z, g["z"]) < TOL
    assert torch.equal(idx, g["indices"])
    assert torch.equal(q, g["quantized"])
    assert rel_err(recon, g["recon"]) < TOL
    codes = make_input("vq.codes", (2, 11, 4), "int", low=0, high=128)
    assert torch.equal(O.pq_quantize_all(sd, "vq_layer.", codes), g["quantize_all"])


def test_pq_quantizer_reference_main_shape():
    g = golden("pq_b8_w22.npz")
    sd = ref_state_dict("VQ")
    zq = make_input("pq.z", (8, 512, 22)) * 0.5
    q, idx = O.pq_quantize(sd, "vq_layer.", zq, 4)
    assert idx.dtype == torch.int64 and tuple(idx.shape) == (8, 22, 4)
    assert torch.equal(idx, g["indices"])
    assert torch.equal(q, g["quantized"])


def test_legacy_vq():
    g = golden("legacy_vq_b2_t88.npz")
    sd = ref_state_dict("g")
    x = make_input("lvq.x", (2, 270, 88))
    z, e, idx, recon = O.legacy_vq_forward(sd, x)
    assert torch.equal(idx, g["indices"])
    assert torch.equal(e, g["e"])
    assert rel_err(recon, g["recon"]) < TOL


def _predictor_inputs():
    gt = make_input("pred.gt", (2, 376, 88))
    mask = torch.zeros(2, 1, 88)
    mask[:, :, :30] = 1
    return gt, mask, torch.tensor([1, 3]), make_input("pred.aud", (2, 768, 88))


def test_predictor_condition_and_logits():
    g = golden("predictor_b2_t88.npz")
    gt, mask, label, aud = _predictor_inputs()
    audio = O.condition_encoder(ref_state_dict("AudEnc"), aud)
    assert rel_err(audio, g["audio"]) < TOL
    sd = ref_state_dict("Predictor")
    cond = O.predictor_condition(sd, gt * mask, mask, audio)
    assert rel_err(cond, g["condition"]) < TOL
    ids = make_input("pred.ids", (2, 11, 4), "int", low=0, high=129)
    logits = O.tokens_to_logits(sd, ids, label, cond)
    assert rel_err(logits, g["logits"]) < 1e-4


def test_sample_good_injected_noise():
    g = golden("predictor_b2_t88.npz")
    label = torch.tensor([1, 3])
    noise_q = make_input("pred.noise_q", (8, 2, 44, 128), "exp")
    noise_g = make_input("pred.noise_g", (8, 2, 44), "gumbel")
    codes = O.sample_good(ref_state_dict("Predictor"), label, g["condition"], noise_q, noise_g)
    assert tuple(codes.shape) == (2, 11, 4)
    assert int(codes.max()) < 128
    assert torch.equal(codes.long(), g["codes"].long())


def test_mask_by_random_topk():
    g = golden("topk.npz")
    probs = make_input("topk.probs", (3, 44), "uniform", low=0.01, high=1.0)
    probs[0, 5] = float("inf")
    mlen = torch.tensor([[10], [0], [43]])
    gum = make_input("topk.gumbel", (1, 3, 44), "gumbel")[0]
    conf = torch.log(probs) + 2.25 * gum
    cut = torch.take_along_dim(torch.sort(conf, dim=-1).values, mlen, dim=-1)
    assert torch.equal(conf < cut, g["masking"])


def test_refiner():
    g = golden("refiner_b2_t64.npz")
    x = make_input("ref.x", (2, 376, 64))
    aud = make_input("ref.aud", (2, 768, 64))
    mask = torch.zeros(2, 1, 64)
    mask[:, :, :30] = 1
    out = O.rtransformer(ref_state_dict("HFTransformer"), x, aud, mask, torch.tensor([0, 2]))
    assert rel_err(out, g["out"]) < 1e-4
