"""Generate tests/golden/* from the IMPORTED REFERENCE (build container only).

Runs the reference's own modules (via tools/ref_loader.py shims) on weights
from probtalk_amd.weights and inputs from probtalk_amd.weights.make_input, and
writes small I/O fixtures.  Fixtures are data only: generated inputs are NOT
stored (they are regenerated from their names), outputs are.

    python tools/make_golden.py
"""
import contextlib
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_loader  # noqa: E402
from probtalk_amd.weights import generate_state_dict, make_input  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED = 0


def manifest(sd):
    return {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()}


@contextlib.contextmanager
def cpu_device_shims(noise_q, noise_g):
    """Let sample_good (which hard-codes device="cuda", PTransformer.py:597,697) run on CPU
    and replace the two RNG draws by injected noise.  Patches torch, not the reference."""
    o_zeros, o_to, o_multi = torch.zeros, torch.Tensor.to, torch.multinomial
    o_gumbel = torch.distributions.gumbel.Gumbel.sample
    state = {"q": 0, "g": 0}

    def zeros(*a, **k):
        if str(k.get("device", "")).startswith("cuda"):
            k["device"] = "cpu"
        return o_zeros(*a, **k)

    def to(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, str) and x.startswith("cuda")) else x for x in a)
        return o_to(self, *a, **k)

    def multinomial(probs, n, replacement=False, **k):
        assert n == 1
        q = noise_q[state["q"]].reshape(probs.shape)
        state["q"] += 1
        return torch.argmax(probs / q, dim=-1, keepdim=True)

    def gumbel_sample(self, shape=torch.Size()):
        g = noise_g[state["g"]].reshape(tuple(shape))
        state["g"] += 1
        return g

    torch.zeros, torch.Tensor.to, torch.multinomial = zeros, to, multinomial
    torch.distributions.gumbel.Gumbel.sample = gumbel_sample
    try:
        yield
    finally:
        torch.zeros, torch.Tensor.to, torch.multinomial = o_zeros, o_to, o_multi
        torch.distributions.gumbel.Gumbel.sample = o_gumbel


def main():
    os.makedirs(OUT, exist_ok=True)
    ref_loader.load()
    from nets.inpainting.vqvae_1d_sc import VQVAE_SC, ConditionEncoder
    from nets.inpainting.predictornet import PredictorNet
    from nets.inpainting.RTransformer import RTransformer
    from nets.spg.vqvae_1d import VQVAE

    torch.set_grad_enabled(False)
    man = {}

    # ---------------- PQ-VAE (vq_teacher.py:64-67 constructor arguments) ----------------
    vq = VQVAE_SC(376, 512, 128, 1024, 2, 512, q_type="pro", groups=4, share_code=False).eval()
    man["VQ"] = manifest(vq.state_dict())
    vq.load_state_dict(generate_state_dict(vq.state_dict(), SEED, "VQ."))
    x = make_input("vq.x", (2, 376, 88))
    z, qo, _ = vq.encode(x)
    recon = vq.decode(qo.quantized, None)
    codes = make_input("vq.codes", (2, 11, 4), "int", low=0, high=128)
    qall = vq.vq_layer.quantize_all(codes)
    np.savez_compressed(os.path.join(OUT, "vq_b2_t88.npz"), z=z.numpy(), quantized=qo.quantized.numpy(),
                        indices=qo.loss.numpy(), recon=recon.numpy(), quantize_all=qall.numpy())
    # standalone quantizer at the reference's own __main__ shape (vqvae_modules.py:683-689), non-shared
    zq = make_input("pq.z", (8, 512, 22)) * 0.5
    qo2 = vq.vq_layer(zq)
    np.savez_compressed(os.path.join(OUT, "pq_b8_w22.npz"), quantized=qo2.quantized.numpy(),
                        indices=qo2.loss.numpy())

    # ---------------- legacy single-codebook VQ (smplx_body_vq.py:53-55 with vq.json values) ----------------
    lvq = VQVAE(270, 64, 128, 1024, 2, 512).eval()
    man["g"] = manifest(lvq.state_dict())
    lvq.load_state_dict(generate_state_dict(lvq.state_dict(), SEED, "g."))
    xl = make_input("lvq.x", (2, 270, 88))
    e, lrec = lvq(xl)
    _, lat = lvq.encode(xl)
    np.savez_compressed(os.path.join(OUT, "legacy_vq_b2_t88.npz"), e=e.numpy(), indices=lat.numpy(),
                        recon=lrec.numpy())

    # ---------------- audio condition encoder (predictor.py:67-68) ----------------
    ae = ConditionEncoder(in_dim=768, num_hiddens=256, num_residual_layers=2, num_residual_hiddens=256).eval()
    man["AudEnc"] = manifest(ae.state_dict())
    ae.load_state_dict(generate_state_dict(ae.state_dict(), SEED, "AudEnc."))
    aud = make_input("pred.aud", (2, 768, 88))
    audio = ae(aud)

    # ---------------- predictor network (predictor.py:72-76 with transformer.json) ----------------
    pn = PredictorNet(False, 376, 128, 512, 10, 6, 4, groups=4, identity=True, maskgit=True, maskgit_T=8,
                      transformer=True, text=None, audio="wav2vec", motion_context=True).eval()
    man["Predictor"] = manifest(pn.state_dict())
    pn.load_state_dict(generate_state_dict(pn.state_dict(), SEED, "Predictor."))
    gt = make_input("pred.gt", (2, 376, 88))
    mask = torch.zeros(2, 1, 88)
    mask[:, :, :30] = 1
    label = torch.tensor([1, 3])
    # the motion-context condition: replay PredictorNet.forward's first half through its own sub-modules
    mm = pn.motion_embedding(torch.cat([gt * mask, mask], dim=1))
    for layer in pn.enc:
        mm = layer(mm)
    cond = torch.cat([mm, audio], dim=1)
    ids = make_input("pred.ids", (2, 11, 4), "int", low=0, high=129)
    logits = pn.decoder.tokens_to_logits(ids, label, cond)
    noise_q = make_input("pred.noise_q", (8, 2, 44, 128), "exp")
    noise_g = make_input("pred.noise_g", (8, 2, 44), "gumbel")
    with cpu_device_shims(noise_q, noise_g):
        codes_out = pn(gt * mask, torch.zeros(2, 11, 4), mask, label, audio, None, 1)
    np.savez_compressed(os.path.join(OUT, "predictor_b2_t88.npz"), audio=audio.numpy(), condition=cond.numpy(),
                        logits=logits.numpy(), codes=codes_out.numpy())
    # mask_by_random_topk alone (PTransformer.py:696-704)
    probs = make_input("topk.probs", (3, 44), "uniform", low=0.01, high=1.0)
    probs[0, 5] = float("inf")
    mlen = torch.tensor([[10.0], [0.0], [43.0]])
    g1 = make_input("topk.gumbel", (1, 3, 44), "gumbel")
    with cpu_device_shims(None, g1):
        masking = pn.decoder.mask_by_random_topk(mlen, probs, temperature=2.25)
    np.savez_compressed(os.path.join(OUT, "topk.npz"), masking=masking.numpy())

    # ---------------- refiner network (refiner.py:69-70) ----------------
    rt = RTransformer(376, 768, 512, 6, use_label=True, motion_context=True).eval()
    man["HFTransformer"] = manifest(rt.state_dict())
    rt.load_state_dict(generate_state_dict(rt.state_dict(), SEED, "HFTransformer."))
    xr = make_input("ref.x", (2, 376, 64))
    audr = make_input("ref.aud", (2, 768, 64))
    maskr = torch.zeros(2, 1, 64)
    maskr[:, :, :30] = 1
    out = rt(xr, audr, maskr, torch.tensor([0, 2]))
    np.savez_compressed(os.path.join(OUT, "refiner_b2_t64.npz"), out=out.numpy())

    # PE table: store a hash + a few rows
    pe = rt.state_dict()["PE.position_encoding.weight"]
    man["_pe_sha256"] = hashlib.sha256(pe.numpy().tobytes()).hexdigest()
    man["_param_counts"] = {"VQ": sum(p.numel() for p in vq.parameters()),
                            "Predictor": sum(p.numel() for p in pn.parameters()),
                            "AudEnc": sum(p.numel() for p in ae.parameters()),
                            "HFTransformer": sum(p.numel() for p in rt.parameters())}
    with open(os.path.join(OUT, "state_dict_manifest.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)
    print("golden fixtures written to", OUT)
    for fn in sorted(os.listdir(OUT)):
        print(f"  {fn}: {os.path.getsize(os.path.join(OUT, fn))} B")


if __name__ == "__main__":
    main()
