"""Golden fixture of one legacy VQ-VAE TRAIN step from the IMPORTED REFERENCE (build container only).

Runs the reference's own VQVAE (nets/spg/vqvae_1d.py:147-173, the model behind `s2g_body_vq`, built with config/vq.json's
values as smplx_body_vq.py:53-55 does) in train mode on generator weights: batch-statistic BatchNorm,
VectorQuantizerEMA's training branch (EMA update, commitment loss, straight-through).  `smplx_body_vq.py` itself cannot
be imported here (data_utils pulls librosa / textgrid), so its loss (get_loss, 170-190: three lines of abs-mean) is
applied by the oracle's restatement `legacy_vq_losses`.  Data only.

    python tools/make_golden_legacy_vq_train.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_loader  # noqa: E402
from oracle import probtalk_oracle_train as OT  # noqa: E402
from probtalk_amd.weights import generate_state_dict, make_input  # noqa: E402
from tests.helpers import LEGACY_VQ_TRAIN_GRAD_KEYS  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    ref_loader.load()
    from nets.spg.vqvae_1d import VQVAE
    B, T = 6, 88
    vq = VQVAE(270, 64, 128, 1024, 2, 512)
    vq.load_state_dict(generate_state_dict(vq.state_dict(), 0, "g."))
    vq.train()
    x = make_input("lvqtrain.x", (B, 270, T))
    e_q_loss, pred = vq(gt_poses=x, result_form="part")
    e_q_loss = e_q_loss.mean()
    rec, parts = OT.legacy_vq_losses(pred, x)
    loss = rec + e_q_loss
    loss.backward()
    sd = vq.state_dict()
    named = dict(vq.named_parameters())
    out = {"loss": loss.detach().numpy(), "e_q_loss": e_q_loss.detach().numpy(), "pred": pred.detach().numpy()[:, :16]}
    for k, v in parts.items():
        out[k] = v.detach().numpy()
    out["embeddings"] = sd["vq_layer.embeddings"].numpy()
    out["ema_cluster_size"] = sd["vq_layer.ema_cluster_size.hidden"].numpy()
    out["ema_dw"] = sd["vq_layer.ema_dw.hidden"].numpy()
    out["enc_project_running_mean"] = sd["encoder.project.norm.running_mean"].numpy()
    out["dec_up3_running_var"] = sd["decoder._up_3.norm.running_var"].numpy()
    for i, k in enumerate(LEGACY_VQ_TRAIN_GRAD_KEYS):
        g = named[k].grad
        out[f"grad_{i}"] = (g[:4] if g.numel() > 20000 else g).numpy()
    out["grad_norm"] = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(p.grad) for p in named.values()
                                                              if p.grad is not None])).numpy()
    np.savez_compressed(os.path.join(OUT, "legacy_vq_train_b6_t88.npz"), **out)
    print("loss", float(loss), "e_q", float(e_q_loss), {k: float(v) for k, v in parts.items()})
    print("codes used:", int((sd["vq_layer.ema_cluster_size.hidden"] > 0).sum()), "of 128")
    print(os.path.getsize(os.path.join(OUT, "legacy_vq_train_b6_t88.npz")), "bytes")


if __name__ == "__main__":
    main()
