"""Golden fixture of one PQ-VAE TRAIN step from the IMPORTED REFERENCE (build container only).

Runs the reference's own VQVAE_SC in train mode (vq_teacher.py:167: `self.VQ(gt_poses, result_form='full')`) on
generator weights: batch-statistic BatchNorm, ProductQuantization's training branch with its EMA codebook update and
dead-code re-seeding, straight-through decode.  `vq_teacher.py` itself cannot be imported here (it pulls librosa /
textgrid through data_utils), so its loss (get_loss, 184-232: a dozen lines of F.smooth_l1_loss / abs-mean) is applied by
the oracle's restatement `vq_losses`; what this fixture pins is the module's train forward, the EMA side effects and the
gradients autograd derives from them.  Data only.

    python tools/make_golden_vq_train.py
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
from tests.helpers import VQ_TRAIN_GRAD_KEYS  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    ref_loader.load()
    from nets.inpainting.vqvae_1d_sc import VQVAE_SC
    B, T = 12, 88
    vq = VQVAE_SC(376, 512, 128, 1024, 2, 512, q_type="pro", groups=4)
    vq.load_state_dict(generate_state_dict(vq.state_dict(), 0, "VQ."))
    vq.train()
    x = make_input("vqtrain.x", (B, 376, T))
    z, qo, pred = vq(gt_poses=x, result_form="full")
    rec, parts = OT.vq_losses(pred.transpose(1, 2), x.transpose(1, 2))
    loss = rec + qo.loss.mean()
    loss.backward()
    sd = vq.state_dict()
    named = dict(vq.named_parameters())
    out = {"loss": loss.detach().numpy(), "e_q_loss": qo.loss.detach().numpy(), "pred": pred.detach().numpy()[:, :, :16],
           "z": z.detach().numpy()}
    for k, v in parts.items():
        out[k] = v.detach().numpy()
    for i in range(4):
        out[f"embeddings_{i}"] = sd[f"vq_layer.embeddings_{i}"].numpy()
        out[f"ema_cluster_size_{i}"] = sd[f"vq_layer.ema_cluster_size.{i}.hidden"].numpy()
        out[f"ema_dw_{i}"] = sd[f"vq_layer.ema_dw.{i}.hidden"].numpy()
    out["enc_project_running_mean"] = sd["encoder.project.norm.running_mean"].numpy()
    out["dec_up4_running_var"] = sd["decoder._up_4.norm.running_var"].numpy()
    for i, k in enumerate(VQ_TRAIN_GRAD_KEYS):
        g = named[k].grad
        out[f"grad_{i}"] = (g[:4] if g.numel() > 20000 else g).numpy()
    out["grad_norm"] = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(p.grad) for p in named.values()
                                                              if p.grad is not None])).numpy()
    np.savez_compressed(os.path.join(OUT, "vq_train_b12_t88.npz"), **out)
    print("loss", float(loss), "e_q", float(qo.loss), {k: float(v) for k, v in parts.items()})
    print("dead codes re-seeded per group:",
          [int((sd[f"vq_layer.ema_cluster_size.{i}.hidden"] == 1.0).sum()) for i in range(4)])
    print(os.path.getsize(os.path.join(OUT, "vq_train_b12_t88.npz")), "bytes")


if __name__ == "__main__":
    main()
