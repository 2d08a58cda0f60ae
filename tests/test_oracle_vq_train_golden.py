"""CPU: the PQ-VAE train-step oracle (oracle/probtalk_oracle_train.py: pq_train, vq_losses, vq_train_grads) against the
fixture produced by the imported reference's VQVAE_SC in train mode (tools/make_golden_vq_train.py): losses, latent,
reconstruction, EMA-updated codebooks / cluster sizes / code sums (including re-seeded dead codes), BatchNorm running
statistics and a selection of gradients."""
import os

import numpy as np
import torch

from oracle import probtalk_oracle_train as OT
from probtalk_amd.weights import make_input
from tests.helpers import GOLDEN, VQ_TRAIN_GRAD_KEYS, ref_state_dict, rel_err


def vq_train_case():
    return make_input("vqtrain.x", (12, 376, 88))


def load_gold():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "vq_train_b12_t88.npz")).items()}


def test_vq_train_oracle_matches_reference_fixture():
    g = load_gold()
    x = vq_train_case()
    loss, parts, grads, new, pred, idx = OT.vq_train_grads(ref_state_dict("VQ"), x)
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    for k in ("face_loss", "body_loss", "hand_loss", "exp_loss", "velocity_loss", "e_q_loss"):
        assert abs(float(parts[k]) - float(g[k])) < 1e-5 * abs(float(g[k])), k
    assert rel_err(pred[:, :, :16], g["pred"]) < 1e-5
    reseeded = 0
    for i in range(4):
        assert rel_err(new[f"vq_layer.embeddings_{i}"], g[f"embeddings_{i}"]) < 1e-5
        assert rel_err(new[f"vq_layer.ema_cluster_size.{i}.hidden"], g[f"ema_cluster_size_{i}"]) < 1e-5
        assert rel_err(new[f"vq_layer.ema_dw.{i}.hidden"], g[f"ema_dw_{i}"]) < 1e-5
        reseeded += int((g[f"ema_cluster_size_{i}"] == 1.0).sum())
    assert reseeded > 0                     # the fixture exercises the dead-code branch
    assert rel_err(new["encoder.project.norm.running_mean"], g["enc_project_running_mean"]) < 1e-5
    assert rel_err(new["decoder._up_4.norm.running_var"], g["dec_up4_running_var"]) < 1e-5
    for i, k in enumerate(VQ_TRAIN_GRAD_KEYS):
        got = grads[k]
        got = got[:4] if got.numel() > 20000 else got
        assert rel_err(got, g[f"grad_{i}"]) < 2e-4, k
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(v) for v in grads.values()]))
    assert abs(float(total) - float(g["grad_norm"])) < 1e-4 * float(g["grad_norm"])
