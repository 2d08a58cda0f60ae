"""CPU: the legacy VQ-VAE train-step oracle (oracle/probtalk_oracle_train.py: vq_ema_train, legacy_vq_losses,
legacy_vq_train_grads) against the fixture produced by the imported reference's VQVAE in train mode
(tools/make_golden_legacy_vq_train.py): losses, reconstruction, EMA-updated codebook / cluster sizes / code sums,
BatchNorm running statistics and a selection of gradients."""
import os

import numpy as np
import torch

from oracle import probtalk_oracle_train as OT
from probtalk_amd.weights import make_input
from tests.helpers import GOLDEN, LEGACY_VQ_TRAIN_GRAD_KEYS, ref_state_dict, rel_err


def legacy_vq_train_case():
    return make_input("lvqtrain.x", (6, 270, 88))


def load_gold():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "legacy_vq_train_b6_t88.npz")).items()}


def test_legacy_vq_train_oracle_matches_reference_fixture():
    g = load_gold()
    x = legacy_vq_train_case()
    loss, parts, grads, new, pred, idx = OT.legacy_vq_train_grads(ref_state_dict("g"), x)
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    for k in ("rec_loss", "velocity_loss", "e_q_loss"):
        assert abs(float(parts[k]) - float(g[k])) < 1e-5 * abs(float(g[k])), k
    assert rel_err(pred[:, :16], g["pred"]) < 1e-5
    assert rel_err(new["vq_layer.embeddings"], g["embeddings"]) < 1e-5
    assert rel_err(new["vq_layer.ema_cluster_size.hidden"], g["ema_cluster_size"]) < 1e-5
    assert rel_err(new["vq_layer.ema_dw.hidden"], g["ema_dw"]) < 1e-5
    assert rel_err(new["encoder.project.norm.running_mean"], g["enc_project_running_mean"]) < 1e-5
    assert rel_err(new["decoder._up_3.norm.running_var"], g["dec_up3_running_var"]) < 1e-5
    for i, k in enumerate(LEGACY_VQ_TRAIN_GRAD_KEYS):
        got = grads[k]
        got = got[:4] if got.numel() > 20000 else got
        assert rel_err(got, g[f"grad_{i}"]) < 2e-4, k
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(v) for v in grads.values()]))
    assert abs(float(total) - float(g["grad_norm"])) < 1e-4 * float(g["grad_norm"])
