"""CPU: the train-step oracle (oracle/probtalk_oracle_train.py) against the fixture produced by the imported reference
(tools/make_golden_train.py): loss, logits, per-module gradient norms, a selection of gradients, BatchNorm running mean."""
import os

import numpy as np
import torch

from oracle import probtalk_oracle_train as OT
from probtalk_amd.weights import make_input
from tests.helpers import GOLDEN, TRAIN_GRAD_KEYS, ref_state_dict, rel_err


def train_case():
    B, T, Tq, G = 2, 88, 11, 4
    aud = make_input("train.aud", (B, 768, T))
    gt = make_input("train.gt", (B, 376, T))
    mask = torch.zeros(B, 1, T)
    mask[1, :, 20:50] = 1
    label = torch.tensor([2, 0])
    state = make_input("train.state", (B, Tq, G), "int", low=0, high=128)
    keep = make_input("train.keep", (B, Tq * G), "uniform")
    return aud, gt * mask, mask, label, state, keep, 0.5


def grad_keys():
    return TRAIN_GRAD_KEYS


def test_train_oracle_matches_reference_fixture():
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "predictor_train_b2_t88.npz")).items()}
    loss, grads, running, logits = OT.predictor_train_grads(ref_state_dict("AudEnc"), ref_state_dict("Predictor"), *train_case())
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    assert rel_err(logits, g["logits"]) < 1e-5
    na, _ = OT.clip_coef([v for (t, _), v in grads.items() if t == "AudEnc"], 1e9)
    npd, _ = OT.clip_coef([v for (t, _), v in grads.items() if t == "Predictor"], 1e9)
    assert abs(float(na) - float(g["grad_norm_audenc"])) < 1e-4 * float(g["grad_norm_audenc"])
    assert abs(float(npd) - float(g["grad_norm_predictor"])) < 1e-4 * float(g["grad_norm_predictor"])
    for i, key in enumerate(grad_keys()):
        ref = g[f"grad_{i}"]
        got = grads[key]
        got = got[:8] if got.numel() > 20000 else got
        assert rel_err(got, ref) < 2e-4, key
    assert rel_err(running["feature_map.1.running_mean"], g["audenc_feature_map_1_running_mean"]) < 1e-5
