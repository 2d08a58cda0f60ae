"""CPU: the refiner train oracle (oracle/probtalk_oracle_train.py: refiner_losses, refiner_train_grads) against the
fixture produced by the imported reference's RTransformer in train mode (tools/make_golden_refiner_train.py)."""
import os

import numpy as np
import torch

from oracle import probtalk_oracle_train as OT
from tests.helpers import GOLDEN, REFINER_TRAIN_GRAD_KEYS, ref_state_dict, refiner_train_case, rel_err


def load_gold():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "refiner_train_b2_t64.npz")).items()}


def test_refiner_train_oracle_matches_reference_fixture():
    g = load_gold()
    loss, parts, grads, pred = OT.refiner_train_grads(ref_state_dict("HFTransformer"), *refiner_train_case())
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * float(g["loss"])
    for k in ("face_loss", "body_loss", "hand_loss", "exp_loss", "rec_loss", "velocity_loss"):
        assert abs(float(parts[k]) - float(g[k])) < 1e-5 * float(g[k]), k
    assert rel_err(pred, g["pred"]) < 1e-5
    for i, k in enumerate(REFINER_TRAIN_GRAD_KEYS):
        got = grads[k]
        got = got[:8] if got.numel() > 20000 else got
        assert rel_err(got, g[f"grad_{i}"]) < 2e-4, k
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(v) for v in grads.values()]))
    assert abs(float(total) - float(g["grad_norm"])) < 1e-4 * float(g["grad_norm"])
