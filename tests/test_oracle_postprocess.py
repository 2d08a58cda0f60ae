"""CPU: the post-process oracle (oracle/postprocess.py) against the fixture made from the imported reference functions
(tools/make_golden_postprocess.py): 6-D -> rotation matrix -> axis-angle incl. the small-angle branch and angles up to
pi, part2full's lower-body insertion, c_index_6d."""
import math
import os

import numpy as np
import torch

from oracle import postprocess as OP
from tests.helpers import GOLDEN


def load_gold():
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(GOLDEN, "postprocess.npz")).items()}


def test_postprocess_oracle_matches_reference_fixture():
    g = load_gold()
    out = OP.postprocess(g["denormalized"], None)
    assert tuple(out.shape) == tuple(g["out"].shape) == (3, 32, 265)
    assert torch.equal(out, g["out"])                     # same torch operations in the same order: bit for bit
    # the planted edge cases did reach the special branches
    n_e = int(g["n_edge"])
    aa = out[2][:, list(range(3)) + list(range(18, 21)) + list(range(27, 30)) + list(range(36, 165))].reshape(-1, 3)[:n_e]
    ang = aa.norm(dim=-1)
    assert int((ang < 1e-6).sum()) >= 8 and int((ang > math.pi - 2e-3).sum()) >= 8
    # fixed lower-body joints
    assert torch.equal(out[0, 5, 3:18], OP.LOWER_POSE[:15]) and torch.equal(out[1, 0, 30:36], OP.LOWER_POSE[21:27])


def test_denormalize_gathers_the_fbhe_statistics():
    g = load_gold()
    ns = g["norm_stats"]
    assert tuple(ns.shape) == (2, 1, 1, 430)
    cidx = OP.c_index()
    assert cidx.shape == (270,) and cidx[0] == 36 and cidx[-1] == 329
    x = torch.randn(2, 376, 5, generator=torch.Generator().manual_seed(0))
    y = OP.denormalize(x, ns[0], ns[1], cidx)
    mean, std = ns[0].reshape(-1), ns[1].reshape(-1)
    assert torch.equal(y[:, 3], x[:, 3] * std[3] + mean[3])                       # jaw: statistics 0..5
    assert torch.equal(y[:, 6], x[:, 6] * std[36] + mean[36])                     # first kept body scalar: 6-D index 36
    assert torch.equal(y[:, 276], x[:, 276] * std[330] + mean[330])               # expression: the last 100
    out = OP.postprocess(x, ns, full=False)
    assert tuple(out.shape) == (2, 5, 238)
