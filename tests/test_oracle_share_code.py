"""CPU: the oracle's share_code=True branch of ProductQuantization (oracle/probtalk_oracle_train.py::pq_train,
vqvae_modules.py:452-459,496-512) against the fixture made by the imported reference (tools/make_golden_share_code.py):
two training steps on a module whose buffer names no longer alias (as after a device move), then the eval forward."""
import os

import numpy as np
import torch

from oracle import probtalk_oracle as O
from oracle import probtalk_oracle_train as OT
from tests.helpers import GOLDEN, rel_err, share_code_case


def test_share_code_oracle_matches_reference_fixture():
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "pq_share_code_b12_w22.npz")).items()}
    cb, xs, ws = share_code_case()
    K, D = cb.shape
    sd = {f"embeddings_{i}": cb.clone() for i in range(4)}
    sd["ema_cluster_size.hidden"] = torch.zeros(K)
    sd["ema_dw.hidden"] = torch.zeros(K, D)
    for step in range(2):
        x = xs[step].clone().requires_grad_(True)
        q, loss, idx, new = OT.pq_train(sd, "", x, 4, counter=step + 1, share_code=True)
        (q * ws[step]).sum().backward()
        assert torch.equal(q.detach()[:3], g[f"q_{step}"])
        assert abs(float(loss) - float(g[f"loss_{step}"])) < 1e-6 * abs(float(g[f"loss_{step}"]))
        assert rel_err(x.grad[:3], g[f"dx_{step}"]) < 1e-6
        assert set(new) == {"embeddings_0", "ema_cluster_size.hidden", "ema_dw.hidden"}
        assert rel_err(new["embeddings_0"][::3], g[f"embeddings_0_{step}"]) < 1e-5
        assert rel_err(new["ema_cluster_size.hidden"], g[f"ema_cluster_size_{step}"]) < 1e-6
        assert rel_err(new["ema_dw.hidden"][::3], g[f"ema_dw_{step}"]) < 1e-6
        sd.update(new)
    assert not torch.equal(sd["embeddings_0"], sd["embeddings_1"])       # group 0 trains, groups 1.. keep the initial book
    q, idx = O.pq_quantize(sd, "", xs[2], 4)
    assert torch.equal(idx.to(torch.int32), g["idx_eval"])
    assert torch.equal(q[:3], g["q_eval"])
