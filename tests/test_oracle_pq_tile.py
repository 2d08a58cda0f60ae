"""CPU: the oracle's restatement of ProductQuantization._tile (oracle/probtalk_oracle_train.py::tile, vqvae_modules.py:571-580;
callers 509-512 and 525-534) against the fixture made by the imported reference with torch.randn_like replaced by prepared
draws (tools/make_golden_pq_tile.py): two training steps on batches with fewer latent chunks than codes, per-group code
books (configs[0]'s shape: B=8, T=88 -> 88 rows < 128 codes) and share_code."""
import os

import numpy as np
import torch

from oracle import probtalk_oracle_train as OT
from tests.helpers import GOLDEN, pq_tile_case, rel_err

G, K = 4, 128


def _gold():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "pq_tile_b8_w11.npz")).items()}


def test_tile_per_group_oracle_matches_reference_fixture():
    g, case = _gold(), pq_tile_case()
    sd = {}
    for i in range(G):
        sd[f"embeddings_{i}"] = case["cb"][i].clone()
        sd[f"ema_dw.{i}.hidden"] = case["cb"][i].clone()
        sd[f"ema_cluster_size.{i}.hidden"] = torch.zeros(K)
    for step in range(2):
        x = case["xs"][step].clone().requires_grad_(True)
        q, loss, idx, new = OT.pq_train(sd, "", x, G, counter=step + 1, tile_noise=case["noise"][step])
        q.square().sum().backward()
        assert torch.equal(q.detach(), g[f"q_{step}"])
        assert abs(float(loss) - float(g[f"loss_{step}"])) < 1e-6 * abs(float(g[f"loss_{step}"]))
        assert rel_err(x.grad, g[f"dx_{step}"]) < 1e-6
        for i in range(G):
            # re-seeded codes ARE rows of the tiled batch (chunk k mod 88 plus noise * 0.01 / sqrt(128)): bit for bit
            flat = case["xs"][step].permute(0, 2, 1).reshape(-1, G, 128)
            code_rand = OT.tile(flat[:, i], K, case["noise"][step][i])[:K]
            dead = (g[f"embeddings_{i}_{step}"] == code_rand).all(dim=1)
            assert int(dead.sum()) > 0
            assert torch.equal(new[f"embeddings_{i}"][dead], code_rand[dead])
            assert rel_err(new[f"embeddings_{i}"], g[f"embeddings_{i}_{step}"]) < 1e-5
            assert rel_err(new[f"ema_cluster_size.{i}.hidden"], g[f"ema_cluster_size_{i}_{step}"]) < 1e-6
            assert rel_err(new[f"ema_dw.{i}.hidden"], g[f"ema_dw_{i}_{step}"]) < 1e-6
        sd.update(new)


def test_tile_share_code_oracle_matches_reference_fixture():
    g, case = _gold(), pq_tile_case()
    sd = {f"embeddings_{i}": case["cb_shared"].clone() for i in range(G)}
    sd["ema_dw.hidden"] = case["cb_shared"].clone()
    sd["ema_cluster_size.hidden"] = torch.zeros(K)
    for step in range(2):
        x = case["xs_shared"][step].clone().requires_grad_(True)
        q, loss, idx, new = OT.pq_train(sd, "", x, G, counter=step + 1, share_code=True, tile_noise=case["noise_shared"][step])
        assert torch.equal(q.detach(), g[f"s_q_{step}"])
        assert abs(float(loss) - float(g[f"s_loss_{step}"])) < 1e-6 * abs(float(g[f"s_loss_{step}"]))
        assert rel_err(new["embeddings_0"], g[f"s_embeddings_0_{step}"]) < 1e-5
        assert rel_err(new["ema_cluster_size.hidden"], g[f"s_ema_cluster_size_{step}"]) < 1e-6
        assert rel_err(new["ema_dw.hidden"], g[f"s_ema_dw_{step}"]) < 1e-6
        sd.update(new)
