"""Golden fixture of ProductQuantization's training branch on batches with FEWER latent chunks than codes, from the
IMPORTED REFERENCE (build container only): `_tile` (vqvae_modules.py:571-580) repeats the batch and adds
torch.randn_like(out) * 0.01 / sqrt(D) before dead codes are re-seeded from it (525-534; share_code: 509-512).

torch.randn_like is replaced for the duration of the forward by a function that returns prepared N(0,1) draws (the first
K rows of each call, zeros after: only out[:K] is read), as tools/make_golden.py replaces the sampler's draws.  Two
training steps each for the per-group books (B=8, T'=11: configs[0]'s shape, 88 rows < 128 codes) and for share_code
(64 chunks < 128 codes).  Data only.

    python tools/make_golden_pq_tile.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_loader  # noqa: E402
from tests.helpers import pq_tile_case  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "pq_tile_b8_w11.npz")
C, K, G = 512, 128, 4


class injected_randn_like:
    """torch.randn_like -> the next prepared [K, D] block on top of zeros of the asked shape."""

    def __init__(self, blocks):
        self.blocks = list(blocks)
        self.calls = 0

    def __enter__(self):
        self.orig = torch.randn_like

        def fake(t, **kw):
            blk = self.blocks[self.calls]
            self.calls += 1
            out = torch.zeros_like(t)
            assert t.shape[0] >= blk.shape[0] and t.shape[1] == blk.shape[1], (t.shape, blk.shape)
            out[:blk.shape[0]] = blk
            return out
        torch.randn_like = fake
        return self

    def __exit__(self, *exc):
        torch.randn_like = self.orig
        return False


def main():
    ref_loader.load()
    from nets.spg.vqvae_modules import ProductQuantization
    case = pq_tile_case()
    out = {}
    # ---- per-group code books (config/vq.json: share_code off)
    pq = ProductQuantization(C, K, 0.25, 0.99, epsilon=1e-5, num_chunks=G)
    with torch.no_grad():
        for i in range(G):
            getattr(pq, f"embeddings_{i}").copy_(case["cb"][i])
            pq.ema_dw[i].hidden.copy_(case["cb"][i])          # ExponentialMovingAverage(emb, decay) starts from the book
    pq.train()
    for step in range(2):
        x = case["xs"][step].clone().requires_grad_(True)
        with injected_randn_like([case["noise"][step][i] for i in range(G)]) as inj:
            qo = pq(x)
        assert inj.calls == G, inj.calls
        qo.quantized.square().sum().backward()
        sd = pq.state_dict()
        out[f"q_{step}"] = qo.quantized.detach().numpy()
        out[f"loss_{step}"] = qo.loss.detach().numpy()
        out[f"dx_{step}"] = x.grad.numpy()
        for i in range(G):
            out[f"embeddings_{i}_{step}"] = sd[f"embeddings_{i}"].numpy().copy()
            out[f"ema_cluster_size_{i}_{step}"] = sd[f"ema_cluster_size.{i}.hidden"].numpy().copy()
            out[f"ema_dw_{i}_{step}"] = sd[f"ema_dw.{i}.hidden"].numpy().copy()
        dead = sum(int((sd[f"ema_cluster_size.{i}.hidden"] == 1.0).sum()) for i in range(G))
        print("per-group step", step, "loss", float(qo.loss), "codes re-seeded", dead)
    # ---- share_code=True
    pq = ProductQuantization(C, K, 0.25, 0.99, epsilon=1e-5, num_chunks=G, share_code=True)
    pq._apply(lambda t: t.clone())                     # what a device move does: the buffer names stop aliasing
    with torch.no_grad():
        for i in range(G):
            getattr(pq, f"embeddings_{i}").copy_(case["cb_shared"])
        pq.ema_dw.hidden.copy_(case["cb_shared"])
    pq.train()
    for step in range(2):
        x = case["xs_shared"][step].clone().requires_grad_(True)
        with injected_randn_like([case["noise_shared"][step]]) as inj:
            qo = pq(x)
        assert inj.calls == 1, inj.calls
        qo.quantized.square().sum().backward()
        sd = pq.state_dict()
        out[f"s_q_{step}"] = qo.quantized.detach().numpy()
        out[f"s_loss_{step}"] = qo.loss.detach().numpy()
        out[f"s_embeddings_0_{step}"] = sd["embeddings_0"].numpy().copy()
        out[f"s_ema_cluster_size_{step}"] = sd["ema_cluster_size.hidden"].numpy().copy()
        out[f"s_ema_dw_{step}"] = sd["ema_dw.hidden"].numpy().copy()
        print("share_code step", step, "loss", float(qo.loss))
    np.savez_compressed(OUT, **out)
    print(os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
