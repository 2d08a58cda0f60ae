"""GPU: ProductQuantization's training branch on batches with FEWER latent chunks than codes (`_tile`,
vqvae_modules.py:571-580) through the module API and the C ABI (pt_train_pq_ema / pt_train_pq_ema_shared), against the
fixture of the imported reference (tools/make_golden_pq_tile.py; `_tile`'s N(0,1) draws injected) and, with the kernel's
own generator, on statistics.  configs[0]'s own shape (B=8, T=88 -> 88 rows < 128 codes) takes a training step."""
import os

import numpy as np
import pytest
import torch

from oracle import probtalk_oracle_train as OT
from tests.helpers import GOLDEN, pq_tile_case, rel_err

pytestmark = pytest.mark.gpu
G, K, C = 4, 128, 512


def _gold():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "pq_tile_b8_w11.npz")).items()}


def test_tile_per_group_train_steps_match_reference_fixture():
    from probtalk_amd.nets.modules import ProductQuantization
    g, case = _gold(), pq_tile_case()
    pq = ProductQuantization(C, K, 0.25, 0.99, epsilon=1e-5, num_chunks=G)
    with torch.no_grad():
        for i in range(G):
            getattr(pq, f"embeddings_{i}").copy_(case["cb"][i])
            pq.ema_dw[i].hidden.copy_(case["cb"][i])
    pq.cuda().train()
    for step in range(2):
        x = case["xs"][step].cuda().requires_grad_(True)
        pq.__dict__["_pt_tile_noise"] = case["noise"][step]
        qo = pq(x)
        qo.quantized.square().sum().backward()
        # the forward value of the straight-through output is x + (q - x): equal to the code up to one rounding of x's size
        assert rel_err(qo.quantized.detach().cpu(), g[f"q_{step}"]) < 1e-6
        assert abs(float(qo.loss) - float(g[f"loss_{step}"])) < 1e-5 * abs(float(g[f"loss_{step}"]))
        assert rel_err(x.grad.cpu(), g[f"dx_{step}"]) < 1e-5
        flat = case["xs"][step].permute(0, 2, 1).reshape(-1, G, C // G)
        for i in range(G):
            # the re-seeded codes of this step: the rows of the reference's new code book that ARE the tiled batch (chunk
            # k mod 88 plus noise * 0.01 / sqrt(128): a rounded product and a rounded sum).  Bit for bit on the GPU too.
            code_rand = OT.tile(flat[:, i], K, case["noise"][step][i])[:K]
            dead = (g[f"embeddings_{i}_{step}"] == code_rand).all(dim=1)
            emb = getattr(pq, f"embeddings_{i}").cpu()
            assert int(dead.sum()) > 0
            assert torch.equal(emb[dead], code_rand[dead]), f"group {i}: re-seeded codes differ"
            assert rel_err(emb, g[f"embeddings_{i}_{step}"]) < 1e-5
            assert rel_err(pq.ema_cluster_size[i].hidden.cpu(), g[f"ema_cluster_size_{i}_{step}"]) < 1e-5
            assert rel_err(pq.ema_dw[i].hidden.cpu(), g[f"ema_dw_{i}_{step}"]) < 1e-5


def test_tile_share_code_train_steps_match_reference_fixture():
    from probtalk_amd.nets.modules import ProductQuantization
    g, case = _gold(), pq_tile_case()
    pq = ProductQuantization(C, K, 0.25, 0.99, epsilon=1e-5, num_chunks=G, share_code=True)
    pq.cuda()                                           # the buffer names stop aliasing, as in the fixture
    with torch.no_grad():
        for i in range(G):
            getattr(pq, f"embeddings_{i}").copy_(case["cb_shared"])
        pq.ema_dw.hidden.copy_(case["cb_shared"])
    pq.train()
    for step in range(2):
        x = case["xs_shared"][step].cuda().requires_grad_(True)
        pq.__dict__["_pt_tile_noise"] = case["noise_shared"][step]
        qo = pq(x)
        assert rel_err(qo.quantized.detach().cpu(), g[f"s_q_{step}"]) < 1e-6
        assert abs(float(qo.loss) - float(g[f"s_loss_{step}"])) < 1e-5 * abs(float(g[f"s_loss_{step}"]))
        assert rel_err(pq.embeddings_0.cpu(), g[f"s_embeddings_0_{step}"]) < 1e-5
        assert rel_err(pq.ema_cluster_size.hidden.cpu(), g[f"s_ema_cluster_size_{step}"]) < 1e-5
        assert rel_err(pq.ema_dw.hidden.cpu(), g[f"s_ema_dw_{step}"]) < 1e-5


def test_tile_kernel_generator_draws_unit_normals():
    """Without injected draws the kernel generates `_tile`'s noise itself: the re-seeded codes minus their seed chunks,
    divided by 0.01 / sqrt(D), must look like N(0,1) (mean, variance, no repeats between codes or calls)."""
    from probtalk_amd.nets.modules import ProductQuantization
    case = pq_tile_case()
    draws = []
    for rep in range(2):
        pq = ProductQuantization(C, K, 0.25, 0.99, epsilon=1e-5, num_chunks=G)
        with torch.no_grad():
            for i in range(G):
                getattr(pq, f"embeddings_{i}").copy_(case["cb"][i])
                pq.ema_dw[i].hidden.copy_(case["cb"][i])
        pq.cuda().train()
        torch.manual_seed(100 + rep)                    # the seed of the kernel's generator comes from torch's CPU generator
        x = case["xs"][0].cuda()
        pq(x)
        flat = x.permute(0, 2, 1).reshape(-1, G, C // G)               # (88, G, D)
        for i in range(G):
            dead = (pq.ema_cluster_size[i].hidden == 1.0).nonzero().flatten()
            seed_rows = flat[dead % flat.shape[0], i]
            draws.append(((getattr(pq, f"embeddings_{i}")[dead] - seed_rows) / (0.01 / (C // G) ** 0.5)).flatten().cpu())
    a, b = torch.cat(draws[:G]), torch.cat(draws[G:])
    assert a.numel() > 20000
    assert abs(float(a.mean())) < 0.03 and abs(float(a.var()) - 1.0) < 0.05, (float(a.mean()), float(a.var()))
    assert float(a.abs().max()) < 6.5
    assert float((a[:b.numel()] - b[:a.numel()]).abs().mean()) > 0.5, "two calls drew the same noise"


def test_vq_teacher_trains_at_configs0_shape():
    """BASELINE configs[0]: B=8 clips of 88 frames -> 88 latent rows < 128 codes.  The s2g_body_vqt train step runs (it
    raised before `_tile` was built), its losses are finite and go down over a few steps."""
    import math
    from probtalk_amd import nets, synthetic
    from probtalk_amd.config import default_transformer_config
    from probtalk_amd.weights import fill_module, make_input
    w = nets.init_model("s2g_body_vqt", synthetic.args(infer=False), default_transformer_config(vq_path=None))
    fill_module(w.VQ, 0, "VQ.")
    B, T = 8, 88
    bat = {"aud_feat": make_input("c0.aud", (B, 64, T)), "poses": 0.5 * make_input("c0.poses", (B, 330, T)),
           "expression": 0.5 * make_input("c0.expr", (B, 100, T)), "speaker": torch.zeros(B, dtype=torch.long), "epoch": 0}
    losses = []
    for _ in range(8):
        _, ld = w(bat)
        total = 0.1 * (ld["face_loss"] + ld["exp_loss"]) + 0.4 * (ld["body_loss"] + ld["hand_loss"]) + ld["velocity_loss"] + ld["e_q_loss"]
        assert math.isfinite(total)
        losses.append(total)
    assert min(losses[-2:]) < losses[0], losses
    assert w.VQ.vq_layer.ema_dw[0].counter == 8
