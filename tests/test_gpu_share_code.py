"""GPU: ProductQuantization(share_code=True) (vqvae_modules.py:452-459,496-512; configured off in config/vq.json:44, its
code path exists) through the module API against the fixture of the imported reference (tools/make_golden_share_code.py):
two training steps (one code book trained by all groups' chunks, dead codes re-seeded from the group-major chunks, EMA
state kept, straight-through gradient) and the eval forward on the diverged code books."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, rel_err, share_code_case

pytestmark = pytest.mark.gpu


def test_share_code_train_steps_and_eval_match_reference_fixture():
    from probtalk_amd.nets.modules import ProductQuantization
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "pq_share_code_b12_w22.npz")).items()}
    cb, xs, ws = share_code_case()
    G = 4
    pq = ProductQuantization(512, 1024, 0.25, 0.99, epsilon=1e-5, num_chunks=G, share_code=True)
    assert sorted(pq.state_dict()) == sorted([f"embeddings_{i}" for i in range(G)] + ["ema_cluster_size.hidden", "ema_dw.hidden"])
    pq.load_state_dict({**{f"embeddings_{i}": cb for i in range(G)}, "ema_cluster_size.hidden": torch.zeros(1024),
                        "ema_dw.hidden": torch.zeros(1024, 128)})
    pq.cuda().train()
    for step in range(2):
        x = xs[step].cuda().requires_grad_(True)
        qo = pq(x)
        (qo.quantized * ws[step].cuda()).sum().backward()
        # the forward value of the straight-through output is x + (q - x): equal to the code up to one rounding of x's size
        assert rel_err(qo.quantized.detach().cpu()[:3], g[f"q_{step}"]) < 1e-6
        assert abs(float(qo.loss) - float(g[f"loss_{step}"])) < 1e-5 * abs(float(g[f"loss_{step}"]))
        assert rel_err(x.grad.cpu()[:3], g[f"dx_{step}"]) < 1e-5
        assert rel_err(pq.embeddings_0.cpu()[::3], g[f"embeddings_0_{step}"]) < 1e-5
        assert rel_err(pq.ema_cluster_size.hidden.cpu(), g[f"ema_cluster_size_{step}"]) < 1e-5
        assert rel_err(pq.ema_dw.hidden.cpu()[::3], g[f"ema_dw_{step}"]) < 1e-5
        assert pq.ema_dw.counter == step + 1
        for i in range(1, G):                                       # only embeddings_0 trains (line 512)
            assert torch.equal(getattr(pq, f"embeddings_{i}").cpu(), cb)
    pq.eval()
    qo = pq(xs[2].cuda())
    assert qo.loss.dtype == torch.int64
    assert torch.equal(qo.loss.cpu().to(torch.int32), g["idx_eval"])
    # group 0's codes come from the trained book (equal to the reference's to rounding), groups 1.. from the initial one
    assert rel_err(qo.quantized.cpu()[:3], g["q_eval"]) < 1e-5
    assert torch.equal(qo.quantized.cpu()[:3, 128:], g["q_eval"][:, 128:])


def test_share_code_short_batch_is_tiled_not_rejected():
    """rows * G < K: the reference's noisy re-tiling (vqvae_modules.py:573-577) applies; parity is pinned in
    tests/test_gpu_pq_tile.py, here only that the step runs and re-seeds dead codes from the tiled batch."""
    from probtalk_amd.nets.modules import ProductQuantization
    pq = ProductQuantization(512, 1024, 0.25, 0.99, num_chunks=4, share_code=True).cuda().train()
    before = pq.embeddings_0.clone()
    pq(torch.randn(2, 512, 22, device="cuda") * 0.05)
    assert bool(torch.isfinite(pq.embeddings_0).all()) and not torch.equal(pq.embeddings_0, before)
