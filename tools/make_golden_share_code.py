"""Golden fixture of ProductQuantization(share_code=True) from the IMPORTED REFERENCE (build container only).

The reference configures share_code off (config/vq.json:44); its code path exists (vqvae_modules.py:452-459,496-512) and
this fixture pins it: two training steps (EMA update of the one code book, dead codes re-seeded from the group-major
latent chunks, EMA state kept) and the eval forward afterwards.  The module is passed through `_apply(clone)` first, which
is what `.cuda()` / `.to(device)` do to it: nn.Module maps every buffer NAME on its own, so the G names stop aliasing one
tensor, training updates embeddings_0 only (line 512) and groups 1.. keep searching their initial copy.  Data only.

    python tools/make_golden_share_code.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_loader  # noqa: E402
from probtalk_amd.weights import make_input  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "pq_share_code_b12_w22.npz")
B, C, W, K, G = 12, 512, 22, 1024, 4


def inputs():
    cb = (make_input("pqshare.codebook", (K, C // G)) * 0.05).contiguous()
    xs = [(make_input(f"pqshare.x{i}", (B, C, W)) * 0.05).contiguous() for i in range(3)]
    return cb, xs


def main():
    ref_loader.load()
    from nets.spg.vqvae_modules import ProductQuantization
    pq = ProductQuantization(C, K, 0.25, 0.99, epsilon=1e-5, num_chunks=G, share_code=True)
    pq._apply(lambda t: t.clone())                     # what a device move does: the buffer names stop aliasing
    cb, xs = inputs()
    with torch.no_grad():
        for i in range(G):
            getattr(pq, f"embeddings_{i}").copy_(cb)
    out = {}
    pq.train()
    for step in range(2):
        x = xs[step].clone().requires_grad_(True)
        qo = pq(x)
        (qo.quantized * make_input(f"pqshare.w{step}", (B, C, W))).sum().backward()
        sd = pq.state_dict()
        out[f"q_{step}"] = qo.quantized.detach().numpy()[:3]                   # slices keep the fixture around 1 MB
        out[f"loss_{step}"] = qo.loss.detach().numpy()
        out[f"dx_{step}"] = x.grad.numpy()[:3]
        out[f"embeddings_0_{step}"] = sd["embeddings_0"].numpy()[::3].copy()
        out[f"ema_cluster_size_{step}"] = sd["ema_cluster_size.hidden"].numpy().copy()
        out[f"ema_dw_{step}"] = sd["ema_dw.hidden"].numpy()[::3].copy()
        print("step", step, "loss", float(qo.loss), "codes re-seeded",
              int((sd["embeddings_0"][:, 0] == xs[step].permute(0, 2, 1).reshape(-1, G, C // G).permute(1, 0, 2)
                   .reshape(-1, C // G)[:K, 0]).sum()))
    assert torch.equal(pq.state_dict()["embeddings_1"], cb), "groups 1.. keep the initial code book"
    pq.eval()
    with torch.no_grad():
        qo = pq(xs[2])
    out["q_eval"] = qo.quantized.numpy()[:3]
    out["idx_eval"] = qo.loss.numpy().astype(np.int32)
    np.savez_compressed(OUT, **out)
    print(os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
