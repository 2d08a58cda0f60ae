"""Golden fixture of the refiner TRAIN forward/backward from the IMPORTED REFERENCE (build container only).

Runs the reference's own RTransformer in train mode (refiner.py:296) on generator weights with every Dropout probability
set to 0 (masks come from torch's RNG).  refiner.py itself cannot be imported here (librosa / textgrid via data_utils), so
its get_loss (313-391) is applied through the oracle's restatement `refiner_losses`; the fixture pins the module's train
forward and the gradients autograd derives.  Data only.

    python tools/make_golden_refiner_train.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_loader  # noqa: E402
from oracle import probtalk_oracle_train as OT  # noqa: E402
from probtalk_amd.weights import generate_state_dict  # noqa: E402
from tests.helpers import REFINER_TRAIN_GRAD_KEYS, refiner_train_case  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    ref_loader.load()
    from nets.inpainting.RTransformer import RTransformer
    rt = RTransformer(376, 768, 512, 6, use_label=True, motion_context=True)
    rt.load_state_dict(generate_state_dict(rt.state_dict(), 0, "HFTransformer."))
    rt.train()
    for m in rt.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
        if isinstance(m, nn.MultiheadAttention):
            m.dropout = 0.0
    named = dict(rt.named_parameters())
    if "--search" in sys.argv:
        from tests.helpers import ref_state_dict, rel_err
        for tag in range(12):
            case = refiner_train_case(str(tag))
            rt.zero_grad()
            OT.refiner_losses(rt(case[1], case[0], case[3], case[4]), case[2], case[3])[0].backward()
            _, _, grads, _ = OT.refiner_train_grads(ref_state_dict("HFTransformer"), *case)
            worst = max(rel_err(grads[k], named[k].grad) for k in REFINER_TRAIN_GRAD_KEYS)
            print("tag", tag, "worst oracle-vs-reference gradient error", worst, flush=True)
        return
    aud, inp, gt, mask, label = refiner_train_case()
    pred = rt(inp, aud, mask, label)
    loss, parts = OT.refiner_losses(pred, gt, mask)
    loss.backward()
    out = {"loss": loss.detach().numpy(), "pred": pred.detach().numpy()}
    for k, v in parts.items():
        out[k] = v.detach().numpy()
    for i, k in enumerate(REFINER_TRAIN_GRAD_KEYS):
        g = named[k].grad
        out[f"grad_{i}"] = (g[:8] if g.numel() > 20000 else g).numpy()
    out["grad_norm"] = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(p.grad) for p in named.values()
                                                              if p.grad is not None])).numpy()
    np.savez_compressed(os.path.join(OUT, "refiner_train_b2_t64.npz"), **out)
    print("loss", float(loss), {k: float(v) for k, v in parts.items()}, "norm", float(out["grad_norm"]))
    print(os.path.getsize(os.path.join(OUT, "refiner_train_b2_t64.npz")), "bytes")


if __name__ == "__main__":
    main()
