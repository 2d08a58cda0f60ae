"""CPU error budget of the bf16 mode: the fp32 oracle with its matmul operands rounded to bf16 at chosen sites.

Answers which rounding sets the distance between the bf16 mode and the fp32 reference (tests/test_gpu_bf16.py): it runs
the oracle's predictor logits (B=2, T'=11) and refiner (B=2, T=64) on the golden inputs with
  * weights of every Linear / 1x1 conv rounded to bf16,
  * activations entering them rounded to bf16,
  * the attention operands (q, k, v, and the softmax probabilities) rounded to bf16,
  * the residual stream rounded to bf16 after every post-LN,
in the combinations listed below, and prints ||d|| / ||ref|| (l2-rel) and max|d| / max|ref| (max-rel).
Accumulation stays fp32 (as on the MFMA).  Run:  python tools/bf16_error_budget.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import probtalk_oracle as O          # noqa: E402
from probtalk_amd.weights import make_input      # noqa: E402
from tests.helpers import golden, ref_state_dict  # noqa: E402

from oracle.bf16_sim import bf16_operands     # noqa: E402


def run(fn, **sites):
    with bf16_operands(**{k: sites.get(k, False) for k in ("weights", "activations", "attention", "residual")}):
        return fn()


def errs(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm()), float((a - b).abs().max() / b.abs().max())


def main():
    sd_p, sd_r = ref_state_dict("Predictor"), ref_state_dict("HFTransformer")
    g = golden("predictor_b2_t88.npz")
    ids = make_input("pred.ids", (2, 11, 4), "int", low=0, high=129)
    label = torch.tensor([1, 3])
    x = make_input("ref.x", (2, 376, 64))
    aud = make_input("ref.aud", (2, 768, 64))
    mask = torch.zeros(2, 1, 64)
    mask[:, :, :30] = 1
    x3, a3 = make_input("ref.x300", (2, 376, 300)), make_input("ref.a300", (2, 768, 300))
    m3 = torch.zeros(2, 1, 300)
    m3[:, :, 100:130] = 1
    cases = {
        "predictor logits (B=2,T'=11)": lambda: O.tokens_to_logits(sd_p, ids, label, g["condition"]),
        "refiner (B=2,T=64)": lambda: O.rtransformer(sd_r, x, aud, mask, torch.tensor([0, 2])),
        "refiner (B=2,T=300)": lambda: O.rtransformer(sd_r, x3, a3, m3, torch.tensor([3])),
    }
    combos = [("weights only", dict(weights=True)), ("activations only", dict(activations=True)), ("attention operands only", dict(attention=True)),
              ("residual stream only", dict(residual=True)), ("weights + activations", dict(weights=True, activations=True)),
              ("w + a + attention", dict(weights=True, activations=True, attention=True)),
              ("all four (the bf16 mode's roundings)", dict(weights=True, activations=True, attention=True, residual=True))]
    with torch.no_grad():
        for name, fn in cases.items():
            ref = fn()
            print(name)
            for cname, c in combos:
                out = run(fn, **c)
                e2, em = errs(out, ref)
                print(f"   {cname:40s} l2-rel {e2:.3e}   max-rel {em:.3e}")

if __name__ == "__main__":
    main()
