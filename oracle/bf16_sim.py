"""TEST INFRASTRUCTURE (not product code): the fp32 oracle with the roundings of the bf16 mode applied on the CPU.

`with bf16_operands():` makes every oracle function round the operands of its Linears / convs (weights and
activations), of the attention products (q, k, v, probabilities) and the residual stream after each post-LN to bf16,
accumulating in fp32 as the MFMA does.  Nothing else changes.  What it is for: the distance between this simulation and
the fp32 oracle is the error that bf16 OPERANDS cost by themselves, whatever the kernels do; tests/test_gpu_bf16.py
bounds the HIP path's worst-element error by a small multiple of it, and tools/bf16_error_budget.py prints the split
per rounding site (weights alone: 6.1-6.5e-3 max-rel, activations alone: 6.5-7.6e-3, both: 0.9-1.2e-2 on the fixtures).
"""
import contextlib

import torch
import torch.nn.functional as F

from . import probtalk_oracle as O

_CFG = dict(w=False, a=False, attn=False, resid=False)


def _r(t, on):
    return t.to(torch.bfloat16).to(torch.float32) if on and t.dtype == torch.float32 else t


class _Shim:
    def __init__(self, real):
        self._real = real

    def __getattr__(self, k):
        return getattr(self._real, k)


class _FShim(_Shim):
    def linear(self, x, w, b=None):
        return F.linear(_r(x, _CFG["a"]), _r(w, _CFG["w"]), b)

    def conv1d(self, x, w, b=None, *a, **k):
        return F.conv1d(_r(x, _CFG["a"]), _r(w, _CFG["w"]), b, *a, **k)

    def conv_transpose1d(self, x, w, b=None, *a, **k):
        return F.conv_transpose1d(_r(x, _CFG["a"]), _r(w, _CFG["w"]), b, *a, **k)


class _TShim(_Shim):
    def matmul(self, a, b):
        return torch.matmul(_r(a, _CFG["attn"]), _r(b, _CFG["attn"]))


@contextlib.contextmanager
def bf16_operands(weights=True, activations=True, attention=True, residual=True):
    f0, t0, dl0 = O.F, O.torch, O.decoder_layer
    old = dict(_CFG)
    _CFG.update(w=weights, a=activations, attn=attention, resid=residual)

    def dl(sd, p, x, mem, le):
        return _r(dl0(sd, p, _r(x, _CFG["resid"]), mem, le), _CFG["resid"])

    O.F, O.torch, O.decoder_layer = _FShim(F), _TShim(torch), dl
    try:
        yield
    finally:
        O.F, O.torch, O.decoder_layer = f0, t0, dl0
        _CFG.update(old)
