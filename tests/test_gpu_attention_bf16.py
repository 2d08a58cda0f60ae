"""GPU test of the bf16 attention core through the C ABI (pt_attention_bf16) against a float64 softmax(QK^T/8)V fed the
same bf16-rounded operands: the predictor's self / cross shapes, the refiner's 300-frame shape and ragged small ones."""
import pytest
import torch

from probtalk_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,Nq,Nk", [(3, 148, 148), (2, 148, 37), (2, 300, 300), (1, 50, 7), (2, 33, 160), (1, 1, 1),
                                      # beyond the LDS-resident key range: K / V streamed in 256-key chunks
                                      (2, 600, 600), (1, 700, 545), (1, 40, 1000), (1, 1800, 1800)])
def test_attention_bf16_matches_float64(B, Nq, Nk):
    L, H = _lib.lib(), 8
    g = torch.Generator().manual_seed(B * 1000 + Nq + Nk)
    q = torch.randn(B * Nq, 512, generator=g).bfloat16()
    kv = torch.randn(B * Nk, 1024, generator=g).bfloat16()
    qd, kvd = q.cuda(), kv.cuda()
    o = torch.full((B * Nq, 512 + 8), 3.0, device="cuda", dtype=torch.bfloat16)
    _lib.check(L.pt_attention_bf16(_lib.ptr(qd), 512, _lib.ptr(kvd), 1024, _lib.ptr(kvd[:, 512:]), 1024, _lib.ptr(o), 520, B, Nq, Nk, H,
                                   _lib.stream()))
    torch.cuda.synchronize()
    qf = q.double().view(B, Nq, H, 64).transpose(1, 2)
    kf = kv[:, :512].double().view(B, Nk, H, 64).transpose(1, 2)
    vf = kv[:, 512:].double().view(B, Nk, H, 64).transpose(1, 2)
    p = torch.softmax(qf @ kf.transpose(-1, -2) / 8.0, dim=-1)
    ref = (p @ vf).transpose(1, 2).reshape(B * Nq, 512)
    got = o.cpu().double()
    assert float((got[:, :512] - ref).abs().max()) / float(ref.abs().max()) < 1e-2      # bf16 probabilities and output
    assert bool((got[:, 512:] == 3.0).all())                                          # nothing outside the 512 columns
