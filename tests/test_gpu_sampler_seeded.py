"""GPU tests of the seeded MaskGIT sampler: both random draws generated inside the step kernel.
* the noise it uses (pt_maskgit_noise) has the right distributions: Exp(1) for the categorical draw, Gumbel(0,1) for
  the confidence, no repeats between passes or seeds;
* feeding exactly that noise to the injected-noise entry point (the parity mode, itself checked against the oracle in
  tests/test_gpu_transformer.py) reproduces the seeded codes bit for bit;
* same seed -> same codes, other seed -> other codes; the default call draws its seed from torch's CPU generator."""
import math

import pytest
import torch

from probtalk_amd import _lib

pytestmark = pytest.mark.gpu


def _noise(seed, step, B, N, V):
    q = torch.empty(B, N, V, device="cuda")
    g = torch.empty(B, N, device="cuda")
    _lib.check(_lib.lib().pt_maskgit_noise(seed, step, B, N, V, _lib.ptr(q), _lib.ptr(g), _lib.stream()))
    torch.cuda.synchronize()
    return q, g


def test_noise_distributions():
    B, N, V = 64, 148, 128
    q, g = _noise(1234, 0, B, N, V)
    qd = q.double().flatten()
    assert bool(torch.isfinite(q).all()) and float(q.min()) > 0
    n = qd.numel()
    assert abs(float(qd.mean()) - 1.0) < 5 / math.sqrt(n)                 # Exp(1): mean 1, variance 1
    assert abs(float(qd.var()) - 1.0) < 0.01
    for x in (0.1, 0.5, 1.0, 2.0, 4.0):                                  # CDF 1 - exp(-x)
        assert abs(float((qd < x).double().mean()) - (1 - math.exp(-x))) < 4 / math.sqrt(n)
    # Gumbel(0,1): mean = Euler-Mascheroni, variance pi^2/6, CDF exp(-exp(-x)); more samples from several passes
    gs = torch.cat([_noise(1234, s, B, N, V)[1].flatten() for s in range(8)]).double()
    m = gs.numel()
    assert abs(float(gs.mean()) - 0.5772156649) < 5 * math.sqrt(math.pi ** 2 / 6 / m)
    assert abs(float(gs.var()) - math.pi ** 2 / 6) < 0.05
    for x in (-1.0, 0.0, 1.0, 3.0):
        assert abs(float((gs < x).double().mean()) - math.exp(-math.exp(-x))) < 4 / math.sqrt(m)
    # streams do not repeat: another pass, another seed, and the two draws of one pass are all different sequences
    q1, g1 = _noise(1234, 1, B, N, V)
    q2, _ = _noise(1235, 0, B, N, V)
    assert float((q == q1).float().mean()) < 1e-4 and float((q == q2).float().mean()) < 1e-4
    assert abs(float(torch.corrcoef(torch.stack([q.flatten(), q1.flatten()]))[0, 1])) < 0.01
    assert abs(float(torch.corrcoef(torch.stack([q[:, :, 0].flatten(), g.flatten()]))[0, 1])) < 0.02


def _predictor():
    from probtalk_amd.synthetic import build_pipeline
    pred, _ = build_pipeline(seed=0)
    pred._eval()
    return pred.Predictor.decoder


def test_seeded_sampler_equals_injected_sampler_on_its_own_noise():
    from probtalk_amd.weights import make_input
    dec = _predictor()
    B, Tq, T = 3, 11, 8
    N, V = Tq * 4, 128
    cond = make_input("seeded.cond", (B, 768, Tq)).cuda()
    label = torch.tensor([0, 3, 1]).cuda()
    seed = 0x1234567890ABCDEF
    codes = dec.sample_good(label, cond, None, T=T, seed=seed)
    again = dec.sample_good(label, cond, None, T=T, seed=seed)
    other = dec.sample_good(label, cond, None, T=T, seed=seed + 1)
    assert torch.equal(codes, again)
    assert not torch.equal(codes, other)
    assert int(codes.min()) >= 0 and int(codes.max()) < V                  # every token decided, no mask id left
    qs, gs = zip(*[_noise(seed, s, B, N, V) for s in range(T)])
    injected = dec.sample_good(label, cond, None, T=T, noise=(torch.stack(qs), torch.stack(gs)))
    assert torch.equal(codes, injected)


def test_default_call_follows_torch_manual_seed():
    from probtalk_amd.weights import make_input
    dec = _predictor()
    cond = make_input("seeded.cond2", (2, 768, 11)).cuda()
    label = torch.tensor([2, 2]).cuda()
    torch.manual_seed(7)
    a = dec.sample_good(label, cond, None, T=8)
    b = dec.sample_good(label, cond, None, T=8)
    torch.manual_seed(7)
    c = dec.sample_good(label, cond, None, T=8)
    assert torch.equal(a, c) and not torch.equal(a, b)


def test_uniform_extremes_stay_inside_the_open_interval():
    """The generator's bits -> uniform map at the ends of its 23-bit range (and beyond it: upper bits are masked off):
    u never reaches 0 or 1, so Exp(1) > 0 and Gumbel are finite for every possible draw."""
    pats = [0, 1, 2, (1 << 22) - 1, 1 << 22, (1 << 23) - 2, (1 << 23) - 1, (1 << 24) - 1, 0xFFFFFFFF, 1 << 23]
    bits = torch.tensor(pats, dtype=torch.int64).to(torch.uint32).cuda() if hasattr(torch, "uint32") else None
    if bits is None:
        bits = torch.tensor([p - (1 << 32) if p >= (1 << 31) else p for p in pats], dtype=torch.int32).cuda()
    n = len(pats)
    u, q, g = (torch.empty(n, device="cuda") for _ in range(3))
    _lib.check(_lib.lib().pt_maskgit_noise_from_bits(_lib.ptr(bits), n, _lib.ptr(u), _lib.ptr(q), _lib.ptr(g), _lib.stream()))
    torch.cuda.synchronize()
    ud = u.double().cpu()
    expect = torch.tensor([((p & 0x7FFFFF) + 0.5) / 2.0 ** 23 for p in pats], dtype=torch.float64)
    assert torch.equal(ud, expect)                                        # k + 0.5 is exact in fp32
    assert float(ud.min()) == 2.0 ** -24 and float(ud.max()) == 1 - 2.0 ** -24
    assert bool(torch.isfinite(q).all()) and float(q.min()) > 0
    assert bool(torch.isfinite(g).all())


def test_last_pass_ignores_the_gumbel_draw():
    """temp = 4.5 * (1 - ratio) is 0 in the last pass (PTransformer.py:624-627): an infinite injected Gumbel value must not
    turn the confidence into NaN and leave mask ids in the output."""
    from probtalk_amd.weights import make_input
    dec = _predictor()
    B, Tq, T = 2, 11, 8
    N, V = Tq * 4, 128
    cond = make_input("seeded.cond3", (B, 768, Tq)).cuda()
    label = torch.tensor([1, 2]).cuda()
    qs, gs = zip(*[_noise(99, s, B, N, V) for s in range(T)])
    gs = torch.stack(gs)
    ref = dec.sample_good(label, cond, None, T=T, noise=(torch.stack(qs), gs))
    gs2 = gs.clone()
    gs2[T - 1] = float("inf")
    out = dec.sample_good(label, cond, None, T=T, noise=(torch.stack(qs), gs2))
    assert int(out.min()) >= 0 and int(out.max()) < V
    assert torch.equal(out, ref)
