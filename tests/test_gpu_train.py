"""GPU parity of the training path: every train_ops operator (forward and backward, HIP kernels through the C ABI) against
float64 torch autograd on the CPU, the whole predictor train forward/backward against the train-step oracle and the
reference-generated fixture, and TrainWrapper.__call__ (clip + AdamW) against the oracle's optimizer restatement.
Dropout is off in the parity cases (the reference's masks come from torch's RNG stream); its kernel is checked on its
own statistics and on forward/backward mask replay.  Tolerance: 1e-4 relative to the tensor's max (north_star fp32)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import probtalk_oracle_train as OT
from probtalk_amd import _lib
from probtalk_amd import train_ops as T
from probtalk_amd import training
from tests.helpers import GOLDEN, ref_state_dict, rel_err
from tests.test_oracle_train_golden import grad_keys, train_case

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _rows(x_ncl, ld=None):
    """(B,C,L) -> channels-last rows [B*L, ld] with zero pad columns (CPU float32)."""
    B, C, L = x_ncl.shape
    ld = ld or (C + 31) // 32 * 32
    out = torch.zeros(B * L, ld)
    out[:, :C] = x_ncl.permute(0, 2, 1).reshape(B * L, C)
    return out


def _ncl(rows, B, C):
    return rows.reshape(B, -1, rows.shape[1])[:, :, :C].permute(0, 2, 1)


def _leaf(t):
    return t.detach().clone().cuda().requires_grad_(True)


@pytest.mark.parametrize("kind,Cin,N,B,L", [(0, 768, 256, 2, 37), (0, 512, 1536, 1, 90), (0, 384, 200, 3, 5),
                                             (1, 512, 1024, 2, 44), (1, 96, 64, 2, 7), (2, 512, 512, 2, 33),
                                             (3, 256, 256, 2, 44), (3, 512, 512, 3, 43), (3, 64, 96, 1, 6),
                                             (4, 512, 256, 2, 11), (4, 128, 376, 2, 9)])
@pytest.mark.parametrize("slope,use_res", [(-1.0, False), (0.2, True)])
def test_conv_forward_backward(kind, Cin, N, B, L, slope, use_res):
    g = torch.Generator().manual_seed(kind * 100 + Cin + L)
    kt = T._KT[kind]
    x = torch.randn(B, Cin, L, generator=g)
    w = torch.randn((Cin, N, kt) if kind == 4 else (N, Cin, kt), generator=g) * (Cin * kt) ** -0.5
    if kind == 0:
        w = w[..., 0]
    b = torch.randn(N, generator=g)
    Lout = {0: L, 1: L, 2: L, 3: (L - 2) // 2 + 1, 4: 2 * L}[kind]
    res = torch.randn(B, N, Lout, generator=g) if use_res else None
    dy = torch.randn(B, N, Lout, generator=g)
    # float64 reference
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    rr = res.double().requires_grad_(True) if use_res else None
    if kind == 0:
        yr = F.conv1d(xr, wr.unsqueeze(-1), br)
    elif kind in (1, 2):
        yr = F.conv1d(xr, wr, br, padding=kt // 2)
    elif kind == 3:
        yr = F.conv1d(xr, wr, br, stride=2, padding=1)
    else:
        yr = F.conv_transpose1d(xr, wr, br, stride=2, padding=1)
    if use_res:
        yr = yr + rr
    if slope >= 0:
        yr = F.leaky_relu(yr, slope)
    yr.backward(dy.double())
    # HIP
    xd, wd, bd = _leaf(_rows(x)), _leaf(w), _leaf(b)
    rd = _leaf(_rows(res)) if use_res else None
    y = T.conv(xd, wd, bd, kind, B, L, slope=slope, res=rd)
    assert y.shape[0] == B * Lout
    y.backward(_rows(dy, y.shape[1]).cuda())
    assert rel_err(_ncl(y.detach().cpu(), B, N), yr.detach()) < TOL
    assert rel_err(_ncl(xd.grad.cpu(), B, Cin), xr.grad) < TOL
    assert float(xd.grad[:, Cin:].abs().max()) == 0.0 if xd.shape[1] > Cin else True
    assert rel_err(wd.grad.cpu(), wr.grad) < TOL
    assert rel_err(bd.grad.cpu(), br.grad) < TOL
    if use_res:
        assert rel_err(_ncl(rd.grad.cpu(), B, N), rr.grad) < TOL


@pytest.mark.parametrize("kind,Cin,N,B,L", [(0, 377, 130, 3, 50), (0, 512, 512, 64, 88), (1, 256, 384, 40, 45),
                                             (3, 129, 256, 9, 37), (4, 256, 131, 16, 21)])
def test_weight_gradient_kernels_match_float64(kind, Cin, N, B, L, tap_kernel):
    """pt_train_wgrad directly, under both kernels (fp32 MFMA / three-term bf16 split with the transposing loader): ragged
    channel counts (partial float4 patches), row counts that are not multiples of the 32-row K-step, several m-slices
    combined by atomics, every tap geometry."""
    kt = T._KT[kind]
    Lout = {0: L, 1: L, 2: L, 3: (L - 2) // 2 + 1, 4: 2 * L}[kind]
    g = torch.Generator().manual_seed(Cin + N + L)
    x = torch.randn(B, Cin, L, generator=g)
    dy = torch.randn(B, N, Lout, generator=g)
    w = torch.zeros((Cin, N, kt) if kind == 4 else (N, Cin, kt), dtype=torch.float64, requires_grad=True)
    xr = x.double()
    if kind == 0:
        yr = F.conv1d(xr, w[..., :1])
    elif kind in (1, 2):
        yr = F.conv1d(xr, w, padding=kt // 2)
    elif kind == 3:
        yr = F.conv1d(xr, w, stride=2, padding=1)
    else:
        yr = F.conv_transpose1d(xr, w, stride=2, padding=1)
    yr.backward(dy.double())
    ref = w.grad if kind != 0 else w.grad[..., :1]
    xd, dyd = _rows(x).cuda(), _rows(dy).cuda()
    dw = torch.zeros(kt, N, Cin, device="cuda")
    L_ = _lib.lib()
    _lib.check(L_.pt_train_wgrad(_lib.ptr(dyd), dyd.shape[1], _lib.ptr(xd), xd.shape[1], _lib.ptr(dw), N, Cin, kind, B, L,
                                 _lib.stream()))
    got = dw.permute(2, 1, 0) if kind == 4 else dw.permute(1, 2, 0)
    assert rel_err(got.cpu(), ref) < TOL


@pytest.mark.parametrize("rows,use_res", [(77, True), (300, False)])
def test_layer_norm_res(rows, use_res):
    g = torch.Generator().manual_seed(rows)
    x, r = torch.randn(rows, 512, generator=g), torch.randn(rows, 512, generator=g)
    gm, bt = 1 + 0.3 * torch.randn(512, generator=g), 0.2 * torch.randn(512, generator=g)
    dy = torch.randn(rows, 512, generator=g)
    xr, rr, gr, br = (t.double().requires_grad_(True) for t in (x, r, gm, bt))
    yr = F.layer_norm(xr + rr if use_res else xr, (512,), gr, br, 1e-5)
    yr.backward(dy.double())
    xd, rd, gd, bd = _leaf(x), _leaf(r), _leaf(gm), _leaf(bt)
    y = T.layer_norm(xd, gd, bd, res=rd if use_res else None)
    y.backward(dy.cuda())
    assert rel_err(y.detach().cpu(), yr.detach()) < TOL
    assert rel_err(xd.grad.cpu(), xr.grad) < TOL
    if use_res:
        assert rel_err(rd.grad.cpu(), rr.grad) < TOL
    assert rel_err(gd.grad.cpu(), gr.grad) < TOL
    assert rel_err(bd.grad.cpu(), br.grad) < TOL


def test_modsilu():
    g = torch.Generator().manual_seed(3)
    rpc, B = 44, 5
    rows = rpc * B
    y, table = torch.randn(rows, 512, generator=g), 0.5 * torch.randn(4, 1024, generator=g)
    label = torch.tensor([1, 3, 0, 3, 2])
    dh = torch.randn(rows, 512, generator=g)
    yr, tr = y.double().requires_grad_(True), table.double().requires_grad_(True)
    lab = label.repeat_interleave(rpc)
    z = yr * (1 + tr[lab, :512]) + tr[lab, 512:]
    hr = z * torch.sigmoid(z)
    hr.backward(dh.double())
    yd, td = _leaf(y), _leaf(table)
    h = T.ModSiLU.apply(yd, td, label.cuda(), rpc)
    h.backward(dh.cuda())
    assert rel_err(h.detach().cpu(), hr.detach()) < TOL
    assert rel_err(yd.grad.cpu(), yr.grad) < TOL
    assert rel_err(td.grad.cpu(), tr.grad) < TOL


def _mha(q, k, v, B, Nq, Nk):
    qh = q.reshape(B, Nq, 8, 64).transpose(1, 2)
    kh = k.reshape(B, Nk, 8, 64).transpose(1, 2)
    vh = v.reshape(B, Nk, 8, 64).transpose(1, 2)
    a = torch.softmax(qh @ kh.transpose(-1, -2) / 8.0, dim=-1)
    return (a @ vh).transpose(1, 2).reshape(B * Nq, 512)


@pytest.mark.parametrize("B,N", [(3, 44), (2, 100), (1, 7), (2, 180), (1, 212), (1, 300)])   # 97..212 keys: query-blocked kernels
def test_attention_self(B, N):
    g = torch.Generator().manual_seed(N)
    qkv = torch.randn(B * N, 1536, generator=g)
    do = torch.randn(B * N, 512, generator=g)
    r = qkv.double().requires_grad_(True)
    o_ref = _mha(r[:, :512], r[:, 512:1024], r[:, 1024:], B, N, N)
    o_ref.backward(do.double())
    d = _leaf(qkv)
    o = T.attention_self(d, B, N)
    o.backward(do.cuda())
    assert rel_err(o.detach().cpu(), o_ref.detach()) < TOL
    assert rel_err(d.grad.cpu(), r.grad) < TOL


@pytest.mark.parametrize("B,Nq,Nk", [(2, 44, 11), (2, 100, 25), (1, 5, 3), (2, 300, 37), (1, 130, 180)])
def test_attention_cross(B, Nq, Nk):
    g = torch.Generator().manual_seed(Nq)
    q, kv = torch.randn(B * Nq, 512, generator=g), torch.randn(B * Nk, 1024, generator=g)
    do = torch.randn(B * Nq, 512, generator=g)
    qr, kvr = q.double().requires_grad_(True), kv.double().requires_grad_(True)
    o_ref = _mha(qr, kvr[:, :512], kvr[:, 512:], B, Nq, Nk)
    o_ref.backward(do.double())
    qd, kvd = _leaf(q), _leaf(kv)
    o = T.attention_cross(qd, kvd, B, Nq, Nk)
    o.backward(do.cuda())
    assert rel_err(o.detach().cpu(), o_ref.detach()) < TOL
    assert rel_err(qd.grad.cpu(), qr.grad) < TOL
    assert rel_err(kvd.grad.cpu(), kvr.grad) < TOL


def test_attention_dropout_replay_and_scale():
    """p > 0: the backward replays the forward's mask (finite-difference-free check: the op is linear in V, so
    dV from autograd must reproduce o for do = e_j), and the mean output stays unbiased."""
    B, N = 2, 44
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(B * N, 1536, generator=g).cuda()
    seeds = T._Seeds(7)
    outs = []
    for _ in range(24):
        outs.append(T.attention_self(qkv, B, N, 0.1, seeds))
    base = T.attention_self(qkv, B, N)
    mean = torch.stack(outs).mean(0)
    assert float((outs[0] - outs[1]).abs().max()) > 1e-3                       # different seeds, different masks
    assert float((mean - base).abs().mean()) < 0.08 * float(base.abs().mean()) + 0.02
    # linearity in V with a fixed seed: o(V1 + V2) = o(V1) + o(V2), and <do, o> = <dV, V>
    class Fixed:
        def next(self):
            return 12345
    d = qkv.clone().requires_grad_(True)
    o = T.attention_self(d, B, N, 0.1, Fixed())
    do = torch.randn(B * N, 512, generator=g).cuda()
    o.backward(do)
    lhs = float((do.double() * o.detach().double()).sum())
    rhs = float((d.grad[:, 1024:].double() * qkv[:, 1024:].double()).sum())
    assert abs(lhs - rhs) < 1e-4 * abs(lhs)


@pytest.mark.parametrize("B,Nq,Nk,p", [(3, 88, 88, 0.1), (2, 88, 22, 0.1), (1, 7, 3, 0.25), (2, 100, 100, 0.0), (1, 130, 40, 0.1),
                                       (2, 180, 180, 0.1), (1, 97, 212, 0.1), (2, 300, 37, 0.1), (1, 300, 300, 0.1)])
def test_attention_tiled_matches_rowwise(B, Nq, Nk, p):
    """The LDS-tiled (clip, head) kernels and the one-wave-per-row fallback share the dropout mask indexing: same
    outputs and gradients for the same seed.  100x100 .. 97x212 and 300x37 take the query-blocked kernels (K and V
    resident, queries in blocks), 300x300 exceeds their LDS budget: both runs take the fallback there, which checks the
    dispatch."""
    g = torch.Generator().manual_seed(Nq + Nk)
    q, kv = torch.randn(B * Nq, 512, generator=g).cuda(), torch.randn(B * Nk, 1024, generator=g).cuda()
    do = torch.randn(B * Nq, 512, generator=g).cuda()

    class Fixed:
        def next(self):
            return 99

    res = []
    for rowwise in (False, True):
        if rowwise:
            _lib.check(_lib.lib().pt_set_option(b"attn_train_rowwise", 1))
        try:
            qd, kvd = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
            o = T.Attention.apply(qd, kvd, 0, 0, 512, B, Nq, Nk, p, 99) if p > 0 else T.attention_cross(qd, kvd, B, Nq, Nk)
            o.backward(do)
            res.append((o.detach(), qd.grad, kvd.grad))
        finally:
            _lib.check(_lib.lib().pt_set_option(b"attn_train_rowwise", 0))
    for a, b_ in zip(*res):
        assert rel_err(a.cpu(), b_.cpu()) < 1e-5


def test_cross_entropy():
    g = torch.Generator().manual_seed(2)
    logits = 3 * torch.randn(352, 128, generator=g)
    tgt = torch.randint(0, 128, (352,), generator=g)
    lr = logits.double().requires_grad_(True)
    ref = F.cross_entropy(lr, tgt)
    (ref * 1.7).backward()
    ld = _leaf(logits)
    loss = T.CrossEntropy.apply(ld, tgt.cuda())
    (loss * 1.7).backward()
    assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref))
    assert rel_err(ld.grad.cpu(), lr.grad) < TOL


@pytest.mark.parametrize("rows,C,slope,use_res", [(176, 256, 0.1, False), (88, 256, 0.2, True), (33, 64, -1.0, False)])
def test_bn_train(rows, C, slope, use_res):
    g = torch.Generator().manual_seed(rows)
    x = 2 * torch.randn(rows, C, generator=g) + 0.5
    r = torch.randn(rows, C, generator=g)
    gm, bt = 1 + 0.3 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), 1 + torch.rand(C, generator=g)
    dy = torch.randn(rows, C, generator=g)
    xr, rr, gr, br = (t.double().requires_grad_(True) for t in (x, r, gm, bt))
    rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
    yr = F.batch_norm(xr, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5)
    if use_res:
        yr = yr + rr
    if slope >= 0:
        yr = F.leaky_relu(yr, slope)
    yr.backward(dy.double())
    xd, rd, gd, bd = _leaf(x), _leaf(r), _leaf(gm), _leaf(bt)
    rmd, rvd = rm.cuda(), rv.cuda()
    y = T.bn_train(xd, gd, bd, rmd, rvd, slope=slope, res=rd if use_res else None)
    y.backward(dy.cuda())
    assert rel_err(y.detach().cpu(), yr.detach()) < TOL
    assert rel_err(xd.grad.cpu(), xr.grad) < TOL
    assert rel_err(gd.grad.cpu(), gr.grad) < TOL
    assert rel_err(bd.grad.cpu(), br.grad) < TOL
    if use_res:
        assert rel_err(rd.grad.cpu(), rr.grad) < TOL
    assert rel_err(rmd.cpu(), rm_ref) < 1e-5 and rel_err(rvd.cpu(), rv_ref) < 1e-5


def test_gate():
    g = torch.Generator().manual_seed(4)
    hv = torch.randn(130, 1024, generator=g)
    do = torch.randn(130, 512, generator=g)
    r = hv.double().requires_grad_(True)
    ref = torch.tanh(r[:, :512]) * torch.sigmoid(r[:, 512:])
    ref.backward(do.double())
    d = _leaf(hv)
    o = T.Gate.apply(d)
    o.backward(do.cuda())
    assert rel_err(o.detach().cpu(), ref.detach()) < TOL
    assert rel_err(d.grad.cpu(), r.grad) < TOL


def test_embed_pe():
    from oracle.probtalk_oracle import sinusoid_table
    g = torch.Generator().manual_seed(6)
    B, Tq, G = 3, 11, 4
    N = Tq * G
    ids = torch.randint(0, 129, (B * N,), generator=g)
    tok = torch.randn(129, 512, generator=g)
    pe, ppe = sinusoid_table(512, 512), sinusoid_table(512, 512)
    dx = torch.randn(B * N, 512, generator=g)
    tr = tok.double().requires_grad_(True)
    pos = torch.arange(N)
    ref = tr[ids] + (pe[pos // G + 1] + ppe[pos % G + 1]).double().repeat(B, 1)
    ref.backward(dx.double())
    td = _leaf(tok)
    x = T.EmbedPE.apply(ids.int().cuda(), td, pe.cuda(), ppe.cuda(), N, G)
    x.backward(dx.cuda())
    assert rel_err(x.detach().cpu(), ref.detach()) < 1e-6
    assert rel_err(td.grad.cpu(), tr.grad) < TOL


def test_dropout_statistics_and_replay():
    x = torch.ones(1 << 20, device="cuda").requires_grad_(True)
    seeds = T._Seeds(3)
    y = T.dropout(x, 0.1, seeds)
    kept = (y != 0).float().mean().item()
    assert abs(kept - 0.9) < 3e-3
    assert abs(float(y.max()) - 1 / 0.9) < 1e-6
    y.backward(torch.ones_like(y))
    assert bool((x.grad == y.detach()).all())                                   # same mask, same scale
    y2 = T.dropout(x.detach(), 0.1, seeds)
    assert 0.15 < float((y2 != y.detach()).float().mean()) < 0.21               # independent masks: 2*0.1*0.9
    assert T.dropout(x, 0.0, seeds) is x


def test_clip_and_adamw_match_torch():
    g = torch.Generator().manual_seed(8)
    shapes = [(512, 768), (1024,), (129, 512), (3, 5, 7)]
    ps = [torch.randn(s, generator=g) for s in shapes]
    ref = [torch.nn.Parameter(p.double().clone()) for p in ps]
    opt_ref = torch.optim.AdamW(ref, lr=1e-3, betas=(0.9, 0.99))
    mine = [torch.nn.Parameter(p.clone().cuda()) for p in ps]
    opt = T.AdamW([{"params": mine}], lr=1e-3, betas=(0.9, 0.99))
    for step in range(3):
        gs = [torch.randn(s, generator=g) * (3.0 if step == 0 else 0.01) for s in shapes]
        for p, q, gr in zip(ref, mine, gs):
            p.grad, q.grad = gr.double().clone(), gr.clone().cuda()
        n_ref = torch.nn.utils.clip_grad_norm_(ref, 5.0)
        norm, coef = T.clip_grad_norm(mine, 5.0)
        assert abs(float(norm) - float(n_ref)) < 1e-5 * float(n_ref)
        o_norm, o_coef = OT.clip_coef(gs, 5.0)
        assert abs(float(coef) - float(o_coef)) < 1e-5
        opt_ref.step()
        opt.step({q: coef for q in mine})
        for p, q in zip(ref, mine):
            assert rel_err(q.detach().cpu(), p.detach()) < 1e-6
    sd = opt.state_dict()
    assert set(sd) == {"state", "param_groups"} and float(sd["state"][0]["step"]) == 3.0
    opt2 = T.AdamW([{"params": mine}], lr=1e-3, betas=(0.9, 0.99))
    opt2.load_state_dict(sd)
    assert opt2.state[mine[1]]["step"] == 3
    assert torch.equal(opt2.state[mine[1]]["exp_avg"], opt.state[mine[1]]["exp_avg"])


# ------------------------------------------------------------------------------------------------ whole train forward/backward
def _cuda_params(sd):
    out = {}
    for k, v in sd.items():
        t = v.detach().clone().cuda()
        if t.dtype == torch.float32 and "running" not in k and "position_encoding" not in k:
            t.requires_grad_(True)
        out[k] = t
    return out


def test_predictor_train_step_matches_oracle_and_fixture(tap_kernel):
    sa, sp = ref_state_dict("AudEnc"), ref_state_dict("Predictor")
    aud, inp, mask, label, state, keep, ratio = train_case()
    pre, orig_relu = [], torch.relu            # the oracle's ReLU inputs (6 x linear1, then output_conv.0), recorded in order
    def rec(x):
        pre.append(x.detach())
        return orig_relu(x)
    torch.relu = rec
    try:
        loss_o, grads_o, running_o, logits_o = OT.predictor_train_grads(sa, sp, aud, inp, mask, label, state, keep, ratio)
    finally:
        torch.relu = orig_relu
    Pa, Pp = _cuda_params(sa), _cuda_params(sp)
    training.relu_trace = []
    try:
        loss, logits = training.predictor_loss(Pa, Pp, aud.cuda(), inp.cuda(), mask.cuda(), label.cuda(), state.cuda(),
                                               keep.cuda(), ratio, p_drop=0.0, seeds=None, groups=4)
        trace = training.relu_trace
    finally:
        training.relu_trace = None
    loss.backward()
    assert abs(float(loss) - float(loss_o)) < 1e-5 * abs(float(loss_o))
    V = logits_o.shape[-1]
    assert rel_err(logits[:, :V].detach().cpu().reshape(logits_o.shape), logits_o) < TOL
    # ReLU is discontinuous in its derivative: a pre-activation within rounding noise of 0 may land on either side in
    # two correct fp32 implementations, and the unit's whole gradient row (dy * x) appears or vanishes.  Count the FFN
    # units whose sign differs between the HIP path and the oracle; each must be PROVEN to sit at 0 (|pre| < 1e-5 in the
    # oracle), and only then is the comparison widened from the strict bound.
    flips = 0
    for l in range(6):
        po = pre[l].reshape(-1, 2048)
        differ = ((trace[l].cpu() > 0) != (po > 0))
        assert float(po[differ].abs().max()) < 1e-5 if bool(differ.any()) else True, l
        flips += int(differ.sum())
    assert flips <= 4
    tol = 5e-4 if flips == 0 else tap_kernel.bound(3e-2, 5e-2)     # tests/helpers.py: tap_gemm_kernel
    worst, num, den = ("", 0.0), 0.0, 0.0
    gmax = max(float(v.abs().max()) for v in grads_o.values())
    for (tag, k), gref in grads_o.items():
        got = (Pa if tag == "AudEnc" else Pp)[k].grad
        if float(gref.abs().max()) < 1e-6 * gmax:
            # mathematically zero (a conv bias in front of a batch-statistic BatchNorm): rounding noise on both sides
            assert got is None or float(got.abs().max()) < 1e-5 * gmax, (tag, k)
            continue
        assert got is not None, (tag, k)
        e = rel_err(got.cpu(), gref)
        num += float((got.cpu().double() - gref.double()).pow(2).sum())
        den += float(gref.double().pow(2).sum())
        if e > worst[1]:
            worst = (f"{tag}.{k}", e)
    assert worst[1] < tol, worst
    assert math.sqrt(num / den) < (1e-4 if flips == 0 else tap_kernel.bound(2e-3, 1e-2))          # all gradients as one vector
    # BatchNorm running statistics follow nn.BatchNorm1d
    for k, v in running_o.items():
        assert rel_err(Pa[k].cpu(), v) < 1e-5, k
    # ... and the fixture produced by the imported reference
    gold = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "predictor_train_b2_t88.npz")).items()}
    assert abs(float(loss) - float(gold["loss"])) < 1e-5 * abs(float(gold["loss"]))
    for i, key in enumerate(grad_keys()):
        tag, k = key
        got = (Pa if tag == "AudEnc" else Pp)[k].grad.cpu()
        got = got[:8] if got.numel() > 20000 else got
        assert rel_err(got, gold[f"grad_{i}"]) < tol, key
    na, _ = T.clip_grad_norm([v for v in Pa.values() if v.requires_grad], 1e9)
    npd, _ = T.clip_grad_norm([v for v in Pp.values() if v.requires_grad], 1e9)
    ntol = 2e-4 if flips == 0 else 2e-3
    assert abs(float(na) - float(gold["grad_norm_audenc"])) < ntol * float(gold["grad_norm_audenc"])
    assert abs(float(npd) - float(gold["grad_norm_predictor"])) < ntol * float(gold["grad_norm_predictor"])


def _bat(B, T_, seed):
    from probtalk_amd.weights import make_input
    return {"aud_feat": make_input(f"tw.aud{seed}", (B, 768, T_)), "poses": 0.3 * make_input(f"tw.poses{seed}", (B, 330, T_)),
            "expression": 0.3 * make_input(f"tw.expr{seed}", (B, 100, T_)), "speaker": torch.tensor([1, 3, 0, 2][:B]),
            "epoch": 10}


def test_train_wrapper_step_matches_oracle():
    """TrainWrapper.__call__ with the two random draws injected and dropout off: loss, gradient norms and the updated
    parameters against the oracle's forward/backward + clip + AdamW on the CPU."""
    from oracle import probtalk_oracle as O
    from probtalk_amd import synthetic
    pred, _ = synthetic.build_pipeline(seed=0)
    pred.train_dropout = 0.0
    B, T_ = 2, 88
    bat = _bat(B, T_, 0)
    mask = torch.zeros(B, 1, T_)
    mask[1, :, 30:60] = 1
    g = torch.Generator().manual_seed(0)
    bat["_mask"], bat["_keep_scores"] = mask, torch.rand(B, 44, generator=g)
    sa = {k: v.detach().cpu().clone() for k, v in pred.AudEnc.state_dict().items()}
    sp = {k: v.detach().cpu().clone() for k, v in pred.Predictor.state_dict().items()}
    _, ld = pred(bat)
    # oracle side
    poses = torch.cat([bat["poses"][:, :6], bat["poses"][:, pred.c_index], bat["expression"]], 1)
    with torch.no_grad():
        state = pred.VQ.encode(gt_poses=poses.cuda())[1].loss.cpu()       # the frozen tokenizer's own (parity-tested) codes
    _, _, state_o = O.vqvae_encode({k: v.cpu() for k, v in pred.VQ.state_dict().items()}, poses)
    assert float((state == state_o).float().mean()) > 0.99                # near-ties aside (tests/test_gpu_vqvae.py)
    loss_o, grads_o, _, _ = OT.predictor_train_grads(sa, sp, bat["aud_feat"], poses * mask, mask, bat["speaker"], state,
                                                     bat["_keep_scores"], 0.5)
    assert abs(ld["loss"] - float(loss_o)) < 1e-5 * abs(float(loss_o))
    for tag, key, sd, mod in (("AudEnc", "grad_ae", sa, pred.AudEnc), ("Predictor", "grad_p", sp, pred.Predictor)):
        gl = {k: v for (t, k), v in grads_o.items() if t == tag}
        n, coef = OT.clip_coef(list(gl.values()), 5.0)
        assert abs(ld[key] - float(n)) < 2e-4 * float(n)
        new = dict(mod.named_parameters())
        for k, gr in gl.items():
            want, _, _ = OT.adamw_step(sd[k].double(), gr.double() * coef.double(), 0.0, 0.0, 1, 1e-4)
            # AdamW's first step moves every weight by about lr; compare the MOVE, not the weight
            move_ref, move = want - sd[k].double(), new[k].detach().cpu().double() - sd[k].double()
            assert float(move.abs().max()) < 1.01e-4 + 1e-4 * 1e-2 * float(sd[k].abs().max()), (tag, k)
            # ... and where the gradient is not rounding noise (m / sqrt(v) is a sign function on step 1)
            sure = gr.abs() > 1e-3 * gr.abs().max()
            if float(gr.abs().max()) > 1e-6 * float(n):                  # not a mathematically-zero gradient (bias before BN)
                assert float((move - move_ref)[sure].abs().max()) < 0.02 * 1e-4 + 1e-7, (tag, k)
    # the packed inference weights were dropped: the next inference call sees the updated parameters
    assert pred.Predictor.__dict__.get("_pt_handle") is None


def test_train_wrapper_loss_decreases_with_dropout():
    from probtalk_amd import synthetic
    pred, _ = synthetic.build_pipeline(seed=1)
    bat = _bat(4, 180, 1)                       # config.Data.pose.generate_length frames, like the reference's loader
    losses = []
    for _ in range(12):
        _, ld = pred(bat)
        assert math.isfinite(ld["loss"]) and math.isfinite(ld["grad_p"]) and math.isfinite(ld["grad_ae"])
        losses.append(ld["loss"])
    assert min(losses[-3:]) < losses[0] - 0.1, losses
    assert pred.global_step == 12
    sd = pred.state_dict()
    assert set(sd) == {"AudEnc", "Predictor", "TextEnc", "generator_optim", "dropout_seeds"}
    assert sd["dropout_seeds"]["n"] > 0
    # inference still works on the updated weights
    out, _ = pred.infer_on_batch(aud=bat["aud_feat"][:1].cuda(), text=None, id=torch.tensor([0]).cuda(), B=1,
                                 gt_poses=torch.zeros(1, 376, 180, device="cuda"), mask=torch.zeros(1, 1, 180, device="cuda"))
    assert torch.isfinite(out).all()
