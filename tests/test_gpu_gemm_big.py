"""GPU tests of the large-tile bf16 GEMM (gemm_bf16_big.hip) through the C ABI: plain epilogue and the two fused
LayerNorm epilogues, against a float64 PyTorch reference fed the SAME bf16-rounded operands (so only accumulation
order and the bf16 rounding of the output differ)."""
import pytest
import torch

from probtalk_amd import _lib

pytestmark = pytest.mark.gpu


def _mk(M, N, K, seed):
    g = torch.Generator().manual_seed(seed)
    Np = (N + 127) // 128 * 128
    A = torch.randn(M, K, generator=g).bfloat16()
    W = (torch.randn(Np, K, generator=g) * K ** -0.5).bfloat16()
    bias = torch.randn(Np, generator=g)
    return A, W, bias, Np


def _ln(v, g, b):
    mu = v.mean(-1, keepdim=True)
    var = ((v - mu) ** 2).mean(-1, keepdim=True)
    return (v - mu) / torch.sqrt(var + 1e-5) * g + b


@pytest.mark.parametrize("M,N,K,relu", [(300, 512, 512, 0), (1000, 1536, 512, 0), (777, 2048, 512, 1),
                                        (513, 512, 2048, 0), (260, 376, 1024, 0), (4096, 1024, 832, 1)])
def test_big_gemm_plain(M, N, K, relu):
    L = _lib.lib()
    A, W, bias, Np = _mk(M, N, K, 1 + M)
    ref = A.double() @ W[:N].double().t() + bias[:N].double()
    if relu:
        ref = ref.clamp_min(0)
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()
    ldc = N + 8
    C = torch.full((M, ldc), 7.0, device="cuda")
    Cb = torch.full((M, ldc), 7.0, device="cuda", dtype=torch.bfloat16)
    _lib.check(L.pt_gemm_bf16(_lib.ptr(Ad), K, _lib.ptr(Wd), _lib.ptr(bd), M, N, K, Np, _lib.ptr(C), ldc,
                              _lib.ptr(Cb), ldc, relu, _lib.stream()))
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    assert float((C[:, :N].cpu().double() - ref).abs().max()) / scale < 2e-6
    assert float((Cb[:, :N].cpu().double() - ref).abs().max()) / scale < 5e-3          # one bf16 rounding
    assert bool((C[:, N:] == 7.0).all()) and bool((Cb[:, N:] == 7.0).all())            # nothing outside [0, N)


@pytest.mark.parametrize("M,K,ncls", [(148 * 3, 512, 4), (300 * 2 + 17, 2048, 4), (148 * 5, 512, 6), (9472, 512, 4),
                                      (20000, 2048, 4)])
def test_big_gemm_lnmod(M, K, ncls):
    """ncls = 6 takes the two-stage kernel (the phase-pipelined row kernel keeps at most 4 classes of tables in LDS)."""
    L = _lib.lib()
    A, W, bias, Np = _mk(M, 512, K, 11)
    g = torch.Generator().manual_seed(5)
    gamma, beta = 1 + 0.3 * torch.randn(512, generator=g), 0.2 * torch.randn(512, generator=g)
    style = 0.5 * torch.randn(ncls, 1024, generator=g)
    rpc = 148
    nclip = (M + rpc - 1) // rpc
    label = torch.randint(0, ncls, (nclip,), generator=g)
    v = A.double() @ W[:512].double().t() + bias[:512].double()
    lab_row = label[torch.arange(M) // rpc]
    y = _ln(v, gamma.double(), beta.double()) * (1 + style[lab_row, :512].double()) + style[lab_row, 512:].double()
    ref = y * torch.sigmoid(y)
    out = torch.zeros(M, 512, device="cuda", dtype=torch.bfloat16)
    # the entry point takes the LayerNorm affine folded into the per-class table: gamma(1+scale) | beta(1+scale)+shift
    fold = torch.cat([gamma * (1 + style[:, :512]), beta * (1 + style[:, :512]) + style[:, 512:]], dim=1).contiguous()
    Ad, Wd, bd, sd, ld = (t.cuda() for t in (A, W, bias, fold, label))   # keep them alive
    _lib.check(L.pt_gemm_bf16_lnmod(_lib.ptr(Ad), K, _lib.ptr(Wd), _lib.ptr(bd), M, K, Np, _lib.ptr(sd), _lib.ptr(ld),
                                    nclip, rpc, ncls, _lib.ptr(out), _lib.stream()))
    torch.cuda.synchronize()
    assert float((out.cpu().double() - ref).abs().max()) / float(ref.abs().max()) < 5e-3
    # a single label applies to every row
    out1 = torch.zeros(M, 512, device="cuda", dtype=torch.bfloat16)
    one = torch.tensor([2]).cuda()
    _lib.check(L.pt_gemm_bf16_lnmod(_lib.ptr(Ad), K, _lib.ptr(Wd), _lib.ptr(bd), M, K, Np, _lib.ptr(sd), _lib.ptr(one),
                                    1, rpc, ncls, _lib.ptr(out1), _lib.stream()))
    torch.cuda.synchronize()
    y1 = _ln(v, gamma.double(), beta.double()) * (1 + style[2, :512].double()) + style[2, 512:].double()
    ref1 = y1 * torch.sigmoid(y1)
    assert float((out1.cpu().double() - ref1).abs().max()) / float(ref1.abs().max()) < 5e-3


@pytest.mark.parametrize("M,K", [(148 * 2 + 5, 512), (1024, 512), (9472, 2048), (20000, 512)])   # tile heights 32, 32, 64, 128
def test_big_gemm_addln(M, K):
    L = _lib.lib()
    A, W, bias, Np = _mk(M, 512, K, 23)
    g = torch.Generator().manual_seed(9)
    gamma, beta = 1 + 0.3 * torch.randn(512, generator=g), 0.2 * torch.randn(512, generator=g)
    x = torch.randn(M, 512, generator=g)
    ref = _ln(x.double() + A.double() @ W[:512].double().t() + bias[:512].double(), gamma.double(), beta.double())
    xd = x.cuda()
    xb = torch.zeros(M, 512, device="cuda", dtype=torch.bfloat16)
    Ad, Wd, bd, gd, btd = (t.cuda() for t in (A, W, bias, gamma, beta))   # keep them alive
    _lib.check(L.pt_gemm_bf16_addln(_lib.ptr(Ad), K, _lib.ptr(Wd), _lib.ptr(bd), M, K, Np, _lib.ptr(gd), _lib.ptr(btd),
                                    _lib.ptr(xd), _lib.ptr(xb), _lib.stream()))
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    assert float((xd.cpu().double() - ref).abs().max()) / scale < 5e-6
    assert float((xb.cpu().double() - ref).abs().max()) / scale < 5e-3
    # bf16 residual stream: x == NULL, the residual is the bf16 row that is overwritten
    xr = x.bfloat16()
    refb = _ln(xr.double() + A.double() @ W[:512].double().t() + bias[:512].double(), gamma.double(), beta.double())
    xbd = xr.cuda()
    _lib.check(L.pt_gemm_bf16_addln(_lib.ptr(Ad), K, _lib.ptr(Wd), _lib.ptr(bd), M, K, Np, _lib.ptr(gd), _lib.ptr(btd),
                                    None, _lib.ptr(xbd), _lib.stream()))
    torch.cuda.synchronize()
    assert float((xbd.cpu().double() - refb).abs().max()) / float(refb.abs().max()) < 5e-3


def test_big_gemm_rejects_bad_geometry():
    L = _lib.lib()
    A = torch.zeros(64, 96, device="cuda", dtype=torch.bfloat16)
    W = torch.zeros(512, 96, device="cuda", dtype=torch.bfloat16)
    C = torch.zeros(64, 512, device="cuda")
    rc = L.pt_gemm_bf16(_lib.ptr(A), 96, _lib.ptr(W), None, 64, 512, 96, 512, _lib.ptr(C), 512, None, 0, 0,
                        _lib.stream())
    assert rc != 0 and b"multiple of 64" in L.pt_last_error()


# ---- persistent paths: many tiles per workgroup (tile-boundary waits, the K-tile stream running across tiles, the
# draining tail), checked at the bench's row counts against a float64 reference computed on the GPU, and repeated:
# the kernels are deterministic, so any difference between runs is a staging race.
def _ref64(A, W, bias):
    return A.double() @ W.double().t() + bias.double()


@pytest.mark.parametrize("M,N,K", [(75776, 1536, 512), (40000, 2048, 512), (75776, 512, 512), (30001, 512, 2048)])
def test_persistent_plain_matches_reference_and_repeats(M, N, K):
    L = _lib.lib()
    A, W, bias, Np = _mk(M, N, K, 3)
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()
    ref = _ref64(Ad, Wd[:N], bd[:N])
    scale = float(ref.abs().max())
    first = None
    for rep in range(4):
        C = torch.empty(M, N, device="cuda")
        Cb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        _lib.check(L.pt_gemm_bf16(_lib.ptr(Ad), K, _lib.ptr(Wd), _lib.ptr(bd), M, N, K, Np, _lib.ptr(C), N, _lib.ptr(Cb), N, 0,
                                  _lib.stream()))
        torch.cuda.synchronize()
        if first is None:
            first = (C, Cb)
            assert float((C.double() - ref).abs().max()) / scale < 2e-6
            assert float((Cb.double() - ref).abs().max()) / scale < 5e-3
        else:
            assert torch.equal(C, first[0]) and torch.equal(Cb, first[1]), f"run {rep} differs from run 0"


@pytest.mark.parametrize("M,K", [(75776, 512), (75776, 2048), (153600, 512)])
def test_persistent_ln_epilogues_match_reference_and_repeat(M, K):
    L = _lib.lib()
    A, W, bias, Np = _mk(M, 512, K, 31)
    g = torch.Generator().manual_seed(6)
    gamma, beta = 1 + 0.3 * torch.randn(512, generator=g), 0.2 * torch.randn(512, generator=g)
    style = 0.5 * torch.randn(4, 1024, generator=g)
    rpc = 148
    nclip = (M + rpc - 1) // rpc
    label = torch.randint(0, 4, (nclip,), generator=g)
    x0 = torch.randn(M, 512, generator=g)
    fold = torch.cat([gamma * (1 + style[:, :512]), beta * (1 + style[:, :512]) + style[:, 512:]], dim=1).contiguous()
    Ad, Wd, bd, sd, ld, gd, btd, x0d = (t.cuda() for t in (A, W, bias, fold, label, gamma, beta, x0))
    v = _ref64(Ad, Wd[:512], bd[:512])
    lab_row = ld[torch.arange(M, device="cuda") // rpc]
    y = _ln(v, gd.double(), btd.double()) * (1 + style.cuda()[lab_row, :512].double()) + style.cuda()[lab_row, 512:].double()
    ref_mod = y * torch.sigmoid(y)
    ref_add = _ln(x0d.double() + v, gd.double(), btd.double())
    first = None
    for rep in range(3):
        out = torch.empty(M, 512, device="cuda", dtype=torch.bfloat16)
        _lib.check(L.pt_gemm_bf16_lnmod(_lib.ptr(Ad), K, _lib.ptr(Wd), _lib.ptr(bd), M, K, Np, _lib.ptr(sd), _lib.ptr(ld), nclip, rpc, 4,
                                        _lib.ptr(out), _lib.stream()))
        xd = x0d.clone()
        xb = torch.empty(M, 512, device="cuda", dtype=torch.bfloat16)
        _lib.check(L.pt_gemm_bf16_addln(_lib.ptr(Ad), K, _lib.ptr(Wd), _lib.ptr(bd), M, K, Np, _lib.ptr(gd), _lib.ptr(btd), _lib.ptr(xd),
                                        _lib.ptr(xb), _lib.stream()))
        xr = x0d.bfloat16()                                     # bf16 residual stream, updated in place
        _lib.check(L.pt_gemm_bf16_addln(_lib.ptr(Ad), K, _lib.ptr(Wd), _lib.ptr(bd), M, K, Np, _lib.ptr(gd), _lib.ptr(btd), None,
                                        _lib.ptr(xr), _lib.stream()))
        torch.cuda.synchronize()
        if first is None:
            first = (out, xd, xb, xr)
            assert float((out.double() - ref_mod).abs().max()) / float(ref_mod.abs().max()) < 5e-3
            sc = float(ref_add.abs().max())
            assert float((xd.double() - ref_add).abs().max()) / sc < 5e-6
            assert float((xb.double() - ref_add).abs().max()) / sc < 5e-3
            ref_b = _ln(x0d.bfloat16().double() + v, gd.double(), btd.double())
            assert float((xr.double() - ref_b).abs().max()) / float(ref_b.abs().max()) < 5e-3
        else:
            assert torch.equal(out, first[0]) and torch.equal(xd, first[1]) and torch.equal(xb, first[2]) and \
                torch.equal(xr, first[3]), f"run {rep} differs"


# ---- bf16-only output of interior tiles with a successor: the tile's upper half is stored between the phases of the last
# K-tile (store_overlap).  Same accumulators, same rounding: the result equals the stores-after-the-loop path bit for bit.
@pytest.mark.parametrize("M,N,K,relu", [(75776, 2048, 512, 1), (75776, 1536, 512, 0), (40000, 1024, 128, 1),
                                        (30001, 1536, 2048, 0), (66000, 376, 512, 0)])
def test_early_quadrant_stores_equal_late_stores(M, N, K, relu):
    L = _lib.lib()
    A, W, bias, Np = _mk(M, N, K, 17)
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()
    ref = _ref64(Ad, Wd[:N], bd[:N])
    if relu:
        ref = ref.clamp_min(0)
    scale = float(ref.abs().max())
    ldc = N + 8
    outs = []
    try:
        for mode in (1, 0, 1):
            _lib.check(L.pt_set_option(b"store_overlap", mode))
            Cb = torch.full((M, ldc), 7.0, device="cuda", dtype=torch.bfloat16)
            _lib.check(L.pt_gemm_bf16(_lib.ptr(Ad), K, _lib.ptr(Wd), _lib.ptr(bd), M, N, K, Np, None, 0, _lib.ptr(Cb), ldc, relu,
                                      _lib.stream()))
            torch.cuda.synchronize()
            outs.append(Cb)
    finally:
        _lib.check(L.pt_set_option(b"store_overlap", 1))
    assert float((outs[0][:, :N].double() - ref).abs().max()) / scale < 5e-3
    assert bool((outs[0][:, N:] == 7.0).all())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


# ---- option row2 = 1: N = 512 Linears from 20,480 rows on run as 80-row tiles, two workgroups per CU
# (gemm_bf16_row2.hip).  Same K order, same accumulator layout, same epilogue arithmetic as the one-workgroup-per-CU row
# kernel: bit-identical results.  (Its first version exposed a race both kernels shared: the LayerNorm partial sums were
# written to LDS with no lgkmcnt wait in front of the bare s_barrier.)
@pytest.mark.parametrize("M,K", [(40003, 512), (37888, 2048), (20480, 64)])
def test_two_workgroups_per_cu_row_kernel_equals_persistent_row_kernel(M, K):
    L = _lib.lib()
    A, W, bias, Np = _mk(M, 512, K, 41)
    g = torch.Generator().manual_seed(8)
    gamma, beta = 1 + 0.3 * torch.randn(512, generator=g), 0.2 * torch.randn(512, generator=g)
    style = 0.5 * torch.randn(3, 1024, generator=g)
    rpc = 148
    nclip = (M + rpc - 1) // rpc
    label = torch.randint(0, 3, (nclip,), generator=g)
    x0 = torch.randn(M, 512, generator=g)
    fold = torch.cat([gamma * (1 + style[:, :512]), beta * (1 + style[:, :512]) + style[:, 512:]], dim=1).contiguous()
    Ad, Wd, bd, sd, ld, gd, btd, x0d = (t.cuda() for t in (A, W, bias, fold, label, gamma, beta, x0))
    res = []
    try:
        for mode in (1, 0):
            _lib.check(L.pt_set_option(b"row2", mode))
            C = torch.full((M, 520), 7.0, device="cuda")
            Cb = torch.full((M, 520), 7.0, device="cuda", dtype=torch.bfloat16)
            _lib.check(L.pt_gemm_bf16(_lib.ptr(Ad), K, _lib.ptr(Wd), _lib.ptr(bd), M, 512, K, Np, _lib.ptr(C), 520, _lib.ptr(Cb), 520, 1,
                                      _lib.stream()))
            out = torch.empty(M, 512, device="cuda", dtype=torch.bfloat16)
            _lib.check(L.pt_gemm_bf16_lnmod(_lib.ptr(Ad), K, _lib.ptr(Wd), _lib.ptr(bd), M, K, Np, _lib.ptr(sd), _lib.ptr(ld), nclip, rpc, 3,
                                            _lib.ptr(out), _lib.stream()))
            xd = x0d.clone()
            xb = torch.empty(M, 512, device="cuda", dtype=torch.bfloat16)
            _lib.check(L.pt_gemm_bf16_addln(_lib.ptr(Ad), K, _lib.ptr(Wd), _lib.ptr(bd), M, K, Np, _lib.ptr(gd), _lib.ptr(btd), _lib.ptr(xd),
                                            _lib.ptr(xb), _lib.stream()))
            xr = x0d.bfloat16()
            _lib.check(L.pt_gemm_bf16_addln(_lib.ptr(Ad), K, _lib.ptr(Wd), _lib.ptr(bd), M, K, Np, _lib.ptr(gd), _lib.ptr(btd), None,
                                            _lib.ptr(xr), _lib.stream()))
            torch.cuda.synchronize()
            res.append((C, Cb, out, xd, xb, xr))
    finally:
        _lib.check(L.pt_set_option(b"row2", 0))
    ref = (Ad.double() @ Wd[:512].double().t() + bd[:512].double()).clamp_min(0)
    assert float((res[0][0][:, :512].double() - ref).abs().max()) / float(ref.abs().max()) < 2e-6
    assert bool((res[0][0][:, 512:] == 7.0).all()) and bool((res[0][1][:, 512:] == 7.0).all())
    for name, a, b in zip(("fp32", "bf16", "lnmod", "addln x", "addln xb", "addln bf16 stream"), res[0], res[1]):
        assert torch.equal(a, b), f"{name}: two-workgroup kernel differs from the persistent row kernel"


@pytest.mark.parametrize("M,N,K,relu", [(1408, 512, 512, 0), (1408, 1536, 512, 0), (1408, 2048, 512, 1), (1408, 512, 2048, 0),
                                        (148, 128, 512, 0), (1, 512, 512, 0), (65, 72, 128, 1), (2368, 376, 1024, 0),
                                        (3000, 512, 768, 0)])
def test_small_batch_kernel_matches_reference_and_the_persistent_kernels(M, N, K, relu):
    """gemm_sm_kernel (option sm_rows: 64 x 64 tiles, the operands of the whole K requested up front, ring of four 128-deep
    groups beyond K = 512) against float64 and against the kernels it replaces at few rows (sm_rows = 0): ragged M and N, one
    row, K = 128 (one group) to 2048 (ring wraps three times), both outputs, padded leading dimension untouched."""
    from tests.helpers import lib_option
    L = _lib.lib()
    A, W, bias, Np = _mk(M, N, K, 3 + M + N)
    ref = A.double() @ W[:N].double().t() + bias[:N].double()
    if relu:
        ref = ref.clamp_min(0)
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()
    ldc = N + 8
    outs = {}
    for sm in (0, 1 << 20):
        with lib_option("sm_rows", sm):
            C = torch.full((M, ldc), 7.0, device="cuda")
            Cb = torch.full((M, ldc), 7.0, device="cuda", dtype=torch.bfloat16)
            _lib.check(L.pt_gemm_bf16(_lib.ptr(Ad), K, _lib.ptr(Wd), _lib.ptr(bd), M, N, K, Np, _lib.ptr(C), ldc,
                                      _lib.ptr(Cb), ldc, relu, _lib.stream()))
            torch.cuda.synchronize()
        scale = float(ref.abs().max())
        assert float((C[:, :N].cpu().double() - ref).abs().max()) / scale < 2e-6
        assert float((Cb[:, :N].cpu().double() - ref).abs().max()) / scale < 5e-3
        assert bool((C[:, N:] == 7.0).all()) and bool((Cb[:, N:] == 7.0).all())
        outs[sm] = C[:, :N].clone()
    assert float((outs[0] - outs[1 << 20]).abs().max()) / float(ref.abs().max()) < 2e-6
