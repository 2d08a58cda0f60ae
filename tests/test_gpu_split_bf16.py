"""GPU: the fp32 tap-GEMM on the bf16 matrix cores (option split_bf16, gemm.hip: both operands split into three bf16 terms,
six products) against float64, beside the fp32-MFMA kernel it replaces: fp32 accuracy is the claim, so the split kernel's
error is held to the fp32 tolerance AND to a small multiple of what the fp32 MFMA itself does on the same operands."""
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import rel_err
from tests.test_gpu_vqvae import _pack

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,B,L,Cin,N,scale", [(0, 64, 148, 512, 512, 1.0), (0, 16, 148, 2048, 512, 1.0), (0, 8, 300, 512, 2048, 1.0),
                                                  (1, 8, 88, 1024, 1024, 1.0), (3, 4, 75, 256, 512, 1.0), (4, 4, 37, 1024, 512, 1.0),
                                                  (2, 2, 50, 128, 256, 1.0), (0, 2, 37, 768, 512, 1e-12), (0, 2, 37, 768, 512, 1e12)])
def test_split_bf16_gemm_has_fp32_accuracy(kind, B, L, Cin, N, scale):
    from probtalk_amd import _lib
    lib = _lib.lib()
    kt = [1, 3, 7, 4, 4][kind]
    g = torch.Generator().manual_seed(kind * 1000 + Cin + N)
    x = torch.randn(B, Cin, L, generator=g) * scale
    w = torch.randn((Cin, N, kt) if kind == 4 else (N, Cin, kt), generator=g) / (Cin * kt) ** 0.5
    b = torch.randn(N, generator=g) * scale
    xd64, wd64, bd64 = x.double(), w.double(), b.double()
    if kind == 4:
        ref = F.conv_transpose1d(xd64, wd64, bd64, stride=2, padding=1)
    elif kind == 3:
        ref = F.conv1d(xd64, wd64, bd64, stride=2, padding=1)
    else:
        ref = F.conv1d(xd64, wd64, bd64, padding=kt // 2)
    Lout = ref.shape[2]
    wp, Np, Kp = _pack(w, kind)
    xd = x.cuda()
    a_cl = torch.empty(B * L, Kp, device="cuda")
    _lib.check(lib.pt_ncl_to_cl(_lib.ptr(xd), _lib.ptr(a_cl), B, Cin, L, Kp, _lib.stream()))
    ldc = (N + 31) // 32 * 32
    wd, bd = wp.cuda(), b.cuda()
    errs = {}
    try:
        for mode in (1, 0):
            _lib.check(lib.pt_set_option(b"split_bf16", mode))
            c_cl = torch.full((B * Lout, ldc), float("nan"), device="cuda")
            _lib.check(lib.pt_conv1d_cl(_lib.ptr(a_cl), Kp, _lib.ptr(wd), _lib.ptr(bd), N, Np, Kp, kind, B, L, _lib.ptr(c_cl),
                                        ldc, 0, 1.0, None, 0, _lib.stream()))
            out = c_cl[:, :N].reshape(B, Lout, N).permute(0, 2, 1).cpu()
            errs[mode] = rel_err(out, ref)
    finally:
        _lib.check(lib.pt_set_option(b"split_bf16", 1))
    print(f"kind {kind} K={Cin}x{kt} N={N} scale {scale:g}: split-bf16 {errs[1]:.2e}   fp32 MFMA {errs[0]:.2e}  (max-rel vs float64)")
    assert errs[1] < 2e-6, "split-bf16 GEMM is not fp32-accurate"
    assert errs[1] < 3 * errs[0] + 2e-7, "split-bf16 GEMM is much worse than the fp32 MFMA on the same operands"
