"""Random shapes through the small-batch bf16 GEMM against float64 (run on the GPU box)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from probtalk_amd import _lib
L = _lib.lib()
_lib.check(L.pt_set_option(b"sm_rows", 1 << 20))
random.seed(7)
worst = 0.0
for it in range(60):
    M = random.choice([1, 2, 15, 16, 17, 63, 64, 65, 127, 129, 148, 300, 1000, 1408, random.randint(1, 2900)])
    N = random.choice([8, 16, 24, 56, 64, 72, 128, 376, 512, 520, 1536, 2048, 8 * random.randint(1, 200)])
    K = 128 * random.randint(1, 16)
    relu = random.randint(0, 1)
    Np = (N + 127) // 128 * 128
    g = torch.Generator().manual_seed(it)
    A = torch.randn(M, K, generator=g).bfloat16()
    W = (torch.randn(Np, K, generator=g) * K ** -0.5).bfloat16()
    bias = torch.randn(Np, generator=g)
    ref = A.double() @ W[:N].double().t() + bias[:N].double()
    if relu:
        ref = ref.clamp_min(0)
    ldc = N + 8
    C = torch.full((M, ldc), 7.0, device="cuda")
    Cb = torch.full((M, ldc), 7.0, device="cuda", dtype=torch.bfloat16)
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()
    _lib.check(L.pt_gemm_bf16(_lib.ptr(Ad), K, _lib.ptr(Wd), _lib.ptr(bd), M, N, K, Np, _lib.ptr(C), ldc, _lib.ptr(Cb), ldc, relu,
                              _lib.stream()))
    torch.cuda.synchronize()
    scale = float(ref.abs().max()) or 1.0
    e = float((C[:, :N].cpu().double() - ref).abs().max()) / scale
    eb = float((Cb[:, :N].cpu().double() - ref).abs().max()) / scale
    ok = e < 2e-6 and eb < 5e-3 and bool((C[:, N:] == 7.0).all()) and bool((Cb[:, N:] == 7.0).all())
    worst = max(worst, e)
    if not ok:
        print("FAIL", M, N, K, relu, e, eb)
        sys.exit(1)
print("60 shapes ok, worst fp32 error", worst)
