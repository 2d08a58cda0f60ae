import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
from probtalk_amd import _lib
import test_gpu_gemm_big as T
L = _lib.lib()
M, K = 75776, 512
A, W, bias, Np = T._mk(M, 512, K, 31)
g = torch.Generator().manual_seed(6)
gamma, beta = 1 + 0.3 * torch.randn(512, generator=g), 0.2 * torch.randn(512, generator=g)
style = 0.5 * torch.randn(4, 1024, generator=g)
rpc = 148
nclip = (M + rpc - 1) // rpc
label = torch.randint(0, 4, (nclip,), generator=g)
x0 = torch.randn(M, 512, generator=g)
fold = torch.cat([gamma * (1 + style[:, :512]), beta * (1 + style[:, :512]) + style[:, 512:]], dim=1).contiguous()
Ad, Wd, bd, sd, ld, gd, btd, x0d = (t.cuda() for t in (A, W, bias, fold, label, gamma, beta, x0))
res = {}
for mode in (0, 1, 1, 1):
    _lib.check(L.pt_set_option(b"row2", mode))
    out = torch.empty(M, 512, device="cuda", dtype=torch.bfloat16)
    _lib.check(L.pt_gemm_bf16_lnmod(_lib.ptr(Ad), K, _lib.ptr(Wd), _lib.ptr(bd), M, K, Np, _lib.ptr(sd), _lib.ptr(ld), nclip, rpc, 4,
                                    _lib.ptr(out), _lib.stream()))
    xd = x0d.clone()
    xb = torch.empty(M, 512, device="cuda", dtype=torch.bfloat16)
    _lib.check(L.pt_gemm_bf16_addln(_lib.ptr(Ad), K, _lib.ptr(Wd), _lib.ptr(bd), M, K, Np, _lib.ptr(gd), _lib.ptr(btd), _lib.ptr(xd),
                                    _lib.ptr(xb), _lib.stream()))
    xr = x0d.bfloat16()
    _lib.check(L.pt_gemm_bf16_addln(_lib.ptr(Ad), K, _lib.ptr(Wd), _lib.ptr(bd), M, K, Np, _lib.ptr(gd), _lib.ptr(btd), None,
                                    _lib.ptr(xr), _lib.stream()))
    torch.cuda.synchronize()
    cur = (out, xd, xb, xr)
    if mode == 0:
        base = cur
    else:
        for name, a, b in zip(("lnmod", "addln x", "addln xb", "addln_b"), base, cur):
            d = (a.float() - b.float()).abs()
            bad = (d > 0).nonzero()
            print(name, "differing", bad.shape[0], "max", float(d.max()))
            if bad.shape[0]:
                rows = bad[:, 0].unique()
                print("   rows", rows[:30].tolist(), "n rows", rows.numel(), " tiles", (rows // 80).unique()[:20].tolist())
                print("   cols", bad[:, 1].unique()[:40].tolist(), "n cols", bad[:, 1].unique().numel())
