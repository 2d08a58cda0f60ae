"""Random shapes through pt_train_wgrad (split kernel, default) against float64 autograd (run on the GPU box)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from probtalk_amd import _lib
L = _lib.lib()
KT = {0: 1, 1: 3, 2: 7, 3: 4, 4: 4}
random.seed(11)
worst = 0.0
for it in range(40):
    kind = random.choice([0, 0, 1, 2, 3, 4])
    B = random.choice([1, 2, 3, 8, 33])
    Lin = random.choice([2, 5, 8, 11, 22, 37, 88, 90]) if kind != 3 else random.choice([4, 6, 11, 22, 37, 88])
    Cin = random.choice([4, 30, 64, 100, 128, 129, 256, 377, 512])
    N = random.choice([4, 12, 64, 100, 128, 130, 256, 376, 512])
    kt = KT[kind]
    Lout = {0: Lin, 1: Lin, 2: Lin, 3: (Lin - 2) // 2 + 1, 4: 2 * Lin}[kind]
    g = torch.Generator().manual_seed(it)
    x = torch.randn(B, Cin, Lin, generator=g)
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
    ref = w.grad

    def rows(t):
        b, c, l = t.shape
        ld = (c + 31) // 32 * 32
        out = torch.zeros(b * l, ld)
        out[:, :c] = t.permute(0, 2, 1).reshape(b * l, c)
        return out.cuda()
    xd, dyd = rows(x), rows(dy)
    dw = torch.zeros(kt, N, Cin, device="cuda")
    _lib.check(L.pt_train_wgrad(_lib.ptr(dyd), dyd.shape[1], _lib.ptr(xd), xd.shape[1], _lib.ptr(dw), N, Cin, kind, B, Lin, _lib.stream()))
    got = (dw.permute(2, 1, 0) if kind == 4 else dw.permute(1, 2, 0)).cpu().double()
    e = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
    worst = max(worst, e)
    if e > 1e-4:
        print("FAIL", kind, B, Lin, Cin, N, e)
        sys.exit(1)
print("40 shapes ok, worst error", worst)
