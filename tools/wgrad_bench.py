"""Weight-gradient kernel of the train step: fp32 MFMA (option split_bf16 = 0) against the three-term bf16 split (1) at the
predictor train step's shapes; error against float64 (run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from probtalk_amd import _lib

L = _lib.lib()
KT = {0: 1, 1: 3, 2: 7, 3: 4, 4: 4}
shapes = [(0, 512, 88, 512, 512), (0, 512, 88, 512, 1536), (0, 512, 88, 512, 2048), (0, 512, 88, 2048, 512),
          (0, 512, 88, 768, 512), (1, 512, 180, 512, 512), (1, 512, 22, 512, 1024), (3, 512, 180, 256, 512), (4, 64, 22, 512, 256)]
for kind, B, Lin, Cin, N in shapes:
    kt = KT[kind]
    Lout = {0: Lin, 1: Lin, 2: Lin, 3: (Lin - 2) // 2 + 1, 4: 2 * Lin}[kind]
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B * Lin, Cin, device="cuda", generator=g)
    dy = torch.randn(B * Lout, N, device="cuda", generator=g)
    ref = None
    if kind == 0:
        ref = (dy.double().t() @ x.double()).unsqueeze(0)               # [1][N][Cin]
    res = {}
    for opt in (0, 1):
        _lib.check(L.pt_set_option(b"split_bf16", opt))
        dw = torch.zeros(kt, N, Cin, device="cuda")

        def run():
            dw.zero_()
            _lib.check(L.pt_train_wgrad(_lib.ptr(dy), N, _lib.ptr(x), Cin, _lib.ptr(dw), N, Cin, kind, B, Lin, _lib.stream()))
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        res[opt] = dw.clone()
        err = float((dw.double() - ref).abs().max() / ref.abs().max()) if ref is not None else float("nan")
        flops = 2.0 * B * Lout * N * Cin * (kt if kind != 4 else 2)
        print(f"kind {kind} B={B} L={Lin} Cin={Cin} N={N} split={opt} {us:8.1f} us (incl. zero fill) {flops / us / 1e6:7.1f} TF-equiv  err vs f64 {err:.2e}", flush=True)
    d = float((res[0] - res[1]).abs().max() / res[0].abs().max())
    print(f"    split vs fp32-MFMA max diff / max {d:.2e}", flush=True)
_lib.check(L.pt_set_option(b"split_bf16", 1))
