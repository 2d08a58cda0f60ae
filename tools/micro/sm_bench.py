"""Small-batch bf16 GEMM (option sm_rows) against the kernels it replaces, per launch (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from probtalk_amd import _lib
L = _lib.lib()
shapes = [(1408, 512, 512), (1408, 1536, 512), (1408, 2048, 512), (1408, 512, 2048), (1408, 128, 512), (2368, 512, 512), (2368, 1536, 512),
          (2368, 512, 2048), (2880, 512, 512), (4800, 512, 512), (4800, 2048, 512), (9472, 512, 512), (9472, 1536, 512), (148, 512, 512), (148, 1536, 512), (148, 512, 2048)]
for (M, N, K) in shapes:
    Np = (N + 127) // 128 * 128
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(Np, K, device="cuda") * K ** -0.5).bfloat16()
    bias = torch.randn(Np, device="cuda")
    ref = A.float() @ W[:N].float().t() + bias[:N]
    line = f"M={M:5d} N={N:5d} K={K:5d}"
    for sm in (0, 1 << 20):
        _lib.check(L.pt_set_option(b"sm_rows", sm))
        for mode in ("f32", "bf16"):
            C = torch.zeros(M, N, device="cuda") if mode == "f32" else None
            Cb = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) if mode == "bf16" else None
            run = lambda: _lib.check(L.pt_gemm_bf16(_lib.ptr(A), K, _lib.ptr(W), _lib.ptr(bias), M, N, K, Np, _lib.ptr(C), N,
                                                    _lib.ptr(Cb), N, 0, _lib.stream()))
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 50
            out = C if C is not None else Cb.float()
            err = float((out - ref).abs().max() / ref.abs().max())
            line += f" | sm={'on ' if sm else 'off'} {mode:4s} {us:6.1f} us err {err:.1e}"
    print(line, flush=True)
