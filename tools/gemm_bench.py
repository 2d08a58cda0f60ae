"""Micro-benchmark of the bf16 tap-GEMM at the transformer's shapes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probtalk_amd import _lib

L = _lib.lib()
shapes = [(9472, 512, 512), (9472, 1536, 512), (9472, 2048, 512), (9472, 512, 2048), (9472, 128, 512),
          (19200, 512, 512), (19200, 1536, 512), (19200, 2048, 512), (19200, 512, 2048), (19200, 1024, 512),
          (2368, 1024, 512), (8192, 8192, 8192)]
for (M, N, K) in shapes:
    Np = (N + 127) // 128 * 128
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = torch.randn(Np, K, device="cuda").bfloat16()
    bias = torch.randn(Np, device="cuda")
    for mode in ("f32out", "bf16out"):
        C = torch.empty(M, N, device="cuda") if mode == "f32out" else None
        Cb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if mode == "bf16out" else None
        def run():
            _lib.check(L.pt_gemm_bf16(_lib.ptr(A), K, _lib.ptr(W), _lib.ptr(bias), M, N, K, Np, _lib.ptr(C), N,
                                      _lib.ptr(Cb), N, 0, _lib.stream()))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        out = (C if C is not None else Cb.float())
        ref = A.float() @ W[:N].float().t() + bias[:N]
        err = float((out - ref).abs().max() / ref.abs().max())
        print(f"M={M:6d} N={N:5d} K={K:5d} {mode:8s} {us:9.1f} us  {2.0*M*N*K/us/1e6:8.1f} TF  err {err:.1e}", flush=True)
