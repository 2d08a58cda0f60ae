"""A/B of the fp32 tap-GEMM kernels (run on the GPU box): option split_bf16 = 0 (fp32 MFMA) against 1 (three-term bf16
splits on the bf16 MFMA), Linear shapes of the transformers at the bench's row counts."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probtalk_amd import _lib

L = _lib.lib()
shapes = [(75776, 512, 512), (75776, 1536, 512), (75776, 2048, 512), (75776, 512, 2048), (153600, 512, 512), (9472, 512, 512),
          (2368, 1536, 512)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for (M, N, K) in shapes:
    Np = (N + 127) // 128 * 128
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(1, Np, K, device="cuda") * K ** -0.5
    bias = torch.randn(Np, device="cuda")
    C = torch.empty(M, N, device="cuda")
    line = f"M={M:6d} N={N:5d} K={K:5d}"
    for mode in (0, 1):
        _lib.check(L.pt_set_option(b"split_bf16", mode))

        def run():
            _lib.check(L.pt_conv1d_cl(_lib.ptr(A), K, _lib.ptr(W), _lib.ptr(bias), N, Np, K, 0, 1, M, _lib.ptr(C), N, 0, 1.0, None, 0,
                                      _lib.stream()))
        for _ in range(3):
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
        line += f"   split={mode}: {us:9.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF"
    print(line, flush=True)
_lib.check(L.pt_set_option(b"split_bf16", 1))
