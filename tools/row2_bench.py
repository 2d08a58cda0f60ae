"""A/B of the N = 512 row kernels (run on the GPU box): option row2 = 0 (one persistent 8-wave workgroup per CU, 160-row
tiles) against 1 (80-row tiles, two workgroups per CU), per epilogue, at the bench's row counts."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probtalk_amd import _lib

L = _lib.lib()
rows = [int(a) for a in sys.argv[1:]] or [37888, 75776, 76800, 153600]
for M in rows:
    for K in (512, 2048):
        A = torch.randn(M, K, device="cuda").bfloat16()
        W = (torch.randn(512, K, device="cuda") * K ** -0.5).bfloat16()
        bias, gamma, beta = (torch.randn(512, device="cuda") for _ in range(3))
        style = torch.randn(4, 1024, device="cuda")
        label = torch.randint(0, 4, (M // 148 + 1,), device="cuda")
        out = torch.empty(M, 512, device="cuda", dtype=torch.bfloat16)
        xb = torch.randn(M, 512, device="cuda").bfloat16()
        fns = {"std": lambda: _lib.check(L.pt_gemm_bf16(_lib.ptr(A), K, _lib.ptr(W), _lib.ptr(bias), M, 512, K, 512, None, 0,
                                                         _lib.ptr(out), 512, 0, _lib.stream())),
               "lnmod": lambda: _lib.check(L.pt_gemm_bf16_lnmod(_lib.ptr(A), K, _lib.ptr(W), _lib.ptr(bias), M, K, 512,
                                                                 _lib.ptr(style), _lib.ptr(label), M // 148 + 1, 148, 4,
                                                                 _lib.ptr(out), _lib.stream())),
               "addln_b": lambda: _lib.check(L.pt_gemm_bf16_addln(_lib.ptr(A), K, _lib.ptr(W), _lib.ptr(bias), M, K, 512,
                                                                   _lib.ptr(gamma), _lib.ptr(beta), None, _lib.ptr(xb),
                                                                   _lib.stream()))}
        for name, run in fns.items():
            if K == 2048 and name != "lnmod":
                continue
            line = f"M={M:6d} K={K:5d} {name:8s}"
            for mode in (0, 1):
                _lib.check(L.pt_set_option(b"row2", mode))
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 30
                e0.record()
                for _ in range(n):
                    run()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / n
                line += f"  row2={mode}: {us:8.1f} us {2.0 * M * 512 * K / us / 1e6:7.1f} TF"
            print(line, flush=True)
_lib.check(L.pt_set_option(b"row2", 0))
