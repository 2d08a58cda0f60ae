import sys, os
sys.path.insert(0, os.getcwd())
import torch
from probtalk_amd import _lib
L = _lib.lib()
for (M, N, K) in [(75776, 512, 2048), (75776, 512, 512), (75776, 2048, 512)]:
    for pad in (0, 64, 128):
        lda = K + pad
        Ab = torch.randn(M, lda, device="cuda").bfloat16()
        W = torch.randn(N, K, device="cuda").bfloat16()
        bias = torch.randn(N, device="cuda")
        Cb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        def run():
            _lib.check(L.pt_gemm_bf16(_lib.ptr(Ab), lda, _lib.ptr(W), _lib.ptr(bias), M, N, K, N, None, N, _lib.ptr(Cb), N, 0, _lib.stream()))
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        print(f"M={M} N={N} K={K} lda={lda}: {us:.1f} us {2.0*M*N*K/us/1e6:.0f} TF", flush=True)
