"""Micro-benchmark of the bf16 tap-GEMM at the transformer's shapes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probtalk_amd import _lib

L = _lib.lib()
for kv in os.environ.get("PT_OPTIONS", "").split(","):       # e.g. PT_OPTIONS=gemm8_krot=1,store_overlap=0
    if kv:
        _lib.check(L.pt_set_option(kv.split("=")[0].encode(), int(kv.split("=")[1])))
shapes = [(9472, 512, 512), (9472, 1536, 512), (9472, 2048, 512), (9472, 512, 2048), (9472, 128, 512),
          (75776, 512, 512), (75776, 1536, 512), (75776, 2048, 512), (75776, 512, 2048), (75776, 128, 512),
          (153600, 512, 512), (153600, 1536, 512), (153600, 2048, 512), (153600, 512, 2048), (153600, 376, 1024),
          (18944, 1024, 512), (8192, 8192, 8192)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
only_ln = os.environ.get("GEMM_BENCH_LN_ONLY") == "1"
for (M, N, K) in ([] if only_ln else shapes):
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

# fused LayerNorm epilogues (N = 512)
ln_shapes = [(75776, 512), (75776, 2048), (153600, 512)]
if os.environ.get("GEMM_BENCH_LN_M"):
    ln_shapes = [(int(m), k) for m in os.environ["GEMM_BENCH_LN_M"].split(",") for k in (512, 2048)]
for (M, K) in ln_shapes:
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(512, K, device="cuda") * K ** -0.5).bfloat16()
    bias, gamma, beta = (torch.randn(512, device="cuda") for _ in range(3))
    style = torch.randn(4, 1024, device="cuda")
    label = torch.randint(0, 4, (M // 148 + 1,), device="cuda")
    out = torch.empty(M, 512, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(M, 512, device="cuda")
    fns = {"lnmod": lambda: _lib.check(L.pt_gemm_bf16_lnmod(_lib.ptr(A), K, _lib.ptr(W), _lib.ptr(bias), M, K, 512,
                                                             _lib.ptr(style), _lib.ptr(label), M // 148 + 1, 148, 4,
                                                             _lib.ptr(out), _lib.stream())),
           "addln": lambda: _lib.check(L.pt_gemm_bf16_addln(_lib.ptr(A), K, _lib.ptr(W), _lib.ptr(bias), M, K, 512, _lib.ptr(gamma),
                                                             _lib.ptr(beta), _lib.ptr(x), _lib.ptr(out), _lib.stream())),
           "addln_b": lambda: _lib.check(L.pt_gemm_bf16_addln(_lib.ptr(A), K, _lib.ptr(W), _lib.ptr(bias), M, K, 512, _lib.ptr(gamma),
                                                               _lib.ptr(beta), None, _lib.ptr(out), _lib.stream()))}
    for name, run in fns.items():
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
        print(f"M={M:6d} N=  512 K={K:5d} {name:8s} {us:9.1f} us  {2.0*M*512*K/us/1e6:8.1f} TF", flush=True)
