"""Micro-benchmark on the GPU box: fused sub-layer tail (pt_gemm_bf16_sublayer) against the two launches it replaces
(pt_gemm_bf16_lnmod + pt_gemm_bf16_addln) over a sweep of row counts: the numbers behind the dispatch rule in
transformer.hip (whole rounds of the persistent grid) and DESIGN.md's table."""
import sys
import torch
sys.path.insert(0, ".")
from probtalk_amd import _lib

L = _lib.lib()
dev = "cuda"


def t_ms(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


def run(M, K1, rows_per_clip):
    g = torch.Generator(device=dev).manual_seed(0)
    bf = torch.bfloat16
    A = torch.randn(M, K1, device=dev, generator=g).to(bf)
    W1 = (torch.randn(512, K1, device=dev, generator=g) / K1 ** 0.5).to(bf)
    W2 = (torch.randn(512, 512, device=dev, generator=g) / 512 ** 0.5).to(bf)
    W2p = torch.empty_like(W2)
    L.pt_permute_k32(_lib.ptr(W2), _lib.ptr(W2p), 512, 512, _lib.stream())
    b1 = torch.zeros(512, device=dev); b2 = torch.zeros(512, device=dev)
    fold = torch.cat([torch.ones(4, 512), torch.zeros(4, 512)], 1).to(dev).contiguous()
    n_label = (M + rows_per_clip - 1) // rows_per_clip
    label = torch.randint(0, 4, (n_label,), device=dev, generator=g)
    gamma, beta = torch.ones(512, device=dev), torch.zeros(512, device=dev)
    x = torch.randn(M, 512, device=dev, generator=g).to(bf)
    hb = torch.empty(M, 512, dtype=bf, device=dev)
    s = _lib.stream()

    def fused():
        _lib.check(L.pt_gemm_bf16_sublayer(_lib.ptr(A), K1, K1, _lib.ptr(W1), _lib.ptr(b1), _lib.ptr(fold), _lib.ptr(label), n_label,
                                           rows_per_clip, 4, _lib.ptr(W2p), _lib.ptr(b2), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(x),
                                           M, s))

    def old():
        _lib.check(L.pt_gemm_bf16_lnmod(_lib.ptr(A), K1, _lib.ptr(W1), _lib.ptr(b1), M, K1, 512, _lib.ptr(fold), _lib.ptr(label), n_label,
                                        rows_per_clip, 4, _lib.ptr(hb), s))
        _lib.check(L.pt_gemm_bf16_addln(_lib.ptr(hb), 512, _lib.ptr(W2), _lib.ptr(b2), M, 512, 512, _lib.ptr(gamma), _lib.ptr(beta), None,
                                        _lib.ptr(x), s))

    tf, to = t_ms(fused), t_ms(old)
    fl = 2.0 * M * 512 * (K1 + 512)
    tiles = (M + 127) // 128
    print(f"M={M:7d} ({tiles:5d} tiles = {tiles / 256:5.2f} rounds) K1={K1:4d}: fused {tf*1e3:8.1f} us ({fl/tf/1e9:6.1f} TFLOP/s)   "
          f"two launches {to*1e3:8.1f} us ({fl/to/1e9:6.1f} TFLOP/s)   ratio {to/tf:5.2f}", flush=True)


if __name__ == "__main__":
    for K1 in (512, 2048):
        for M, rpc in ((37888, 148), (65536, 148), (75776, 148), (76800, 300), (98304, 300), (131072, 300), (153600, 300), (307200, 300)):
            run(M, K1, rpc)
