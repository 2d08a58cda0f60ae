"""Debug aid: phase timeline of the persistent bf16 GEMM workgroups (pt_debug_set_stamps), run on the GPU box.
python tools/gemm_stamps.py [M N K] [mode]   mode in {std, lnmod, addln}"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probtalk_amd import _lib

L = _lib.lib()
L.pt_debug_set_stamps.argtypes = [C.c_void_p]
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (75776, 512, 512)
mode = sys.argv[4] if len(sys.argv) > 4 else "std"
A = torch.randn(M, K, device="cuda").bfloat16()
Np = (N + 127) // 128 * 128
W = (torch.randn(Np, K, device="cuda") * K ** -0.5).bfloat16()
bias, gamma, beta = (torch.randn(Np, device="cuda") for _ in range(3))
style = torch.randn(4, 1024, device="cuda")
label = torch.randint(0, 4, (M // 148 + 1,), device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
x = torch.randn(M, 512, device="cuda")
stamps = torch.zeros(256 * 64, dtype=torch.int64, device="cuda")


def run():
    if mode == "std":
        _lib.check(L.pt_gemm_bf16(_lib.ptr(A), K, _lib.ptr(W), _lib.ptr(bias), M, N, K, Np, None, 0, _lib.ptr(out), N, 0, _lib.stream()))
    elif mode == "lnmod":
        _lib.check(L.pt_gemm_bf16_lnmod(_lib.ptr(A), K, _lib.ptr(W), _lib.ptr(bias), M, K, Np, _lib.ptr(style), _lib.ptr(label),
                                        M // 148 + 1, 148, 4, _lib.ptr(out), _lib.stream()))
    else:
        _lib.check(L.pt_gemm_bf16_addln(_lib.ptr(A), K, _lib.ptr(W), _lib.ptr(bias), M, K, Np, _lib.ptr(gamma), _lib.ptr(beta),
                                        _lib.ptr(x), _lib.ptr(out), _lib.stream()))


for _ in range(5):
    run()
torch.cuda.synchronize()
L.pt_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
run()
torch.cuda.synchronize()
L.pt_debug_set_stamps(None)
s = stamps.view(256, 64).cpu()
t0 = int(s[s > 0].min())
names = ["start", "k0-landed", "main-done", "math-done", "stores-issued"]
for b in (0, 1, 8, 100, 255):
    row = [(int(v) - t0) / 100.0 for v in s[b] if v > 0]
    print(f"wg {b:3d}: " + "  ".join(f"{names[i % 5][:5]}={v:6.2f}" for i, v in enumerate(row)))
# mean phase durations over all workgroups, first tile and second tile
import statistics
for tile in range(3):
    d = []
    for b in range(256):
        r = s[b, tile * 5:(tile + 1) * 5 + 1].tolist()
        if r[4] > 0:
            nxt = r[5] if r[5] > 0 else r[4]
            d.append([(r[1] - r[0]) / 100, (r[2] - r[1]) / 100, (r[3] - r[2]) / 100, (r[4] - r[3]) / 100, (nxt - r[4]) / 100])
    if d:
        m = [statistics.mean(c) for c in zip(*d)]
        print(f"tile {tile} (n={len(d)}): wait-k0 {m[0]:.2f}  main {m[1]:.2f}  math {m[2]:.2f}  stores {m[3]:.2f}  gap {m[4]:.2f}  us")
