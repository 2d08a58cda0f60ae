"""Micro-benchmark of the bf16 attention core at the pipeline's shapes (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from probtalk_amd import _lib
L = _lib.lib()
for (B, Nq, Nk, name) in [(512, 148, 148, "predictor SA"), (512, 148, 37, "predictor CA"), (512, 300, 300, "refiner SA/CA")]:
    H = 8
    q = torch.randn(B * Nq, 512, device="cuda").bfloat16()
    kv = torch.randn(B * Nk, 1024, device="cuda").bfloat16()
    o = torch.empty(B * Nq, 512, device="cuda", dtype=torch.bfloat16)
    def run():
        _lib.check(L.pt_attention_bf16(_lib.ptr(q), 512, _lib.ptr(kv), 1024, _lib.ptr(kv[:, 512:]), 1024, _lib.ptr(o), 512, B, Nq, Nk, H, _lib.stream()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    qf, kf, vf = q.float().view(B, Nq, H, 64), kv[:, :512].float().view(B, Nk, H, 64), kv[:, 512:].float().view(B, Nk, H, 64)
    ref = torch.nn.functional.scaled_dot_product_attention(qf.transpose(1, 2)[:8], kf.transpose(1, 2)[:8], vf.transpose(1, 2)[:8]).transpose(1, 2).reshape(8 * Nq, 512)
    err = float((o[:8 * Nq].float() - ref).abs().max() / ref.abs().max())
    mb = (q.numel() + kv.numel() + o.numel()) * 2 / 1e6
    print(f"{name:14s} B={B} Nq={Nq} Nk={Nk}: {us:7.1f} us  {mb / us * 1e-3 * 1e3:6.2f} TB/s-equivalent x1e-3... {mb:.0f} MB  err {err:.1e}", flush=True)
