"""Experiment: the 512-clip step as two 256-clip halves on two HIP streams (two model instances, so no shared workspace):
does the hardware fill one kernel's last-round idle CUs with the other stream's workgroups?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from probtalk_amd.synthetic import build_pipeline, holistic_step

B, T, NS = int(os.environ.get("B", 512)), 300, int(os.environ.get("NS", 2))
pipes = [build_pipeline(seed=0) for _ in range(NS)]
for pred, ref in pipes:
    for mod in (pred.Predictor, pred.AudEnc, pred.VQ, ref.RTransformer):
        mod.set_precision("bf16")
g = torch.Generator(device="cuda").manual_seed(1)
aud = torch.randn(B, 768, T, device="cuda", generator=g)
gt = torch.zeros(B, 376, T, device="cuda")
mask = torch.zeros(B, 1, T, device="cuda")
ids = torch.randint(0, 4, (B,), device="cuda", generator=g)
streams = [torch.cuda.Stream() for _ in range(NS)]
h = B // NS


def step():
    outs = []
    for i, ((pred, ref), s) in enumerate(zip(pipes, streams)):
        sl = slice(i * h, (i + 1) * h)
        with torch.cuda.stream(s):
            outs.append(holistic_step(pred, ref, aud[sl], gt[sl], mask[sl], ids[sl], sync=False))
    return outs


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for _ in range(K):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"{NS} streams x {h} clips: {dt * 1e3:.2f} ms/step, {B * T / dt / 1e6:.3f} M frames/s")
