"""Run the three conv stacks of the bf16 mode once at the bench's per-stream shape (256 clips x 300 frames) so that a
rocprofv3 --kernel-trace of this script lists every conv launch in order (which layers cost what)."""
import sys
import torch
sys.path.insert(0, ".")
from probtalk_amd.synthetic import build_pipeline

pred, ref = build_pipeline(seed=0)
for mod in (pred.Predictor, pred.AudEnc, pred.VQ):
    mod.set_precision("bf16")
    mod.eval()
B, T = 256, 300
g = torch.Generator(device="cuda").manual_seed(0)
aud = torch.randn(B, 768, T, device="cuda", generator=g)
mm = torch.randn(B, 376, T, device="cuda", generator=g)
mask = torch.zeros(B, 1, T, device="cuda")
codes = torch.randint(0, 128, (B, 37, 4), device="cuda", generator=g)
for it in range(2):
    audio = pred.AudEnc(aud)
    cond = pred.Predictor.condition(mm, mask, audio)
    out = pred.VQ.decode_codes(codes)
torch.cuda.synchronize()
print(audio.shape, cond.shape, out.shape)
