"""GPU: the bench's own shape in the parity arithmetic.  512 clips x 300 frames per step, fp32, MaskGIT noise injected,
clips split over two HIP streams exactly as bench.py runs them; clips {0, 255, 256, 511} (first / last of each stream's
share) are compared with the CPU oracle run on those four clips alone: codes bit for bit, motion to 1e-4.  This is the
end-to-end check of the persistent multi-tile schedules (75,776 / 153,600-row launches, hundreds of tiles per CU round)
and of the stream split, which the small-batch parity tests do not reach."""
import pytest
import torch

from oracle import probtalk_oracle as O
from probtalk_amd.synthetic import build_pipeline, holistic_step
from probtalk_amd.weights import make_input
from tests.helpers import ref_state_dict, rel_err

pytestmark = pytest.mark.gpu
CLIPS = [0, 255, 256, 511]


def test_bench_shape_two_streams_fp32_matches_oracle_on_sampled_clips():
    B, T, Tq, G, PASSES = 512, 300, 37, 4, 8
    N = Tq * G
    dev = "cuda"
    pred, ref = build_pipeline(seed=0)
    g = torch.Generator(device=dev).manual_seed(99)
    aud = torch.randn(B, 768, T, device=dev, generator=g)
    gt = torch.randn(B, 376, T, device=dev, generator=g) * 0.3
    mask = torch.zeros(B, 1, T, device=dev)
    mask[:, :, :30] = 1                                         # 30 context frames, as the sliced demo path has them
    ids = (torch.arange(B, device=dev) * 7 + 1) % 4
    # noise: Exp(1) / Gumbel(0,1) for every clip on the GPU; the sampled clips' noise comes from the CPU so that the
    # oracle sees the very same numbers
    nq = torch.empty(PASSES, B, N, 128, device=dev).exponential_(generator=g)
    u = torch.rand(PASSES, B, N, device=dev, generator=g).clamp_(1e-7, 1 - 1e-7)
    ng = -torch.log(-torch.log(u))
    nq_s = make_input("full.q", (PASSES, len(CLIPS), N, 128), "exp")
    ng_s = make_input("full.g", (PASSES, len(CLIPS), N), "gumbel")
    nq[:, CLIPS] = nq_s.to(dev)
    ng[:, CLIPS] = ng_s.to(dev)
    # (a) the predictor at full size on one stream: codes and coarse motion of the sampled clips
    body, codes, _ = pred.infer_to_code(aud=aud, text=None, id=ids, B=B, gt_poses=gt, mask=mask, noise=(nq, ng))
    # (b) the whole step as bench.py runs it: two streams, 256 clips each
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    out = holistic_step(pred, ref, aud, gt, mask, ids, noise=(nq, ng), sync=True, streams=streams)
    assert tuple(out.shape) == (B, 376, T) and bool(torch.isfinite(out).all())
    # oracle on the four clips
    sd_vq, sd_a, sd_p, sd_r = (ref_state_dict(k) for k in ("VQ", "AudEnc", "Predictor", "HFTransformer"))
    a_s, g_s, m_s, i_s = aud[CLIPS].cpu(), gt[CLIPS].cpu(), mask[CLIPS].cpu(), ids[CLIPS].cpu()
    with torch.no_grad():
        body_ref, codes_ref = O.predictor_infer(sd_a, sd_p, sd_vq, a_s, g_s, m_s, i_s, nq_s, ng_s)
        out_ref = O.refiner_infer(sd_r, a_s, g_s, m_s, i_s, body_ref)
    same = [bool(torch.equal(codes[c].cpu(), codes_ref[k])) for k, c in enumerate(CLIPS)]
    e_body, e_out = rel_err(body[CLIPS].cpu(), body_ref), rel_err(out[CLIPS].cpu(), out_ref)
    print(f"512 x 300 fp32, two streams: codes identical on clips {CLIPS}: {same}; coarse motion rel err {e_body:.2e}, "
          f"refined motion rel err {e_out:.2e}")
    assert all(same), "MaskGIT codes of a sampled clip differ from the oracle"
    assert e_body < 1e-4 and e_out < 1e-4
    # the two-stream result is the one-stream result (clips are independent)
    out1 = holistic_step(pred, ref, aud, gt, mask, ids, noise=(nq, ng), sync=True)
    assert torch.equal(out1[CLIPS], out[CLIPS])
