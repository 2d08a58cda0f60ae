"""GPU: the bench's own shape in the bench's own arithmetic.  512 clips x 300 frames per step, bf16 mode, the clips
split over two HIP streams exactly as bench.py runs them (256 clips per stream: the persistent multi-tile schedules,
160-row tiles, conv8 at 74-300 tiles).  Sampled codes of a bf16 run cannot be compared (sampling amplifies rounding), so
the step is pinned on the two things that can:
  (a) the MaskGIT pass-0 logits (every token is the mask id in both runs) after the bf16 AudEnc + motion-context encoder,
  (b) the pipeline tail on GIVEN codes: codes -> PQ gather + bf16 conv decoder -> resize / blend -> bf16 refiner -> blend,
both for clips {0, 255, 256, 511} (first / last of each stream's share) against the fp32 CPU oracle run on those four
clips alone, under the one criterion of tests/helpers.py::check_close: l2-relative error < 1e-2, worst element held to
1.3 x the CPU oracle's own bf16-operand floor."""
import pytest
import torch
import torch.nn.functional as F

from oracle import probtalk_oracle as O
from oracle.bf16_sim import bf16_operands
from probtalk_amd.nets.transformer_modules import interp_blend
from probtalk_amd.synthetic import build_pipeline
from probtalk_amd.weights import make_input
from tests.helpers import check_close, ref_state_dict

pytestmark = pytest.mark.gpu
CLIPS = [0, 255, 256, 511]


def test_bench_shape_two_streams_bf16_matches_oracle_on_sampled_clips():
    B, T, Tq, G = 512, 300, 37, 4
    N = Tq * G
    dev = "cuda"
    pred, ref = build_pipeline(seed=0)
    pred._eval()
    ref.RTransformer.eval()
    for mod in (pred.Predictor, pred.AudEnc, pred.VQ, ref.RTransformer):
        mod.set_precision("bf16")
    g = torch.Generator(device=dev).manual_seed(77)
    aud = torch.randn(B, 768, T, device=dev, generator=g)
    gt = torch.randn(B, 376, T, device=dev, generator=g) * 0.3
    mask = torch.zeros(B, 1, T, device=dev)
    mask[:, :, :30] = 1
    ids = (torch.arange(B, device=dev) * 5 + 2) % 4
    codes_in = torch.randint(0, 128, (B, Tq, G), device=dev, generator=g, dtype=torch.int32)
    codes_in[CLIPS] = make_input("full16.codes", (len(CLIPS), Tq, G), "int", low=0, high=128).to(dev, torch.int32)

    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    cur = torch.cuda.current_stream()
    logits0, outs = [], []
    for i, s in enumerate(streams):
        lo, hi = 256 * i, 256 * (i + 1)
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            # (a) predictor up to the sampler, as predictor.TrainWrapper._infer enqueues it
            input_poses = interp_blend(None, T, gt[lo:hi], mask[lo:hi], premask=False)
            audio = pred.AudEnc(aud[lo:hi])
            cond = pred.Predictor.condition(input_poses, mask[lo:hi], audio)
            _, lg = pred.Predictor.decoder.sample_good(ids[lo:hi], cond, None, T=8, return_logits=True, seed=1234 + i)
            logits0.append(lg[0].clone())
            del lg
            # (b) the tail on given codes
            dec = pred.VQ.decode_codes(codes_in[lo:hi])
            body = interp_blend(dec, T, input_poses, mask[lo:hi], premask=False)
            out, _ = ref.infer_on_batch(aud=aud[lo:hi], B=hi - lo, id=ids[lo:hi], gt_poses=gt[lo:hi], mask=mask[lo:hi],
                                        pred_poses=body, sync=False)
            outs.append(out)
    for s in streams:
        cur.wait_stream(s)
    torch.cuda.synchronize()
    logits0 = torch.cat(logits0)[CLIPS].cpu()
    out = torch.cat(outs)[CLIPS].cpu()
    assert bool(torch.isfinite(logits0).all()) and bool(torch.isfinite(out).all())

    sd_vq, sd_a, sd_p, sd_r = (ref_state_dict(k) for k in ("VQ", "AudEnc", "Predictor", "HFTransformer"))
    a_s, g_s, m_s, i_s = aud[CLIPS].cpu(), gt[CLIPS].cpu(), mask[CLIPS].cpu(), ids[CLIPS].cpu()
    c_s = codes_in[CLIPS].cpu().long()
    allmask = torch.full((len(CLIPS), Tq, G), 128, dtype=torch.int32)

    def oracle_logits0():
        cond = O.predictor_condition(sd_p, g_s * m_s, m_s, O.condition_encoder(sd_a, a_s))
        return O.tokens_to_logits(sd_p, allmask, i_s, cond).reshape(len(CLIPS), N, 128)

    def oracle_tail():
        e = O.pq_quantize_all(sd_vq, "vq_layer.", c_s).transpose(1, 2)
        body = F.interpolate(O.vqvae_decode(sd_vq, e), size=T, mode="linear")
        body = g_s * m_s * m_s + body * (1 - m_s)
        return O.refiner_infer(sd_r, a_s, g_s, m_s, i_s, body)

    with torch.no_grad():
        ref_l, ref_o = oracle_logits0(), oracle_tail()
        with bf16_operands():
            fl_l, fl_o = oracle_logits0(), oracle_tail()
    check_close(logits0, ref_l, 1e-2, "bf16 512 x 300, two streams: MaskGIT pass-0 logits", operand_floor=fl_l)
    check_close(out, ref_o, 1e-2, "bf16 512 x 300, two streams: pipeline tail (given codes -> decode -> refine)", operand_floor=fl_o)
