"""Sliding-window drivers either side of the hot path: long audio is generated in 180-frame slices whose
first 30 frames are the previous slice's output fed back as masked motion context.

Two callers exist in the reference and both are mirrored here, slice arithmetic included:

* `holistic_generate`  - scripts/test_holisticbody.py:239-262: every slice goes predictor -> refiner through
  `All_In_One_Model(forward_type='infer_on_batch')`; slices are [0,180), [150,330), then [150+180(i-1), 150+180i).
* `demo_generate`      - scripts/demo.py:113-178: predictor `infer_to_code` over 180-frame slices at stride 150, the
  code grids concatenated (3 overlapping latent steps dropped) and decoded ONCE, then the refiner over 176-frame slices.

The arithmetic of a slice is the HIP path (wrappers' infer_on_batch / infer_to_code, `interp_blend`); what this file adds
is only the windowing: slicing, the feedback copy into the context buffers and the concatenation.
"""
import math

import torch

from .nets.transformer_modules import interp_blend

SLICE = 180          # frames per predictor slice (test_holisticbody.py:240-246, demo.py:116)
COVER = 30           # frames of a slice that repeat the previous slice's tail (demo.py:117)


def holistic_slice_bounds(n_frames):
    """(start, end) of every slice as scripts/test_holisticbody.py:240-246 computes them.  From the third slice on the
    reference's stride is 180 while it still drops 30 frames per slice, so only clips of at most 330 frames stitch back
    to their own length there (the final blend at :262 raises otherwise); the bounds are reproduced as they are."""
    num_slices = 1 + math.ceil((n_frames - SLICE) / (SLICE - COVER))
    out = []
    for i in range(max(num_slices, 1)):
        out.append((0, SLICE) if i == 0 else (150 + SLICE * (i - 1), 150 + SLICE * i))
    return out


def demo_slice_bounds(n_frames, l_pslice=SLICE, cover=COVER):
    """scripts/demo.py:118,133-135 (predictor, l_pslice 180) and :159-163 (refiner, l_pslice 176)."""
    num_slices = math.ceil((n_frames - cover) / (l_pslice - cover))
    return [((l_pslice - cover) * i, (l_pslice - cover) * i + l_pslice) for i in range(num_slices)]


def _stitch(step, bounds, input_gt, input_mask, cover=COVER):
    """The feedback loop shared by both callers (test_holisticbody.py:244-260, demo.py:133-149,161-178): run `step` on
    each window of the context buffers, append its output minus the first `cover` frames, write it back as context."""
    pred, seconds, extras = None, 0.0, []
    for i, (s, e) in enumerate(bounds):
        out = step(i, s, e, input_gt[..., s:e], input_mask[..., s:e])
        pred_0, t = out[0], out[-1]
        extras.append(out[1:-1])
        pred = pred_0 if pred is None else torch.cat([pred, pred_0[..., cover:]], -1)
        input_gt[..., s:e] = pred_0
        input_mask[..., s:e] = 1
        seconds += t
    return pred, extras, seconds


def holistic_generate(generator, aud, text, gt_poses, mask, id, B, noise=None):
    """scripts/test_holisticbody.py:239-262.  `generator` is an All_In_One_Model with body=[predictor, refiner];
    aud (1,768,L), gt_poses (1,C,L), mask (B,1,L); `noise[i]` optionally injects slice i's MaskGIT noise pair
    (parity runs).  Returns (pred (B,C,L), seconds)."""
    input_gt = gt_poses.clone().repeat(B, 1, 1) if gt_poses.shape[0] != B else gt_poses.clone()
    input_mask = mask.clone()

    def step(i, s, e, gt_win, mask_win):
        kw = {} if noise is None else {"noise": noise[i]}
        return generator(forward_type='infer_on_batch', aud=aud[..., s:e], text=None if text is None else text[..., s:e],
                         gt_poses=gt_win, mask=mask_win, id=id, B=B, **kw)

    pred, _, seconds = _stitch(step, holistic_slice_bounds(aud.shape[-1]), input_gt, input_mask)
    if pred.shape[-1] != gt_poses.shape[-1]:
        raise RuntimeError(f"slices stitch to {pred.shape[-1]} frames for a clip of {gt_poses.shape[-1]} "
                           "(the reference's slice stride only closes for clips of at most 330 frames)")
    # pred = gt_poses * mask + pred * (1 - mask)   (test_holisticbody.py:262)
    return interp_blend(pred, pred.shape[-1], gt_poses.expand(B, -1, -1), mask.expand(B, -1, -1), premask=False), seconds


def demo_generate(generator, aud, id, num_sample, noise=None):
    """scripts/demo.py:113-178.  aud (1,768,t) -> (refined (B,C,t), preliminary_motion (B,C,t), codes, seconds).
    The reference joins exactly two code grids (`cat([code_list[0], code_list[1][:, 3:]])`, :151); every further
    slice is joined the same way here (identical for clips of up to 330 frames)."""
    predictor, refiner = generator.body_model[0], generator.body_model[1]
    t = aud.shape[-1]
    dev = aud.device
    C = predictor.full_dim if hasattr(predictor, "full_dim") else 376
    input_gt = torch.zeros([num_sample, C, t], device=dev)
    input_mask = torch.zeros([num_sample, 1, t], device=dev)

    def p_step(i, s, e, gt_win, mask_win):
        kw = {} if noise is None else {"noise": noise[i]}
        return predictor.infer_to_code(aud=aud[..., s:e], text=None, gt_poses=gt_win, mask=mask_win, id=id, B=num_sample, **kw)

    _, extras, sec_p = _stitch(p_step, demo_slice_bounds(t, SLICE), input_gt, input_mask)
    code_list = [x[0] for x in extras]
    codes = torch.cat([code_list[0]] + [c[:, 3:] for c in code_list[1:]], dim=1)
    preliminary = predictor.VQ.decode_codes(codes)                           # quantize_all + transpose + decode (:152-153)
    zeros_gt = torch.zeros_like(input_gt)
    zeros_mk = torch.zeros_like(input_mask)
    preliminary = interp_blend(preliminary, t, zeros_gt, zeros_mk, premask=False)     # F.interpolate(..., size=t) (:155)

    input_mask = torch.zeros([num_sample, 1, t], device=dev)                 # the context POSES stay (:157 resets the mask only)

    def r_step(i, s, e, gt_win, mask_win):
        return refiner.infer_on_batch(aud=aud[..., s:e], pred_poses=preliminary[..., s:e], gt_poses=gt_win, mask=mask_win,
                                      id=id, B=num_sample)

    pred_all, _, sec_r = _stitch(r_step, demo_slice_bounds(t, 176), input_gt, input_mask)
    return pred_all, preliminary, codes, sec_p + sec_r
