"""Sliding-window drivers: slice arithmetic on CPU, stitched output against the oracle on the GPU."""
import pytest
import torch

from probtalk_amd import slicing


def test_holistic_slice_bounds_follow_the_reference_formula():
    # scripts/test_holisticbody.py:240-246 evaluated by hand
    assert slicing.holistic_slice_bounds(180) == [(0, 180)]
    assert slicing.holistic_slice_bounds(300) == [(0, 180), (150, 330)]
    assert slicing.holistic_slice_bounds(330) == [(0, 180), (150, 330)]
    assert slicing.holistic_slice_bounds(331) == [(0, 180), (150, 330), (330, 510)]
    assert slicing.holistic_slice_bounds(88) == [(0, 180)]           # shorter than one slice: 1 + ceil(-92/150) = 1


def test_demo_slice_bounds_follow_the_reference_formula():
    # scripts/demo.py:118,134-135 (l_pslice 180) and :159-163 (l_pslice 176)
    assert slicing.demo_slice_bounds(300) == [(0, 180), (150, 330)]
    assert slicing.demo_slice_bounds(330) == [(0, 180), (150, 330)]
    assert slicing.demo_slice_bounds(331) == [(0, 180), (150, 330), (300, 480)]
    assert slicing.demo_slice_bounds(300, 176) == [(0, 176), (146, 322)]
    assert slicing.demo_slice_bounds(323, 176) == [(0, 176), (146, 322), (292, 468)]


def test_stitch_feeds_each_output_back_as_context():
    L, B, C = 330, 2, 3
    gt, mk = torch.zeros(B, C, L), torch.zeros(B, 1, L)
    seen = []

    def step(i, s, e, g, m):
        seen.append((i, s, e, g.clone(), m.clone()))
        return torch.full((B, C, e - s), float(i + 1)), 0.5

    pred, _, sec = slicing._stitch(step, slicing.holistic_slice_bounds(L), gt, mk)
    assert pred.shape == (B, C, L) and sec == 1.0
    assert bool((pred[..., :180] == 1).all()) and bool((pred[..., 180:] == 2).all())
    # the second slice saw the first one's last 30 frames as masked context
    _, s, e, g, m = seen[1]
    assert (s, e) == (150, 330)
    assert bool((g[..., :30] == 1).all()) and bool((g[..., 30:] == 0).all())
    assert bool((m[..., :30] == 1).all()) and bool((m[..., 30:] == 0).all())


def _pipeline():
    from probtalk_amd.nets.ai1 import All_In_One_Model
    from probtalk_amd.synthetic import build_pipeline
    pred, ref = build_pipeline(seed=0)
    gen = object.__new__(All_In_One_Model)
    gen.face_model, gen.body_model = None, [pred, ref]
    sds = (pred.AudEnc.state_dict(), pred.Predictor.state_dict(), pred.VQ.state_dict(), ref.RTransformer.state_dict())
    return gen, sds


def _noises(tag, B, lengths):
    from probtalk_amd.weights import make_input
    out = []
    for i, n_frames in enumerate(lengths):
        N = 4 * (n_frames // 8)
        out.append((make_input(f"{tag}.q{i}", (8, B, N, 128), "exp"), make_input(f"{tag}.g{i}", (8, B, N), "gumbel")))
    return out


@pytest.mark.gpu
def test_holistic_generate_matches_oracle():
    from oracle import probtalk_oracle as O
    from probtalk_amd.weights import make_input
    from tests.helpers import rel_err
    gen, (sd_a, sd_p, sd_vq, sd_r) = _pipeline()
    B, L = 2, 300                                     # two slices: 180 frames, then 150 with 30 frames of context
    aud = make_input("slice.aud", (1, 768, L))
    gt = make_input("slice.gt", (1, 376, L))
    mask = torch.zeros(B, 1, L)
    mask[:, :, :12] = 1                               # a little ground-truth context at the start of the clip
    ids = torch.tensor([2])
    noises = _noises("slice.n", B, [180, 150])
    out, _ = slicing.holistic_generate(gen, aud.cuda(), None, gt.cuda(), mask.cuda(), ids.cuda(), B,
                                       noise=[(q.cuda(), g.cuda()) for q, g in noises])
    ref = O.holistic_slices_infer(sd_a, sd_p, sd_vq, sd_r, aud.repeat(B, 1, 1), gt.repeat(B, 1, 1), mask, ids.repeat(B), noises)
    assert out.shape == (B, 376, L)
    assert rel_err(out.cpu(), ref) < 1e-4


@pytest.mark.gpu
def test_demo_generate_matches_oracle():
    from oracle import probtalk_oracle as O
    from probtalk_amd.weights import make_input
    from tests.helpers import rel_err
    gen, (sd_a, sd_p, sd_vq, sd_r) = _pipeline()
    B, t = 2, 330
    aud = make_input("demo.aud", (1, 768, t))
    ids = torch.tensor([1])
    noises = _noises("demo.n", B, [180, 180])
    out, prelim, codes, _ = slicing.demo_generate(gen, aud.cuda(), ids.cuda(), B, noise=[(q.cuda(), g.cuda()) for q, g in noises])
    ref, prelim_o, codes_o = O.demo_slices_infer(sd_a, sd_p, sd_vq, sd_r, aud.repeat(B, 1, 1), ids.repeat(B), noises)
    assert codes.shape == (B, 22 + 19, 4) and torch.equal(codes.cpu(), codes_o)
    assert rel_err(prelim.cpu(), prelim_o) < 1e-4
    assert out.shape == ref.shape == (B, 376, 146 + 146 + 38)
    assert rel_err(out.cpu(), ref) < 1e-4
