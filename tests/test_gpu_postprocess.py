"""GPU parity of the pose post-process kernel (pt_postprocess_poses through probtalk_amd.postprocess) against the
reference-generated fixture and the oracle: denormalise + 6-D -> axis-angle + part2full, tolerance 1e-5 of the output
range, including the small-angle branch and angles up to pi."""
import math

import pytest
import torch

from oracle import postprocess as OP
from probtalk_amd import postprocess as PP
from tests.helpers import rel_err
from tests.test_oracle_postprocess import load_gold

pytestmark = pytest.mark.gpu
TOL = 1e-5


def test_postprocess_matches_reference_fixture_incl_edge_cases():
    g = load_gold()
    out = PP.poses_to_axis_angle(g["denormalized"].cuda()).cpu()
    ref = g["out"]
    assert tuple(out.shape) == (3, 32, 265)
    e = rel_err(out, ref)
    print(f"post-process vs reference fixture: max |diff| / max |ref| = {e:.2e}")
    assert e < TOL
    # the planted rotations one by one (clip 2): angle error and axis error, by regime
    n_e = int(g["n_edge"])
    cols = list(range(3)) + list(range(18, 21)) + list(range(27, 30)) + list(range(36, 165))
    a, r = out[2][:, cols].reshape(-1, 3)[:n_e], ref[2][:, cols].reshape(-1, 3)[:n_e]
    ang = r.norm(dim=-1)
    small, near_pi = ang < 1e-6, ang > math.pi - 2e-2
    assert float((a - r)[small].abs().max()) < 1e-9              # small-angle branch: 2 q_xyz / (1 - angle^2/24)
    assert float((a - r)[~small & ~near_pi].abs().max()) < 1e-5
    # near pi the reference's own formula is ill-conditioned (sqrt of 1 + trace -> 0): the kernel follows the same
    # operation order without contraction, so it stays within the bound there too
    assert float((a - r)[near_pi].abs().max()) < TOL * math.pi
    # exact copies: expression coefficients and the fixed lower body
    assert torch.equal(out[:, :, 165:], g["denormalized"][:, 276:].transpose(1, 2))
    assert torch.equal(out[0, 3, 3:18], OP.LOWER_POSE[:15])


@pytest.mark.parametrize("B,T,full", [(1, 1, True), (2, 33, False), (5, 300, True)])
def test_postprocess_with_denormalisation_matches_oracle(B, T, full):
    g = load_gold()
    ns = g["norm_stats"]
    x = torch.randn(B, 376, T, generator=torch.Generator().manual_seed(B * 1000 + T))
    out = PP.poses_to_axis_angle(x.cuda(), ns, full=full).cpu()
    ref = OP.postprocess(x, ns, full=full)
    assert tuple(out.shape) == (B, T, 265 if full else 238)
    assert rel_err(out, ref) < TOL
    # statistics handed over as a CUDA tensor / with an explicit c_index give the same result
    out2 = PP.poses_to_axis_angle(x.cuda(), ns.cuda(), c_index=PP.c_index_6d, full=full).cpu()
    assert torch.equal(out, out2)


def test_postprocess_rejects_bad_arguments():
    with pytest.raises(RuntimeError):
        PP.poses_to_axis_angle(torch.zeros(1, 376, 8))                      # host tensor
    with pytest.raises(RuntimeError):
        PP.poses_to_axis_angle(torch.zeros(1, 238, 8, device="cuda"))       # not the 'fbhe' layout
    with pytest.raises(RuntimeError):
        PP.poses_to_axis_angle(torch.zeros(1, 376, 8, device="cuda"), torch.zeros(2, 100))


def test_postprocess_of_a_generated_batch_is_finite_and_a_rotation():
    """End of the pipeline: whatever the refiner returns becomes valid axis-angle (|angle| <= pi)."""
    x = 3 * torch.randn(4, 376, 64, device="cuda")
    out = PP.poses_to_axis_angle(x)
    assert bool(torch.isfinite(out).all())
    aa = out[:, :, 36:165].reshape(-1, 3)
    assert float(aa.norm(dim=-1).max()) <= math.pi + 1e-4
