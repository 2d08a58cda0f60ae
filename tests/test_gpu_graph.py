"""HIP-graph replay of the full step (probtalk_amd/graphs.py): equal to the eager step with the same seed, new seed ->
new sample, inputs are read from the static buffers at replay time."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_graph_replay_matches_eager_and_follows_seed_and_inputs():
    from probtalk_amd.graphs import GraphedHolisticStep
    from probtalk_amd.synthetic import build_pipeline, holistic_step
    pred, ref = build_pipeline(seed=0)
    B, T = 3, 120
    g = torch.Generator(device="cuda").manual_seed(7)
    aud = torch.randn(B, 768, T, device="cuda", generator=g)
    gt = 0.3 * torch.randn(B, 376, T, device="cuda", generator=g)
    mask = torch.zeros(B, 1, T, device="cuda")
    mask[:, :, :20] = 1
    ids = torch.tensor([0, 2, 3], device="cuda")
    step = GraphedHolisticStep(pred, ref, B, T)
    seed_t = torch.tensor([1234], dtype=torch.int64, device="cuda")
    eager_dev = holistic_step(pred, ref, aud, gt, mask, ids, seed=seed_t).clone()
    # the device-seed entry point draws what the host-seed one draws
    pose_host, codes_host, _ = pred.infer_to_code(aud=aud, text=None, id=ids, B=B, gt_poses=gt, mask=mask, seed=1234)
    pose_dev, codes_dev, _ = pred.infer_to_code(aud=aud, text=None, id=ids, B=B, gt_poses=gt, mask=mask, seed=seed_t)
    assert torch.equal(codes_host, codes_dev) and torch.equal(pose_host, pose_dev)
    out1 = step(aud, gt, mask, ids, seed=1234).clone()
    assert torch.equal(out1, eager_dev), "graph replay differs from the eager step"
    out2 = step(aud, gt, mask, ids, seed=99).clone()
    assert not torch.equal(out1, out2), "a new seed must give a new sample"
    assert torch.equal(step(aud, gt, mask, ids, seed=1234), out1), "same seed, same sample"
    # inputs are taken at replay time
    aud2 = torch.randn(B, 768, T, device="cuda", generator=g)
    out3 = step(aud2, gt, mask, ids, seed=1234).clone()
    assert torch.equal(out3, holistic_step(pred, ref, aud2, gt, mask, ids, seed=seed_t))
    assert bool(torch.isfinite(out3).all())


def test_stream_split_step_equals_single_stream_step():
    """holistic_step(streams=[s0, s1]): clips are independent, so with the sampler's noise injected the split step must
    reproduce the one-stream step bit for bit (per-stream workspaces, joins on the caller's stream)."""
    from probtalk_amd.synthetic import build_pipeline, holistic_step
    from probtalk_amd.weights import make_input
    pred, ref = build_pipeline(seed=0)
    B, T = 5, 88
    g = torch.Generator(device="cuda").manual_seed(3)
    aud = torch.randn(B, 768, T, device="cuda", generator=g)
    gt = 0.3 * torch.randn(B, 376, T, device="cuda", generator=g)
    mask = torch.zeros(B, 1, T, device="cuda")
    mask[:, :, 10:30] = 1
    ids = torch.tensor([0, 1, 2, 3, 1], device="cuda")
    N = (T // 8) * 4
    nq = make_input("split.q", (8, B, N, 128), "exp").cuda()
    ng = make_input("split.g", (8, B, N), "gumbel").cuda()
    one = holistic_step(pred, ref, aud, gt, mask, ids, noise=(nq, ng)).clone()
    streams = [torch.cuda.Stream() for _ in range(2)]
    for _ in range(3):                                    # repeated: a workspace shared between streams would show up as noise
        two = holistic_step(pred, ref, aud, gt, mask, ids, noise=(nq, ng), streams=streams)
        torch.cuda.synchronize()
        assert torch.equal(one, two)
    three = holistic_step(pred, ref, aud, gt, mask, ids, noise=(nq, ng), streams=[torch.cuda.Stream() for _ in range(3)])
    assert torch.equal(one, three)
    # seeded mode: shares draw from different counter streams, the call is repeatable
    a = holistic_step(pred, ref, aud, gt, mask, ids, seed=5, streams=streams).clone()
    b = holistic_step(pred, ref, aud, gt, mask, ids, seed=5, streams=streams)
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())
