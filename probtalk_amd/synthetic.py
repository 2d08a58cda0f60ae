"""Synthetic-weight model builders shared by bench.py, __graft_entry__.smoke() and the tests.

No reference checkpoint exists offline, so the wrappers are built exactly as a user would build
them (`init_model(name, args, config)`, VQ loaded from `config.Model.vq_path`) on weights from
probtalk_amd.weights; the "checkpoint" is written to a temporary file first.
"""
import os
import tempfile
import types

import torch

from .config import default_transformer_config
from .weights import fill_module


def args(infer=True, gpu="cuda"):
    return types.SimpleNamespace(gpu=gpu, infer=infer)


def write_vq_checkpoint(path, seed=0):
    """A `ckpt-*.pth`-shaped file ({'generator': {'VQ': state_dict}}, Trainer.py:201-216) with synthetic weights."""
    from .nets.modules import VQVAE_SC
    vq = fill_module(VQVAE_SC(376, 512, 128, 1024, 2, 512, q_type="pro", groups=4), seed, "VQ.")
    torch.save({"generator": {"VQ": vq.state_dict()}, "epoch": 99, "global_steps": 0}, path)
    return path


def build_pipeline(seed=0, tmpdir=None):
    """(predictor wrapper, refiner wrapper) = the two body models All_In_One_Model chains (ai1.py:36-42)."""
    from . import nets
    own = tmpdir is None
    tmpdir = tmpdir or tempfile.mkdtemp(prefix="probtalk_amd_")
    ckpt = write_vq_checkpoint(os.path.join(tmpdir, f"vq_seed{seed}_{os.getpid()}.pth"), seed)
    cfg = default_transformer_config(vq_path=ckpt)
    a = args()
    pred = nets.init_model("s2g_body_predictor", a, cfg)
    fill_module(pred.AudEnc, seed, "AudEnc.")
    fill_module(pred.Predictor, seed, "Predictor.")
    ref = nets.init_model("s2g_body_refiner", a, cfg)
    fill_module(ref.RTransformer, seed, "HFTransformer.")
    if own:
        try:
            os.remove(ckpt)
            os.rmdir(tmpdir)
        except OSError:
            pass
    return pred, ref


def holistic_step(pred, ref, aud, gt_poses, mask, ids, noise=None, sync=False, seed=None, streams=None):
    """One pass of the full path over one batch (test_holisticbody.py:248-253 via ai1.py:51-57):
    predictor.infer_on_batch -> refiner.infer_on_batch(pred_poses=...).  Returns (B,376,T) poses.

    `streams` (a list of torch.cuda.Stream): the clips are split into len(streams) contiguous shares, each enqueued on
    its own stream, and the caller's stream waits for all of them.  Clips are independent, so the result is the
    concatenation of the shares' results; what it buys is occupancy: every launch of the path ends in a partly filled
    last round of tiles (474 row tiles on 256 CUs) and alternates MFMA-bound main loops with HBM-bound epilogues, and
    workgroups of the other stream's kernels fill those gaps (2 x 256 clips: 74.7 against 81.6 ms per 512-clip step)."""
    B = aud.shape[0]
    if streams and len(streams) > 1 and B >= len(streams):
        if torch.is_tensor(seed):
            raise ValueError("holistic_step: a device-resident seed (graph capture) cannot be combined with streams")
        cur = torch.cuda.current_stream()
        n = len(streams)
        bounds = [(B * i) // n for i in range(n + 1)]
        outs = []
        if seed is None and noise is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        for i, s in enumerate(streams):
            lo, hi = bounds[i], bounds[i + 1]
            part = lambda t: t if t.shape[0] == 1 else t[lo:hi]
            nz = None if noise is None else tuple(t[:, lo:hi] for t in noise)
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                # each share draws from its own counter stream: the sampler indexes its noise by the clip's position
                # inside the call
                o = holistic_step(pred, ref, part(aud), part(gt_poses), part(mask), ids if ids.numel() == 1 else ids[lo:hi],
                                  noise=nz, sync=False, seed=None if seed is None else (seed + 0x9E3779B97F4A7C15 * i) % (1 << 63))
            outs.append(o)
        for s, o in zip(streams, outs):
            cur.wait_stream(s)
            o.record_stream(cur)
        out = torch.cat(outs)
        if sync:
            torch.cuda.synchronize()
        return out
    body, _ = pred.infer_on_batch(aud=aud, text=None, id=ids, B=B, gt_poses=gt_poses, mask=mask, noise=noise,
                                  sync=sync, seed=seed)
    out, _ = ref.infer_on_batch(aud=aud, B=B, id=ids, gt_poses=gt_poses, mask=mask, pred_poses=body, sync=sync)
    return out
