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


def holistic_step(pred, ref, aud, gt_poses, mask, ids, noise=None, sync=False, seed=None):
    """One pass of the full path over one batch (test_holisticbody.py:248-253 via ai1.py:51-57):
    predictor.infer_on_batch -> refiner.infer_on_batch(pred_poses=...).  Returns (B,376,T) poses."""
    B = aud.shape[0]
    body, _ = pred.infer_on_batch(aud=aud, text=None, id=ids, B=B, gt_poses=gt_poses, mask=mask, noise=noise,
                                  sync=sync, seed=seed)
    out, _ = ref.infer_on_batch(aud=aud, B=B, id=ids, gt_poses=gt_poses, mask=mask, pred_poses=body, sync=sync)
    return out
