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


def _torch_adamw_state(params, lr, betas, steps=0, seed=0):
    """A torch.optim.AdamW.state_dict()-shaped entry ('generator_optim' in the reference's checkpoints, predictor.py:136):
    per-parameter step / exp_avg / exp_avg_sq keyed by position, one param_group."""
    g = torch.Generator().manual_seed(seed + 77)
    state = {}
    if steps > 0:
        for i, p in enumerate(params):
            state[i] = {"step": torch.tensor(float(steps)), "exp_avg": 1e-3 * torch.randn(p.shape, generator=g),
                        "exp_avg_sq": 1e-6 * torch.rand(p.shape, generator=g)}
    return {"state": state, "param_groups": [{"lr": lr, "betas": tuple(betas), "eps": 1e-8, "weight_decay": 1e-2,
                                              "amsgrad": False, "params": list(range(len(params)))}]}


def write_reference_checkpoints(outdir, seed=0, optim_steps=3):
    """Files laid out as the reference's trainer writes them (trainer/Trainer.py:201-216: {'generator':
    wrapper.state_dict(), 'epoch', 'global_steps'} in ckpt-<epoch>.pth) for the three wrappers of the path, on generator
    weights: the per-module dicts carry the `module.` prefix nn.DataParallel gives them (predictor.py:100-105,128-143;
    refiner.py's likewise), the positional tables are float64 (PTransformer.py:227-245), and 'generator_optim' is a
    torch.optim.AdamW state_dict.  Returns {'vq': path, 'predictor': path, 'refiner': path}."""
    from .nets.modules import VQVAE_SC, ConditionEncoder
    from .nets.transformer_modules import PredictorNet, RTransformer
    os.makedirs(outdir, exist_ok=True)
    par = lambda sd: {"module." + k: v.clone() for k, v in sd.items()}
    vq = fill_module(VQVAE_SC(376, 512, 128, 1024, 2, 512, q_type="pro", groups=4), seed, "VQ.")
    ae = fill_module(ConditionEncoder(768, 256, 2, 256), seed, "AudEnc.")
    te = ConditionEncoder(768, 256, 2, 256)
    pn = fill_module(PredictorNet(False, 376, 128, 512, 10, 6, 4, groups=4, identity=True, maskgit=True, maskgit_T=8,
                                  transformer=True, text=None, audio="wav2vec", motion_context=True), seed, "Predictor.")
    rt = fill_module(RTransformer(376, 768, 512, 6, use_label=True, motion_context=True), seed, "HFTransformer.")
    assert pn.state_dict()["decoder.PE.position_encoding.weight"].dtype == torch.float64
    paths = {k: os.path.join(outdir, k, "ckpt-99.pth") for k in ("vq", "predictor", "refiner")}
    for p in paths.values():
        os.makedirs(os.path.dirname(p), exist_ok=True)
    torch.save({"generator": {"VQ": vq.state_dict(),
                              "generator_optim": _torch_adamw_state(list(vq.parameters()), 1e-4, (0.9, 0.999), 0, seed)},
                "epoch": 99, "global_steps": 12345}, paths["vq"])
    p_params = list(ae.parameters()) + list(pn.parameters()) + list(te.parameters())
    torch.save({"generator": {"AudEnc": par(ae.state_dict()), "Predictor": par(pn.state_dict()), "TextEnc": par(te.state_dict()),
                              "generator_optim": _torch_adamw_state(p_params, 1e-4, (0.9, 0.99), optim_steps, seed)},
                "epoch": 99, "global_steps": 23456}, paths["predictor"])
    torch.save({"generator": {"HFTransformer": par(rt.state_dict()), "AudEnc": {},
                              "generator_optim": _torch_adamw_state(list(rt.parameters()), 1e-4, (0.9, 0.99), optim_steps, seed)},
                "epoch": 99, "global_steps": 34567}, paths["refiner"])
    return paths


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
