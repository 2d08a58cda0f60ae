"""GPU: checkpoints in the reference's on-disk layout (trainer/Trainer.py:201-216) load through the reference's own entry
points -- All_In_One_Model (nets/ai1.py:10-16,36-42) and init_model(pretrained=True) (nets/init_model.py:6-12) -- and
give bit-for-bit the outputs of the same weights loaded module by module; a training run saved with state_dict() and
resumed with load_state_dict() (Trainer.py:80-86) continues exactly where it stopped."""
import types

import pytest
import torch

from probtalk_amd import nets, synthetic
from probtalk_amd.config import default_transformer_config
from probtalk_amd.weights import make_input

pytestmark = pytest.mark.gpu


def _inputs(B=2, T=88):
    aud = make_input("ck.aud", (B, 768, T)).cuda()
    gt = make_input("ck.gt", (1, 376, T)).cuda()
    mask = torch.zeros(1, 1, T, device="cuda")
    mask[:, :, :30] = 1
    return aud, gt, mask, torch.tensor([1, 3]).cuda()


def test_reference_layout_checkpoints_load_and_match_bit_for_bit(tmp_path):
    paths = synthetic.write_reference_checkpoints(str(tmp_path), seed=0)
    # the files hold what the reference's trainer writes
    ck = torch.load(paths["predictor"], map_location="cpu", weights_only=True)
    assert set(ck) == {"generator", "epoch", "global_steps"}
    assert set(ck["generator"]) == {"AudEnc", "Predictor", "TextEnc", "generator_optim"}
    assert all(k.startswith("module.") for k in ck["generator"]["Predictor"])
    assert ck["generator"]["Predictor"]["module.decoder.PE.position_encoding.weight"].dtype == torch.float64
    cfg = default_transformer_config(vq_path=paths["vq"])
    args = synthetic.args()
    aud, gt, mask, ids = _inputs()
    # (1) the demo's way: All_In_One_Model over [predictor, refiner] (scripts/demo.py / test_holisticbody.py:248-253)
    gen = nets.All_In_One_Model(None, None, ["s2g_body_predictor", "s2g_body_refiner"], [paths["predictor"], paths["refiner"]],
                                "cuda", args, cfg)
    out_a, _ = gen(forward_type="infer_on_batch", target_frame=88, aud=aud, text=None, id=ids, B=2, gt_poses=gt, mask=mask, seed=5)
    # (2) init_model(pretrained=True) per wrapper, chained by hand
    pred = nets.init_model("s2g_body_predictor", args, cfg, pretrained=True, model_path=paths["predictor"])
    ref = nets.init_model("s2g_body_refiner", args, cfg, pretrained=True, model_path=paths["refiner"])
    out_b = synthetic.holistic_step(pred, ref, aud, gt, mask, ids, seed=5)
    # (3) the same generator weights put into the modules directly (how bench.py and smoke() build the pipeline)
    pred_c, ref_c = synthetic.build_pipeline(seed=0)
    out_c = synthetic.holistic_step(pred_c, ref_c, aud, gt, mask, ids, seed=5)
    assert tuple(out_a.shape) == (2, 376, 88)
    assert torch.equal(out_a, out_b) and torch.equal(out_a, out_c)
    # the optimizer state came along (torch.optim.AdamW layout -> the library's AdamW)
    st = pred.generator_optimizer.state
    p0 = next(pred.AudEnc.parameters())
    assert st[p0]["step"] == 3 and st[p0]["exp_avg"].shape == p0.shape
    # the frozen tokenizer is read from config.Model.vq_path by both wrappers (predictor.py:84-85, refiner.py:88-90)
    vq_sd = torch.load(paths["vq"], map_location="cpu", weights_only=True)["generator"]["VQ"]
    assert torch.equal(pred.VQ.state_dict()["vq_layer.embeddings_2"], vq_sd["vq_layer.embeddings_2"])
    # a vq_teacher checkpoint loads into its own wrapper too
    vqt = nets.init_model("s2g_body_vqt", args, cfg, pretrained=True, model_path=paths["vq"])
    assert torch.equal(vqt.VQ.state_dict()["encoder.project.conv.weight"], vq_sd["encoder.project.conv.weight"])


def _bat(B, T_, seed):
    g = torch.Generator().manual_seed(seed)
    mask = torch.zeros(B, 1, T_)
    mask[1, :, 20:50] = 1
    return {"aud_feat": make_input(f"ck.aud{seed}", (B, 768, T_)), "poses": 0.3 * make_input(f"ck.poses{seed}", (B, 330, T_)),
            "expression": 0.3 * make_input(f"ck.expr{seed}", (B, 100, T_)), "speaker": torch.tensor([1, 3, 0, 2][:B]),
            "epoch": 10, "_mask": mask, "_keep_scores": torch.rand(B, (T_ // 8) * 4, generator=g),
            "_noise": torch.randn(B, 376, T_, generator=g)}


@pytest.mark.parametrize("name", ["s2g_body_predictor", "s2g_body_refiner"])
def test_resume_continues_the_training_run(tmp_path, name):
    """Trainer.save_model -> Trainer.resume: three steps in one run == two steps, save, load into a fresh wrapper, one
    step.  Dropout is ON: the seed stream's position is part of the checkpoint, so the resumed step draws the masks the
    uninterrupted run drew."""
    paths = synthetic.write_reference_checkpoints(str(tmp_path), seed=1)
    cfg = default_transformer_config(vq_path=paths["vq"])
    key = "predictor" if "predictor" in name else "refiner"
    args = types.SimpleNamespace(gpu="cuda", infer=False)
    a = nets.init_model(name, args, cfg, pretrained=True, model_path=paths[key])
    bats = [_bat(3, 88, s) for s in range(3)]
    for b in bats[:2]:
        a(b)
    save = str(tmp_path / "ckpt-0.pth")
    torch.save({"generator": a.state_dict(), "epoch": 0, "global_steps": a.global_step}, save)        # Trainer.py:201-216
    _, ld_a = a(bats[2])
    ck = torch.load(save, map_location="cpu", weights_only=True)
    b = nets.init_model(name, args, cfg)
    b.load_state_dict(ck["generator"])                                                                  # Trainer.py:80-86
    b.global_step = ck["global_steps"]
    _, ld_b = b(bats[2])
    assert b.global_step == a.global_step == 3
    assert abs(ld_a["loss"] - ld_b["loss"]) <= 1e-6 * abs(ld_a["loss"])
    mods = [("AudEnc", "Predictor")] if key == "predictor" else [("RTransformer",)]
    for m in mods[0]:
        pa, pb = dict(getattr(a, m).named_parameters()), dict(getattr(b, m).named_parameters())
        for k in pa:
            # same inputs, same masks, same optimizer state; weight gradients are summed with atomics, so the last
            # bits of a step may differ: compare to 1e-3 of one AdamW step (lr 1e-4)
            assert float((pa[k] - pb[k]).abs().max()) < 1e-7 + 1e-3 * 1e-4, (m, k)
