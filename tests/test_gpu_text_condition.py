"""GPU: the text-conditioned predictor (Model.p_text set; off in config/transformer.json but part of the reference's
predictor: predictor.py:69-76,248-254,415-431, predictornet.py:51,93-97).  The condition grows to 1024 channels (motion 512
| audio 256 | text 256); inference and training go through the same kernels.  Pinned by a fixture from the imported
reference (tools/make_golden_text.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import probtalk_oracle as O
from probtalk_amd.weights import generate_state_dict, make_input
from tests.helpers import GOLDEN, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _nets():
    from probtalk_amd.nets.modules import ConditionEncoder
    from probtalk_amd.nets.transformer_modules import PredictorNet
    te = ConditionEncoder(768, 256, 2, 256).eval()
    te.load_state_dict(generate_state_dict(te.state_dict(), 0, "TextEnc."))
    pn = PredictorNet(False, 376, 128, 512, 10, 6, 4, groups=4, identity=True, maskgit=True, maskgit_T=8, transformer=True,
                      text="bert", audio="wav2vec", motion_context=True).eval()
    pn.load_state_dict(generate_state_dict(pn.state_dict(), 0, "PredictorText."))
    return te, pn


def test_text_conditioned_predictor_matches_reference_fixture():
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "predictor_text_b2_t88.npz")).items()}
    te, pn = _nets()
    assert pn.decoder.condi_dim == 1024
    gt = make_input("pred.gt", (2, 376, 88))
    mask = torch.zeros(2, 1, 88)
    mask[:, :, :30] = 1
    audio = make_input("ptext.audio", (2, 256, 11))
    text = te(make_input("ptext.text", (2, 768, 88)).cuda())
    assert rel_err(text.cpu(), g["text"]) < TOL
    cond = pn.condition((gt * mask).cuda(), mask.cuda(), audio.cuda(), text)
    assert tuple(cond.shape) == (2, 1024, 11)
    assert rel_err(cond.cpu(), g["condition"]) < TOL
    ids = make_input("pred.ids", (2, 11, 4), "int", low=0, high=129)
    label = torch.tensor([1, 3])
    logits = pn.decoder.tokens_to_logits(ids.cuda(), label.cuda(), g["condition"].cuda())
    assert rel_err(logits.cpu(), g["logits"]) < TOL
    # the oracle restatement is generic in the feature count
    sd = {k: v.detach().cpu() for k, v in pn.state_dict().items()}
    cond_o = O.predictor_condition(sd, gt * mask, mask, torch.cat([audio, g["text"]], 1))
    assert rel_err(cond_o, g["condition"]) < 1e-5
    # full sampling call with text (seeded sampler) and the bf16 mode on the wider condition
    codes = pn((gt * mask).cuda(), None, mask.cuda(), label.cuda(), audio.cuda(), text, 1, seed=11)
    assert tuple(codes.shape) == (2, 11, 4) and int(codes.max()) < 128 and int(codes.min()) >= 0
    pn.set_precision("bf16")
    lb = pn.decoder.tokens_to_logits(ids.cuda(), label.cuda(), g["condition"].cuda())
    from tests.helpers import l2_err
    assert l2_err(lb.cpu(), g["logits"]) < 1e-2
    # text features are mandatory exactly when the net was built with text
    with pytest.raises(RuntimeError):
        pn.condition((gt * mask).cuda(), mask.cuda(), audio.cuda())


def test_text_conditioned_train_step_reaches_the_text_encoder():
    """TrainWrapper.__call__ with config.Model.p_text set: loss decreases and TextEnc's parameters move (predictor.py:248-272)."""
    import math
    from probtalk_amd import nets, synthetic
    from probtalk_amd.config import default_transformer_config
    import tempfile
    d = tempfile.mkdtemp(prefix="pt_text_")
    vq = synthetic.write_vq_checkpoint(os.path.join(d, "vq.pth"), 0)
    cfg = default_transformer_config(vq_path=vq)
    cfg.Model.p_text = "bert"
    pred = nets.init_model("s2g_body_predictor", synthetic.args(infer=False), cfg)
    from probtalk_amd.weights import fill_module
    fill_module(pred.AudEnc, 0, "AudEnc.")
    fill_module(pred.TextEnc, 0, "TextEnc.")
    fill_module(pred.Predictor, 0, "PredictorText.")
    w0 = pred.TextEnc.feature_map[0].weight.detach().clone()
    B, T_ = 4, 180                 # config.Data.pose.generate_length frames, like the reference's loader
    bat = {"aud_feat": make_input("tt.aud", (B, 768, T_)), "text_feat": make_input("tt.text", (B, 768, T_)),
           "poses": 0.3 * make_input("tt.poses", (B, 330, T_)), "expression": 0.3 * make_input("tt.expr", (B, 100, T_)),
           "speaker": torch.tensor([1, 3, 0, 2]), "epoch": 10}
    losses = []
    for _ in range(8):
        _, ld = pred(bat)
        assert math.isfinite(ld["loss"])
        losses.append(ld["loss"])
    assert min(losses[-2:]) < losses[0], losses
    assert float((pred.TextEnc.feature_map[0].weight.detach().cpu() - w0).abs().max()) > 0
    # inference with text through the wrapper (predictor.py:398-448: text repeated and resized to the audio length)
    pred.args.infer = True
    out, _ = pred.infer_on_batch(aud=bat["aud_feat"][:1].cuda(), text=make_input("tt.text1", (1, 768, 60)).cuda(),
                                 id=torch.tensor([0]).cuda(), B=2, gt_poses=torch.zeros(1, 376, T_, device="cuda"),
                                 mask=torch.zeros(1, 1, T_, device="cuda"))
    assert tuple(out.shape) == (2, 376, T_) and bool(torch.isfinite(out).all())
