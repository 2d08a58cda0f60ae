"""Golden fixture of the text-conditioned predictor (Model.p_text set: predictor.py:69-76,248-254, predictornet.py:51) from
the IMPORTED REFERENCE (build container only): PredictorNet(text='bert') has a 1024-channel condition (motion 512 | audio
256 | text 256).  Stored: the TextEnc output, the condition tensor and the logits of one tokens_to_logits pass on generator
weights.  Data only.

    python tools/make_golden_text.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_loader  # noqa: E402
from probtalk_amd.weights import generate_state_dict, make_input  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "predictor_text_b2_t88.npz")


def main():
    ref_loader.load()
    from nets.inpainting.vqvae_1d_sc import ConditionEncoder
    from nets.inpainting.predictornet import PredictorNet
    torch.set_grad_enabled(False)
    te = ConditionEncoder(in_dim=768, num_hiddens=256, num_residual_layers=2, num_residual_hiddens=256).eval()
    te.load_state_dict(generate_state_dict(te.state_dict(), 0, "TextEnc."))
    pn = PredictorNet(False, 376, 128, 512, 10, 6, 4, groups=4, identity=True, maskgit=True, maskgit_T=8, transformer=True,
                      text="bert", audio="wav2vec", motion_context=True).eval()
    assert pn.decoder.condi_emb.in_channels == 1024
    pn.load_state_dict(generate_state_dict(pn.state_dict(), 0, "PredictorText."))
    gt = make_input("pred.gt", (2, 376, 88))
    mask = torch.zeros(2, 1, 88)
    mask[:, :, :30] = 1
    audio = make_input("ptext.audio", (2, 256, 11))                    # AudEnc output stands in (pinned separately)
    text = te(make_input("ptext.text", (2, 768, 88)))
    mm = pn.motion_embedding(torch.cat([gt * mask, mask], dim=1))
    for layer in pn.enc:
        mm = layer(mm)
    cond = torch.cat([mm, audio, text], dim=1)                         # predictornet.py:93-97
    ids = make_input("pred.ids", (2, 11, 4), "int", low=0, high=129)
    logits = pn.decoder.tokens_to_logits(ids, torch.tensor([1, 3]), cond)
    np.savez_compressed(OUT, text=text.numpy(), condition=cond.numpy(), logits=logits.numpy())
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
