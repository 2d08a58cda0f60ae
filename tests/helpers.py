"""Shared test helpers: synthetic state_dicts from the committed manifest of the
reference's state_dict keys (tests/golden/state_dict_manifest.json)."""
import functools
import json
import os

import numpy as np
import torch

from probtalk_amd.weights import make_tensor

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@functools.lru_cache(None)
def manifest():
    with open(os.path.join(GOLDEN, "state_dict_manifest.json")) as f:
        return json.load(f)


def golden(name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, name)).items()}


@functools.lru_cache(None)
def _sd_cached(module, seed):
    from oracle.probtalk_oracle import sinusoid_table
    out = {}
    for k, (shape, dtype) in manifest()[module].items():
        if "position_encoding" in k:
            out[k] = sinusoid_table(shape[1], shape[0] - 1)
        else:
            out[k] = make_tensor(f"{module}.{k}", shape, getattr(torch, dtype), seed)
    return out


def ref_state_dict(module, seed=0):
    """Synthetic weights for reference module `module` in {VQ, g, AudEnc, Predictor, HFTransformer}."""
    return dict(_sd_cached(module, seed))


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


# gradients stored in tests/golden/predictor_train_b2_t88.npz as grad_<i> (tools/make_golden_train.py)
TRAIN_GRAD_KEYS = [
    ("AudEnc", "feature_map.0.weight"), ("AudEnc", "feature_map.1.weight"), ("AudEnc", "_down_2.conv.weight"),
    ("AudEnc", "_down_3.residual_layer.bias"), ("AudEnc", "_down_3.norm.bias"),
    ("Predictor", "motion_embedding.bias"), ("Predictor", "enc.0.vert_stack.weight"), ("Predictor", "enc.2.conv.weight"),
    ("Predictor", "enc.9.res.weight"), ("Predictor", "decoder.tok_emb.weight"), ("Predictor", "decoder.class_embed.weight"),
    ("Predictor", "decoder.condi_emb.bias"),
    ("Predictor", "decoder.transformer_decoder.layers.0.self_attn.in_proj_bias"),
    ("Predictor", "decoder.transformer_decoder.layers.0.style_gourp_1.emb_layers_2.1.bias"),
    ("Predictor", "decoder.transformer_decoder.layers.0.style_gourp_1.norm_2.weight"),
    ("Predictor", "decoder.transformer_decoder.layers.2.multihead_attn.in_proj_bias"),
    ("Predictor", "decoder.transformer_decoder.layers.3.linear1.bias"),
    ("Predictor", "decoder.transformer_decoder.layers.5.norm3.weight"),
    ("Predictor", "decoder.transformer_decoder.layers.5.style_gourp_3.out_layers_2.2.bias"),
    ("Predictor", "decoder.output_conv.2.bias"),
]
