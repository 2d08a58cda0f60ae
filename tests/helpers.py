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
    """max-based relative error: max|a - b| / max|b|."""
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def l2_err(a, b):
    """relative error of the tensor in the 2-norm: ||a - b|| / ||b||."""
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def check_close(out, ref, tol, what, operand_floor=None, floor_factor=1.3):
    """The parity criterion of BOTH arithmetic modes (north_star: "within 1e-4 relative error in fp32, 1e-2 in bf16"):
    relative error ||out - ref|| / ||ref|| < tol against the fp32 oracle / reference golden.  The worst element
    (max|out - ref| / max|ref|) is printed beside it and bounded too:
      * fp32 (no operand_floor): by the same tol (fp32 MFMA differs from the CPU by accumulation order only);
      * bf16: by tol as well, or where the floor is above it by floor_factor x the worst-element error of
        `operand_floor`, the CPU oracle run with nothing changed
        but its matmul operands rounded to bf16 (oracle/bf16_sim.py).  That error (0.9-1.2e-2 of the output range on
        the fixtures; weights alone give 6e-3) is the price of bf16 operands, not of a kernel, so 1e-2 on the worst
        element cannot be promised by ANY bf16-operand implementation; what a kernel can be held to is adding little to it.
    """
    e2, em = l2_err(out, ref), rel_err(out, ref)
    msg = f"{what}: l2-rel {e2:.3e}  max-rel {em:.3e}"
    if operand_floor is not None:
        f2, fm = l2_err(operand_floor, ref), rel_err(operand_floor, ref)
        msg += f"   [bf16-operand floor on the CPU: l2-rel {f2:.3e}  max-rel {fm:.3e}]"
    print(msg)
    assert e2 < tol, msg
    if operand_floor is None:
        assert em < tol, msg
    else:
        assert em < max(tol, floor_factor * fm), msg
    return e2, em


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

# gradients stored in tests/golden/vq_train_b12_t88.npz as grad_<i> (tools/make_golden_vq_train.py)
VQ_TRAIN_GRAD_KEYS = [
    "encoder.project.conv.weight", "encoder.project.norm.weight", "encoder._enc_1._layers.0.conv.weight",
    "encoder._enc_2.norm.bias", "encoder._down_2.residual_layer.weight", "encoder._down_3.conv.bias",
    "encoder._enc_4.conv.weight", "encoder.pre_vq_conv.weight", "encoder.pre_vq_conv.bias",
    "decoder.aft_vq_conv.weight", "decoder._dec_1._layers.1.norm.weight", "decoder._up_2.conv.weight",
    "decoder._up_3.residual_layer.bias", "decoder._dec_3.conv.weight", "decoder._up_4.norm.bias",
    "decoder._dec_4._layers.0.conv.bias", "decoder.project.weight", "decoder.project.bias",
]

# gradients stored in tests/golden/legacy_vq_train_b6_t88.npz as grad_<i> (tools/make_golden_legacy_vq_train.py)
LEGACY_VQ_TRAIN_GRAD_KEYS = [
    "encoder.project.conv.weight", "encoder.project.norm.weight", "encoder._enc_1._layers.0.conv.weight",
    "encoder._down_1.residual_layer.weight", "encoder._enc_2.norm.bias", "encoder._down_2.conv.weight",
    "encoder._enc_3.conv.weight", "encoder.pre_vq_conv.weight", "encoder.pre_vq_conv.bias",
    "decoder.aft_vq_conv.weight", "decoder._dec_1._layers.1.norm.weight", "decoder._up_2.conv.weight",
    "decoder._up_3.residual_layer.bias", "decoder._dec_3.conv.weight", "decoder.project.weight", "decoder.project.bias",
]

# gradients stored in tests/golden/refiner_train_b2_t64.npz as grad_<i> (tools/make_golden_refiner_train.py)
REFINER_TRAIN_GRAD_KEYS = [
    "input_conv.weight", "input_conv.bias", "audio_conv.weight", "class_embed.weight",
    "transformer_decoder.layers.0.self_attn.in_proj_weight", "transformer_decoder.layers.0.norm1.weight",
    "transformer_decoder.layers.1.multihead_attn.in_proj_bias", "transformer_decoder.layers.2.linear1.bias",
    "transformer_decoder.layers.3.style_gourp_2.emb_layers_2.1.weight",
    "transformer_decoder.layers.4.style_gourp_3.out_layers_2.2.weight", "transformer_decoder.layers.5.norm3.bias",
    "output_conv.0.weight", "output_conv.2.bias",
]


REFINER_TRAIN_TAG = "1"


def refiner_train_case(tag=None):
    """(aud, input_poses, gt_poses, mask, label) of the refiner train fixture: B=2, T=64, one clip with a 30-frame
    context window; input = gt*mask + (gt + noise)*(1-mask), the shape of refiner.py:267-270.  The input tag was picked
    (tools/make_golden_refiner_train.py --search) so that no ReLU unit of the six FFNs sits within fp32 rounding of 0:
    the reference's nn.MultiheadAttention and the oracle's explicit attention round differently, and one such unit
    flipping side changes every upstream gradient at the 1e-3 level."""
    from probtalk_amd.weights import make_input
    tag = REFINER_TRAIN_TAG if tag is None else tag
    B, T = 2, 64
    aud = make_input(f"rtrain{tag}.aud", (B, 768, T))
    gt = make_input(f"rtrain{tag}.gt", (B, 376, T))
    mask = torch.zeros(B, 1, T)
    mask[1, :, 10:40] = 1
    vq = gt + 0.3 * make_input(f"rtrain{tag}.vq", (B, 376, T))
    return aud, gt * mask + vq * (1 - mask), gt, mask, torch.tensor([1, 3])


def share_code_case():
    """Inputs of tests/golden/pq_share_code_b12_w22.npz (tools/make_golden_share_code.py): the initial code book of all
    four groups, three latent batches (B=12, C=512, W=22; two train steps + one eval), the two upstream gradients."""
    from probtalk_amd.weights import make_input
    B, C, W, K, G = 12, 512, 22, 1024, 4
    cb = (make_input("pqshare.codebook", (K, C // G)) * 0.05).contiguous()
    xs = [(make_input(f"pqshare.x{i}", (B, C, W)) * 0.05).contiguous() for i in range(3)]
    ws = [make_input(f"pqshare.w{i}", (B, C, W)) for i in range(2)]
    return cb, xs, ws


# ---- the fp32 tap-GEMM has two kernels (option split_bf16): the bf16-MFMA kernel with three-term operand splits (default)
# and the fp32-MFMA kernel.  Both are fp32-accurate (tests/test_gpu_split_bf16.py), but they round differently, so in a
# train step DIFFERENT (Leaky)ReLU units that sit within rounding of 0 land on the other side than in the oracle, and a
# flipped unit moves its whole gradient row.  The train-step parity tests run under both kernels: the bounds of the
# flipped case were set on the fp32-MFMA kernel's outcome and stay as they were for it; the split kernel's are the
# module-level bounds of tests/test_gpu_module_train.py (worst tensor 5e-2 of its maximum, 1e-2 in L2 over all gradients).
TAP_GEMM_KERNELS = [1, 0]
TAP_GEMM_IDS = ["split_bf16", "fp32_mfma"]


class tap_gemm_kernel:
    def __init__(self, split):
        self.split = split

    def __enter__(self):
        from probtalk_amd import _lib
        _lib.check(_lib.lib().pt_set_option(b"split_bf16", self.split))
        return self

    def __exit__(self, *exc):
        from probtalk_amd import _lib
        _lib.check(_lib.lib().pt_set_option(b"split_bf16", 1))
        return False

    def bound(self, fp32_mfma, split_bf16):
        return split_bf16 if self.split else fp32_mfma


class lib_option:
    """with lib_option("sm_rows", 0): ... -- sets a run-time switch of the library and restores its previous value."""
    def __init__(self, name, value):
        self.name, self.value = name.encode(), value

    def __enter__(self):
        import ctypes as C
        from probtalk_amd import _lib
        old = C.c_int(0)
        _lib.check(_lib.lib().pt_get_option(self.name, C.byref(old)))
        self.old = old.value
        _lib.check(_lib.lib().pt_set_option(self.name, self.value))
        return self

    def __exit__(self, *exc):
        from probtalk_amd import _lib
        _lib.check(_lib.lib().pt_set_option(self.name, self.old))
        return False


def pq_tile_case():
    """Inputs of tests/golden/pq_tile_b8_w11.npz (tools/make_golden_pq_tile.py): batches with FEWER latent chunks than codes
    (configs[0]'s own shape, B=8, T=88 -> 88 rows < 128 codes), so that ProductQuantization._tile (vqvae_modules.py:571-580)
    repeats the batch and adds noise.  Per-group books: code books [G,K,D], two latent batches (8,512,11), the N(0,1) draws
    that stand for torch.randn_like, [step][G,K,D].  share_code: one book [K,D], batches (2,512,8) (64 chunks < 128 codes),
    draws [step][K,D]."""
    from probtalk_amd.weights import make_input
    C, K, G = 512, 128, 4
    D = C // G
    case = {"cb": (make_input("pqtile.codebook", (G, K, D)) * 0.05).contiguous(),
            "xs": [(make_input(f"pqtile.x{i}", (8, C, 11)) * 0.05).contiguous() for i in range(2)],
            "noise": [make_input(f"pqtile.noise{i}", (G, K, D)) for i in range(2)],
            "cb_shared": (make_input("pqtile.codebook_shared", (K, D)) * 0.05).contiguous(),
            "xs_shared": [(make_input(f"pqtile.xs{i}", (2, C, 8)) * 0.05).contiguous() for i in range(2)],
            "noise_shared": [make_input(f"pqtile.noise_shared{i}", (K, D)) for i in range(2)]}
    return case
