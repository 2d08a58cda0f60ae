"""CPU oracle for ProbTalk's token hot path — TEST INFRASTRUCTURE, NOT PRODUCT.

A functional PyTorch-CPU fp32 restatement of the reference's eval-mode
algorithm for the path named by BASELINE.json `north_star`.  Every function
takes a plain `state_dict` (same keys as the reference module it restates) so
the same synthetic weights (probtalk_amd.weights) drive the reference (in the
build container), this oracle, and the HIP path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this module; the product package never does (tests/test_layout.py
greps for it).

Pinning: `tools/make_golden.py` runs the *imported reference modules* on the
same weights/inputs and commits the I/O under tests/golden/;
tests/test_oracle_golden.py checks this file against those fixtures.

Reference lines restated (relative to /root/reference):
  nets/spg/vqvae_modules.py:107-245   ConvNormRelu / Res_CNR_Stack
  nets/spg/vqvae_modules.py:279-377   VectorQuantizerEMA (legacy VQ, eval)
  nets/spg/vqvae_modules.py:437-590   ProductQuantization (eval)
  nets/inpainting/vqvae_1d_sc.py:70-91,197-324  ConditionEncoder, EncoderSC, DecoderSC, VQVAE_SC
  nets/spg/vqvae_1d.py:66-187         legacy Encoder/Decoder/VQVAE
  nets/inpainting/gated_pixelcnn_1d.py:30-83    GatedConv
  nets/inpainting/predictornet.py:13-112        PredictorNet
  nets/inpainting/PTransformer.py:28-165,214-372,394-704  style block, PE, decoder layer, MaskGIT
  nets/inpainting/RTransformer.py:181-230       RTransformer
  nets/inpainting/predictor.py:398-448,597-647  infer_on_batch / infer_to_code
  nets/inpainting/refiner.py:512-558            infer_on_batch
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
LN_EPS = 1e-5
N_HEAD = 8


# --------------------------------------------------------------------------
# conv blocks (vqvae_modules.py:107-245)
# --------------------------------------------------------------------------
def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"],
                        False, 0.0, BN_EPS)


def _leaky(x, slope):
    if slope == 0.0:
        return torch.relu(x)
    return F.leaky_relu(x, slope)


def conv_norm_relu(sd, p, x, sample="none", residual=False, slope=0.2):
    """ConvNormRelu.forward, eval: act(BN(conv(x)) [+ residual_layer(x)])  (vqvae_modules.py:193-200)."""
    w, b = sd[p + "conv.weight"], sd[p + "conv.bias"]
    if sample == "none":
        out = F.conv1d(x, w, b, stride=1, padding=1)
    elif sample == "down":
        out = F.conv1d(x, w, b, stride=2, padding=1)
    elif sample == "up":
        out = F.conv_transpose1d(x, w, b, stride=2, padding=1)
    else:
        raise ValueError(sample)
    out = _bn(sd, p + "norm.", out)
    if residual:
        rw, rb = sd.get(p + "residual_layer.weight"), sd.get(p + "residual_layer.bias")
        if sample == "down":
            out = out + F.conv1d(x, rw, rb, stride=2, padding=1)
        elif sample == "up":
            out = out + F.conv_transpose1d(x, rw, rb, stride=2, padding=1)
        elif rw is None:
            out = out + x
        else:
            out = out + F.conv1d(x, sd[p + "residual_layer.0.weight"], sd[p + "residual_layer.0.bias"], padding=1)
    return _leaky(out, slope)


def res_cnr_stack(sd, p, x, layers=2, slope=0.2):
    """Res_CNR_Stack.forward: relu(BN(conv(layers(x))) + x)  (vqvae_modules.py:238-245)."""
    h = x
    for i in range(layers):
        h = conv_norm_relu(sd, f"{p}_layers.{i}.", h, "none", False, slope)
    h = _bn(sd, p + "norm.", F.conv1d(h, sd[p + "conv.weight"], sd[p + "conv.bias"], padding=1))
    return torch.relu(h + x)


# --------------------------------------------------------------------------
# PQ-VAE (vqvae_1d_sc.py:197-324; vqvae_modules.py:437-590)
# --------------------------------------------------------------------------
def encoder_sc(sd, p, x):
    h = conv_norm_relu(sd, p + "project.", x)
    h = res_cnr_stack(sd, p + "_enc_1.", h)
    h = conv_norm_relu(sd, p + "_down_1.", h, "down", True)
    h = res_cnr_stack(sd, p + "_enc_2.", h)
    h = conv_norm_relu(sd, p + "_down_2.", h, "down", True)
    h = res_cnr_stack(sd, p + "_enc_3.", h)
    h = conv_norm_relu(sd, p + "_down_3.", h, "down", True)
    h = res_cnr_stack(sd, p + "_enc_4.", h)
    return F.conv1d(h, sd[p + "pre_vq_conv.weight"], sd[p + "pre_vq_conv.bias"])


def decoder_sc(sd, p, e):
    h = F.conv1d(e, sd[p + "aft_vq_conv.weight"], sd[p + "aft_vq_conv.bias"])
    h = res_cnr_stack(sd, p + "_dec_1.", h)
    h = conv_norm_relu(sd, p + "_up_2.", h, "up", True)
    h = res_cnr_stack(sd, p + "_dec_2.", h)
    h = conv_norm_relu(sd, p + "_up_3.", h, "up", True)
    h = res_cnr_stack(sd, p + "_dec_3.", h)
    h = conv_norm_relu(sd, p + "_up_4.", h, "up", True)
    h = res_cnr_stack(sd, p + "_dec_4.", h)
    return F.conv1d(h, sd[p + "project.weight"], sd[p + "project.bias"])


def code_distances(flat_x, emb):
    """(sum x^2 + sum e^2) - 2 x E^T, the reference's association order (vqvae_modules.py:549-553)."""
    return (torch.sum(flat_x ** 2, dim=1, keepdim=True) + torch.sum(emb ** 2, dim=1)
            - 2.0 * torch.matmul(flat_x, emb.t()))


def pq_quantize(sd, p, z, groups):
    """ProductQuantization.forward, eval, r=0: returns (quantized (B,C,W) contiguous, indices (B,W,G) int64)."""
    B, C, W = z.shape
    D = C // groups
    flat = z.permute(0, 2, 1).reshape(-1, groups, D)
    idx, q = [], []
    for i in range(groups):
        emb = sd[f"{p}embeddings_{i}"]
        ind = torch.argmin(code_distances(flat[:, i], emb), dim=1)
        idx.append(ind)
        q.append(F.embedding(ind, emb).reshape(B, W, D))
    quantized = torch.cat(q, dim=-1).permute(0, 2, 1).contiguous()
    indices = torch.stack(idx, dim=1).reshape(B, W, groups)
    return quantized, indices


def pq_quantize_all(sd, p, indices):
    """quantize_all: (B,W,G) int -> (B,W,G*D)  (vqvae_modules.py:557-564)."""
    return torch.cat([F.embedding(indices[..., i].long(), sd[f"{p}embeddings_{i}"])
                      for i in range(indices.shape[-1])], dim=-1)


def pq_margins(sd, p, z, groups):
    """fp64 gap between best and second-best code per (row, group): used to classify near-ties."""
    B, C, W = z.shape
    D = C // groups
    flat = z.permute(0, 2, 1).reshape(-1, groups, D).double()
    out = []
    for i in range(groups):
        d = code_distances(flat[:, i], sd[f"{p}embeddings_{i}"].double())
        top = torch.topk(d, 2, dim=1, largest=False).values
        out.append(top[:, 1] - top[:, 0])
    return torch.stack(out, dim=1).reshape(B, W, groups)


def vqvae_encode(sd, x, groups=4):
    """VQVAE_SC.encode (vqvae_1d_sc.py:317-320): z, quantized, indices."""
    z = encoder_sc(sd, "encoder.", x)
    q, idx = pq_quantize(sd, "vq_layer.", z, groups)
    return z, q, idx


def vqvae_decode(sd, e):
    return decoder_sc(sd, "decoder.", e)


def vqvae_forward(sd, x, groups=4):
    z, q, idx = vqvae_encode(sd, x, groups)
    return z, q, idx, vqvae_decode(sd, q)


# --------------------------------------------------------------------------
# legacy single-codebook VQ-VAE (vqvae_1d.py:66-187; vqvae_modules.py:279-377)
# --------------------------------------------------------------------------
def legacy_encoder(sd, p, x):
    h = conv_norm_relu(sd, p + "project.", x)
    h = res_cnr_stack(sd, p + "_enc_1.", h)
    h = conv_norm_relu(sd, p + "_down_1.", h, "down", True)
    h = res_cnr_stack(sd, p + "_enc_2.", h)
    h = conv_norm_relu(sd, p + "_down_2.", h, "down", True)
    h = res_cnr_stack(sd, p + "_enc_3.", h)
    return F.conv1d(h, sd[p + "pre_vq_conv.weight"], sd[p + "pre_vq_conv.bias"])


def legacy_decoder(sd, p, e):
    h = F.conv1d(e, sd[p + "aft_vq_conv.weight"], sd[p + "aft_vq_conv.bias"])
    h = res_cnr_stack(sd, p + "_dec_1.", h)
    h = conv_norm_relu(sd, p + "_up_2.", h, "up", True)
    h = res_cnr_stack(sd, p + "_dec_2.", h)
    h = conv_norm_relu(sd, p + "_up_3.", h, "up", True)
    h = res_cnr_stack(sd, p + "_dec_3.", h)
    return F.conv1d(h, sd[p + "project.weight"], sd[p + "project.bias"])


def legacy_vq_forward(sd, x):
    """VQVAE.forward eval (vqvae_1d.py:163-168): e (B,D,W), indices (B,W), recon."""
    z = legacy_encoder(sd, "encoder.", x)
    B, D, W = z.shape
    emb = sd["vq_layer.embeddings"]
    flat = z.permute(0, 2, 1).reshape(-1, D)
    ind = torch.argmin(code_distances(flat, emb), dim=1)
    e = F.embedding(ind, emb).reshape(B, W, D).permute(0, 2, 1).contiguous()
    return z, e, ind.reshape(B, W), legacy_decoder(sd, "decoder.", e)


# --------------------------------------------------------------------------
# audio condition encoder (vqvae_1d_sc.py:70-91)
# --------------------------------------------------------------------------
def condition_encoder(sd, x, p=""):
    h = F.conv1d(x, sd[p + "feature_map.0.weight"], sd[p + "feature_map.0.bias"])
    h = F.leaky_relu(_bn(sd, p + "feature_map.1.", h), 0.1)
    for i in (1, 2, 3):
        h = conv_norm_relu(sd, f"{p}_down_{i}.", h, "down", True)
    return h


# --------------------------------------------------------------------------
# motion-context gated convs (gated_pixelcnn_1d.py:39-83, predictornet.py:31-49,87-100)
# --------------------------------------------------------------------------
GATED_SPEC = [  # (kernel, stride, padding) for enc.0 .. enc.9
    (7, 1, 3), (3, 1, 1), (4, 2, 1), (3, 1, 1), (4, 2, 1), (3, 1, 1), (4, 2, 1), (3, 1, 1), (3, 1, 1), (3, 1, 1)]


def gated_conv(sd, p, x, k, s, pad):
    hv = F.conv1d(x, sd[p + "vert_stack.weight"], sd[p + "vert_stack.bias"], stride=s, padding=pad)
    a, b = hv.chunk(2, dim=1)
    out = torch.tanh(a) * torch.sigmoid(b)
    out = F.conv1d(out, sd[p + "res.weight"], sd[p + "res.bias"])
    if s == 2:
        return out + F.conv1d(x, sd[p + "conv.weight"], sd[p + "conv.bias"], stride=s, padding=pad)
    return out + x


def predictor_condition(sd, masked_motion, mask, audio, p=""):
    """PredictorNet.forward up to the decoder call: (B,768,T') condition tensor."""
    mm = torch.cat([masked_motion, mask], dim=1)
    mm = F.conv1d(mm, sd[p + "motion_embedding.weight"], sd[p + "motion_embedding.bias"])
    for i, (k, s, pad) in enumerate(GATED_SPEC):
        mm = gated_conv(sd, f"{p}enc.{i}.", mm, k, s, pad)
    return torch.cat([mm, audio], dim=1)


# --------------------------------------------------------------------------
# transformer pieces (PTransformer.py)
# --------------------------------------------------------------------------
def sinusoid_table(d_model, max_seq_len):
    """PositionalEncoding table incl. the zero pad row 0, float64 (PTransformer.py:227-245)."""
    pe = np.array([[pos / np.power(10000, 2.0 * (j // 2) / d_model) for j in range(d_model)]
                   for pos in range(max_seq_len)])
    pe[:, 0::2] = np.sin(pe[:, 0::2])
    pe[:, 1::2] = np.cos(pe[:, 1::2])
    return torch.cat((torch.zeros([1, d_model]), torch.from_numpy(pe)))


def mha(sd, p, q_in, kv_in):
    """nn.MultiheadAttention forward, eval, no masks: (B,Nq,C),(B,Nk,C) -> (B,Nq,C)."""
    w, b = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    C = q_in.shape[-1]
    q = F.linear(q_in, w[:C], b[:C])
    k = F.linear(kv_in, w[C:2 * C], b[C:2 * C])
    v = F.linear(kv_in, w[2 * C:], b[2 * C:])
    B, Nq, _ = q.shape
    Nk = k.shape[1]
    hd = C // N_HEAD
    q = q.reshape(B, Nq, N_HEAD, hd).transpose(1, 2)
    k = k.reshape(B, Nk, N_HEAD, hd).transpose(1, 2)
    v = v.reshape(B, Nk, N_HEAD, hd).transpose(1, 2)
    att = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
    o = torch.matmul(att, v).transpose(1, 2).reshape(B, Nq, C)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def style_block(sd, p, h, label_embed):
    """StylizationBlock_V3 type 'sb' (PTransformer.py:134-140): per-token adaLN -> SiLU -> Linear."""
    emb = F.linear(F.silu(label_embed), sd[p + "emb_layers_2.1.weight"], sd[p + "emb_layers_2.1.bias"]).unsqueeze(1)
    scale, shift = torch.chunk(emb, 2, dim=2)
    C = h.shape[-1]
    h = F.layer_norm(h, (C,), sd[p + "norm_2.weight"], sd[p + "norm_2.bias"], LN_EPS) * (1 + scale) + shift
    return F.linear(F.silu(h), sd[p + "out_layers_2.2.weight"], sd[p + "out_layers_2.2.bias"])


def decoder_layer(sd, p, x, mem, label_embed):
    """DeepSupervisionLayer.forward, batch-first restatement (PTransformer.py:314-372)."""
    C = x.shape[-1]

    def ln(t, name):
        return F.layer_norm(t, (C,), sd[p + name + ".weight"], sd[p + name + ".bias"], LN_EPS)

    x = ln(x + style_block(sd, p + "style_gourp_1.", mha(sd, p + "self_attn.", x, x), label_embed), "norm1")
    x = ln(x + style_block(sd, p + "style_gourp_2.", mha(sd, p + "multihead_attn.", x, mem), label_embed), "norm2")
    ff = F.linear(torch.relu(F.linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])),
                  sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    x = ln(x + style_block(sd, p + "style_gourp_3.", ff, label_embed), "norm3")
    return x


def _num_layers(sd, p):
    n = 0
    while f"{p}transformer_decoder.layers.{n}.norm1.weight" in sd:
        n += 1
    return n


def tokens_to_logits(sd, ids, label, condition, p="decoder."):
    """PTransformer.tokens_to_logits (PTransformer.py:657-694): ids (B,T,G) -> logits (B,T,G,num_code)."""
    b, t, g = ids.shape
    x = F.embedding(ids.long(), sd[p + "tok_emb.weight"]).reshape(b, t * g, -1)
    label_embed = F.embedding(label.long(), sd[p + "class_embed.weight"])
    pe = sd[p + "PE.position_encoding.weight"]            # float64 (1801, C)
    ppe = sd[p + "product_PE.position_encoding.weight"]   # float64 (G+1, C)
    c_pe = pe[1:t + 1]
    x_pe = c_pe.repeat_interleave(g, 0) + ppe[1:g + 1].repeat(t, 1)
    x = (x + x_pe).to(torch.float32)
    cond = F.conv1d(condition, sd[p + "condi_emb.weight"], sd[p + "condi_emb.bias"]).transpose(1, 2)
    cond = (cond + c_pe).to(torch.float32)
    for i in range(_num_layers(sd, p)):
        x = decoder_layer(sd, f"{p}transformer_decoder.layers.{i}.", x, cond, label_embed)
    h = x.transpose(1, 2)
    h = torch.relu(F.conv1d(h, sd[p + "output_conv.0.weight"], sd[p + "output_conv.0.bias"]))
    h = F.conv1d(h, sd[p + "output_conv.2.weight"], sd[p + "output_conv.2.bias"])
    return h.permute(0, 2, 1).reshape(b, t, g, -1)


def categorical_from_noise(logits, q_exp):
    """Categorical(logits).sample() with the Exp(1) draw injected: torch.multinomial's
    single-draw path is argmax(probs / q), q ~ Exp(1)."""
    probs = torch.softmax(logits, dim=-1)
    return torch.argmax(probs / q_exp, dim=-1)


def maskgit_step(logits, cur_ids, q_exp, gumbel, step, T, mask_id, n_total, temperature=4.5):
    """One iteration of sample_good's loop body after the logits call (PTransformer.py:607-638)."""
    sampled = categorical_from_noise(logits, q_exp).to(torch.int32)
    unknown = cur_ids == mask_id
    sampled = torch.where(unknown, sampled, cur_ids)
    ratio = 1.0 * (step + 1) / T
    mask_ratio = np.cos(ratio * np.pi / 2)
    probs = torch.softmax(logits, dim=-1)
    sel = torch.take_along_dim(probs, sampled.long().unsqueeze(-1), -1).squeeze(-1)
    sel = torch.where(unknown, sel, torch.tensor(float("inf")))
    unknown0 = torch.full((cur_ids.shape[0],), n_total, dtype=torch.int64)
    mask_len = torch.floor(unknown0 * mask_ratio).unsqueeze(1)
    mask_len = torch.maximum(torch.zeros_like(mask_len),
                             torch.minimum(unknown.sum(dim=-1, keepdim=True) - 1, mask_len))
    conf = torch.log(sel) + (temperature * (1.0 - ratio)) * gumbel
    cut = torch.take_along_dim(torch.sort(conf, dim=-1).values, mask_len.long(), dim=-1)
    return torch.where(conf < cut, torch.tensor(mask_id, dtype=torch.int32), sampled)


def sample_good(sd, label, condition, noise_q, noise_g, groups=4, T=8, num_code=128, p="decoder.",
                return_logits=False):
    """MaskGIT decoding (PTransformer.py:590-642) with injected noise
    noise_q: (T,B,N,num_code) Exp(1) draws, noise_g: (T,B,N) Gumbel(0,1) draws."""
    B, Tm = condition.shape[0], condition.shape[2]
    N = Tm * groups
    ids = torch.full((B, N), num_code, dtype=torch.int32)
    all_logits = []
    for t in range(T):
        logits = tokens_to_logits(sd, ids.reshape(B, Tm, groups), label, condition, p).reshape(B, N, -1)
        all_logits.append(logits)
        ids = maskgit_step(logits, ids, noise_q[t], noise_g[t], t, T, num_code, N)
    ids = ids.reshape(B, Tm, groups)
    return (ids, all_logits) if return_logits else ids


# --------------------------------------------------------------------------
# refiner network (RTransformer.py:209-230)
# --------------------------------------------------------------------------
def rtransformer(sd, x, condition, mask, label):
    T = x.shape[2]
    h = F.conv1d(x, sd["input_conv.weight"], sd["input_conv.bias"])
    pe = sd["PE.position_encoding.weight"][1:T + 1]
    h = (h.transpose(1, 2) + pe).to(torch.float32)
    label_embed = F.embedding(label.long(), sd["class_embed.weight"])
    onehot = F.one_hot(mask.to(torch.long), 2).squeeze(1).transpose(1, 2)
    cond = torch.cat([condition, onehot], dim=1)
    cond = F.conv1d(cond, sd["audio_conv.weight"], sd["audio_conv.bias"]).transpose(1, 2)
    cond = (cond + pe).to(torch.float32)
    for i in range(_num_layers(sd, "")):
        h = decoder_layer(sd, f"transformer_decoder.layers.{i}.", h, cond, label_embed)
    h = h.transpose(1, 2)
    h = torch.relu(F.conv1d(h, sd["output_conv.0.weight"], sd["output_conv.0.bias"]))
    return F.conv1d(h, sd["output_conv.2.weight"], sd["output_conv.2.bias"])


# --------------------------------------------------------------------------
# wrapper-level inference glue (predictor.py:398-448; refiner.py:512-558)
# --------------------------------------------------------------------------
def predictor_infer(sd_aud, sd_pred, sd_vq, aud, gt_poses, mask, label, noise_q, noise_g, groups=4, T=8):
    """predictor.TrainWrapper.infer_on_batch / infer_to_code after the repeat(B): returns (poses, codes)."""
    input_poses = gt_poses * mask
    audio = condition_encoder(sd_aud, aud)
    cond = predictor_condition(sd_pred, input_poses, mask, audio)
    codes = sample_good(sd_pred, label, cond, noise_q, noise_g, groups, T)
    e = pq_quantize_all(sd_vq, "vq_layer.", codes).transpose(1, 2)
    pred = vqvae_decode(sd_vq, e)
    pred = F.interpolate(pred, size=aud.shape[2], mode="linear")
    return input_poses * mask + pred * (1 - mask), codes


def refiner_infer(sd_ref, aud, gt_poses, mask, label, pred_poses):
    """refiner.TrainWrapper.infer_on_batch (two_stage, type 'fbhe')."""
    pred = F.interpolate(pred_poses, size=aud.shape[2], mode="linear")
    blank = gt_poses * mask + torch.zeros_like(pred) + pred * (1 - mask)
    out = rtransformer(sd_ref, blank, aud, mask, label)
    return gt_poses * mask + out * (1 - mask)


# --------------------------------------------------------------------------
# sliding-window callers of the path (scripts/test_holisticbody.py:239-262; scripts/demo.py:113-178)
# --------------------------------------------------------------------------
def holistic_slices_infer(sd_aud, sd_pred, sd_vq, sd_ref, aud, gt_poses, mask, label, noises):
    """test_holisticbody.py:239-262 with predictor -> refiner per slice (ai1.py:51-57).  aud (B,768,L) already repeated,
    gt_poses (B,C,L), mask (B,1,L); noises[i] = (noise_q, noise_g) of slice i."""
    L = aud.shape[-1]
    num_slices = 1 + math.ceil((L - 180) / 150)
    input_gt, input_mask, pred = gt_poses.clone(), mask.clone(), None
    for i in range(num_slices):
        s = 0 if i == 0 else 150 + 180 * (i - 1)
        e = 180 if i == 0 else 150 + 180 * i
        a, g, m = aud[..., s:e], input_gt[..., s:e], input_mask[..., s:e]
        body, _ = predictor_infer(sd_aud, sd_pred, sd_vq, a, g, m, label, noises[i][0], noises[i][1])
        pred_0 = refiner_infer(sd_ref, a, g, m, label, body)
        pred = pred_0 if pred is None else torch.cat([pred, pred_0[..., 30:]], -1)
        input_gt[..., s:e] = pred_0
        input_mask[..., s:e] = 1
    return gt_poses * mask + pred * (1 - mask)


def demo_slices_infer(sd_aud, sd_pred, sd_vq, sd_ref, aud, label, noises, C=376):
    """demo.py:113-178: predictor codes per 180-frame slice (stride 150), ONE decode of the joined code grid, refiner over
    176-frame slices (stride 146).  Returns (refined, preliminary_motion, codes)."""
    B, _, t = aud.shape
    input_gt, input_mask = torch.zeros(B, C, t), torch.zeros(B, 1, t)
    code_list = []
    for i in range(math.ceil((t - 30) / 150)):
        s, e = 150 * i, 150 * i + 180
        pred_0, codes = predictor_infer(sd_aud, sd_pred, sd_vq, aud[..., s:e], input_gt[..., s:e], input_mask[..., s:e], label,
                                        noises[i][0], noises[i][1])
        code_list.append(codes)
        input_gt[..., s:e] = pred_0
        input_mask[..., s:e] = 1
    codes = torch.cat([code_list[0]] + [c[:, 3:] for c in code_list[1:]], dim=1)
    prelim = vqvae_decode(sd_vq, pq_quantize_all(sd_vq, "vq_layer.", codes).transpose(1, 2))
    prelim = F.interpolate(prelim, size=t, mode="linear")
    input_mask, pred_all = torch.zeros(B, 1, t), None
    for i in range(math.ceil((t - 30) / 146)):
        s, e = 146 * i, 146 * i + 176
        pred_0 = refiner_infer(sd_ref, aud[..., s:e], input_gt[..., s:e], input_mask[..., s:e], label, prelim[..., s:e])
        pred_all = pred_0 if pred_all is None else torch.cat([pred_all, pred_0[..., 30:]], -1)
        input_gt[..., s:e] = pred_0
        input_mask[..., s:e] = 1
    return pred_all, prelim, codes
