"""Training-mode forward of the predictor path on the HIP kernels (functional, keyed by the reference's parameter names).

Restates, in train mode: ConditionEncoder.forward (nets/inpainting/vqvae_1d_sc.py:86-91 with batch-statistic BatchNorm
and Dropout, nets/spg/vqvae_modules.py:193-200), PredictorNet.forward train branch (nets/inpainting/predictornet.py:
76-104), GatedConv (gated_pixelcnn_1d.py:74-83), PTransformer.forward (PTransformer.py:443-504) with
DeepSupervisionLayer (314-372) and StylizationBlock_V3 'sb' (134-140), and the cross-entropy of predictor.py:261.
Every operator is a `train_ops` autograd Function whose forward and backward are HIP kernels.
"""
import math

import torch

from . import train_ops as T


def tensors_of(module):
    """name -> tensor for parameters and buffers (BatchNorm running statistics are updated in place)."""
    d = dict(module.named_parameters())
    d.update(dict(module.named_buffers()))
    return d


# ------------------------------------------------------------------------------------------- audio condition encoder
def cond_encoder_train(P, x_rows, B, Tn, p_drop, seeds, prefix=""):
    """(B*T, in) rows -> (B*T/8, 256) rows."""
    g = lambda k: P[prefix + k]
    h = T.conv(x_rows, g("feature_map.0.weight"), g("feature_map.0.bias"), 0, B, Tn)
    h = T.bn_train(h, g("feature_map.1.weight"), g("feature_map.1.bias"), g("feature_map.1.running_mean"),
                   g("feature_map.1.running_var"), slope=0.1)
    h = T.dropout(h, p_drop, seeds)
    L = Tn
    for i in (1, 2, 3):
        p = f"_down_{i}."
        c = T.conv(h, g(p + "conv.weight"), g(p + "conv.bias"), 3, B, L)
        c = T.dropout(c, p_drop, seeds)
        r = T.conv(h, g(p + "residual_layer.weight"), g(p + "residual_layer.bias"), 3, B, L)
        h = T.bn_train(c, g(p + "norm.weight"), g(p + "norm.bias"), g(p + "norm.running_mean"), g(p + "norm.running_var"),
                       slope=0.2, res=r)
        L = (L - 2) // 2 + 1
    return h, L


# ------------------------------------------------------------------------------------------- motion-context encoder
relu_trace = None      # tests set this to a list to receive every (Leaky)ReLU output in call order (which side of 0 a unit fell)

GATED_KIND = [2, 1, 3, 1, 3, 1, 3, 1, 1, 1]     # enc.0 .. enc.9: k7 | k3 | k4 s2 | ... (predictornet.py:36-49)


def motion_context_train(P, mm_rows, B, Tn):
    """cat(masked_motion, mask) rows (B*T, 384) -> (B*T/8, 512) rows."""
    h = T.conv(mm_rows, P["motion_embedding.weight"], P["motion_embedding.bias"], 0, B, Tn)
    L = Tn
    for i, kind in enumerate(GATED_KIND):
        p = f"enc.{i}."
        hv = T.conv(h, P[p + "vert_stack.weight"], P[p + "vert_stack.bias"], kind, B, L)
        Lo = (L - 2) // 2 + 1 if kind == 3 else L
        gte = T.Gate.apply(hv)
        skip = T.conv(h, P[p + "conv.weight"], P[p + "conv.bias"], 3, B, L) if kind == 3 else h
        h = T.conv(gte, P[p + "res.weight"], P[p + "res.bias"], 0, B, Lo, res=skip)
        L = Lo
    return h, L


# ------------------------------------------------------------------------------------------- transformer
def _style(P, p, t, silu_ce, label, rows_per_clip):
    table = T.conv(silu_ce, P[p + "emb_layers_2.1.weight"], P[p + "emb_layers_2.1.bias"], 0, 1, silu_ce.shape[0])
    y = T.layer_norm(t, P[p + "norm_2.weight"], P[p + "norm_2.bias"])
    h = T.ModSiLU.apply(y, table, label, rows_per_clip)
    return T.conv(h, P[p + "out_layers_2.2.weight"], P[p + "out_layers_2.2.bias"], 0, 1, t.shape[0])


def decoder_layer_train(P, p, x, mem, silu_ce, label, B, N, Nm, p_drop, seeds):
    R = B * N
    qkv = T.conv(x, P[p + "self_attn.in_proj_weight"], P[p + "self_attn.in_proj_bias"], 0, 1, R)
    att = T.attention_self(qkv, B, N, p_drop, seeds)
    t = T.conv(att, P[p + "self_attn.out_proj.weight"], P[p + "self_attn.out_proj.bias"], 0, 1, R)
    t = T.dropout(_style(P, p + "style_gourp_1.", t, silu_ce, label, N), p_drop, seeds)
    x = T.layer_norm(x, P[p + "norm1.weight"], P[p + "norm1.bias"], res=t)

    w, b = P[p + "multihead_attn.in_proj_weight"], P[p + "multihead_attn.in_proj_bias"]
    q = T.conv(x, w[:512], b[:512], 0, 1, R)
    kv = T.conv(mem, w[512:], b[512:], 0, 1, B * Nm)
    att = T.attention_cross(q, kv, B, N, Nm, p_drop, seeds)
    t = T.conv(att, P[p + "multihead_attn.out_proj.weight"], P[p + "multihead_attn.out_proj.bias"], 0, 1, R)
    t = T.dropout(_style(P, p + "style_gourp_2.", t, silu_ce, label, N), p_drop, seeds)
    x = T.layer_norm(x, P[p + "norm2.weight"], P[p + "norm2.bias"], res=t)

    hdn = T.conv(x, P[p + "linear1.weight"], P[p + "linear1.bias"], 0, 1, R, slope=0.0)
    if relu_trace is not None:
        relu_trace.append(hdn.detach())
    hdn = T.dropout(hdn, p_drop, seeds)
    t = T.conv(hdn, P[p + "linear2.weight"], P[p + "linear2.bias"], 0, 1, R)
    t = T.dropout(_style(P, p + "style_gourp_3.", t, silu_ce, label, N), p_drop, seeds)
    return T.layer_norm(x, P[p + "norm3.weight"], P[p + "norm3.bias"], res=t)


def ptransformer_train(P, ids, label, cond_rows, B, Tq, G, p_drop, seeds, prefix="decoder."):
    """ids (B, Tq, G) with the mask id where masked; cond_rows (B*Tq, 768) -> logits rows (B*Tq*G, num_code)."""
    g = lambda k: P[prefix + k]
    N = Tq * G
    ids32 = ids.reshape(-1).to(torch.int32).contiguous()
    pe, ppe = g("PE.position_encoding.weight"), g("product_PE.position_encoding.weight")
    x = T.EmbedPE.apply(ids32, g("tok_emb.weight"), pe, ppe, N, G)
    c_pe = pe[1:Tq + 1].to(torch.float32).repeat(B, 1).contiguous()          # frozen sinusoid rows, per clip
    mem = T.conv(cond_rows, g("condi_emb.weight"), g("condi_emb.bias"), 0, 1, B * Tq, res=c_pe)
    ce = g("class_embed.weight")
    silu_ce = ce * torch.sigmoid(ce)                                         # (n_classes, 512): 2k elements, torch glue
    n_layers = 0
    while prefix + f"transformer_decoder.layers.{n_layers}.norm1.weight" in P:
        n_layers += 1
    for i in range(n_layers):
        x = decoder_layer_train(P, f"{prefix}transformer_decoder.layers.{i}.", x, mem, silu_ce, label, B, N, Tq, p_drop, seeds)
    h = T.conv(x, g("output_conv.0.weight"), g("output_conv.0.bias"), 0, 1, B * N, slope=0.0)
    return T.conv(h, g("output_conv.2.weight"), g("output_conv.2.bias"), 0, 1, B * N)


def random_token(state, keep_scores, epoch_ratio, mask_id):
    """PTransformer.random_token('mask') (PTransformer.py:559-576): keep the r = floor((1 - ratio) * N) tokens with the
    largest scores, replace the rest by the mask id.  keep_scores (B, N) plays torch.rand's role."""
    B = state.shape[0]
    flat = state.reshape(B, -1)
    r = math.floor((1 - epoch_ratio) * flat.shape[1])
    keep = torch.zeros_like(flat, dtype=torch.bool)
    if r > 0:
        keep.scatter_(1, keep_scores.topk(r, dim=1).indices, True)
    return torch.where(keep, flat, torch.full_like(flat, mask_id)).reshape(state.shape)


def predictor_loss(Pa, Pp, aud, input_poses, mask, label, state, keep_scores, epoch_ratio, p_drop=0.0, seeds=None, groups=4):
    """One forward of predictor.py:232-261: returns (loss, logits rows)."""
    B, _, Tn = aud.shape
    aud_rows = T.ncl_to_rows(aud)
    audio, Tq = cond_encoder_train(Pa, aud_rows, B, Tn, p_drop, seeds)
    mm = T.ncl_to_rows(torch.cat([input_poses, mask], dim=1))
    motion, Tq2 = motion_context_train(Pp, mm, B, Tn)
    assert Tq == Tq2
    cond = torch.cat([motion, audio], dim=1).contiguous()
    ids = random_token(state, keep_scores, epoch_ratio, Pp["decoder.tok_emb.weight"].shape[0] - 1)
    logits = ptransformer_train(Pp, ids, label.to(torch.int64).contiguous(), cond, B, Tq, groups, p_drop, seeds)
    loss = T.CrossEntropy.apply(logits[:, :Pp["decoder.output_conv.2.weight"].shape[0]].contiguous(),
                                state.reshape(-1).to(torch.int64).contiguous())
    return loss, logits


# ------------------------------------------------------------------------------------------- PQ-VAE (vq_teacher)
def _cnr(P, p, x, B, L, sample="none", residual=False):
    """ConvNormRelu in train mode (vqvae_modules.py:193-200): LeakyReLU(0.2)(BN_batch(conv(x)) [+ residual_layer(x)])."""
    kind = {"none": 1, "down": 3, "up": 4}[sample]
    c = T.conv(x, P[p + "conv.weight"], P[p + "conv.bias"], kind, B, L)
    r = T.conv(x, P[p + "residual_layer.weight"], P[p + "residual_layer.bias"], kind, B, L) if residual else None
    y = T.bn_train(c, P[p + "norm.weight"], P[p + "norm.bias"], P[p + "norm.running_mean"], P[p + "norm.running_var"],
                   slope=0.2, res=r)
    if relu_trace is not None:
        relu_trace.append(y.detach())
    return y, {"none": L, "down": (L - 2) // 2 + 1, "up": 2 * L}[sample]


def _res_stack(P, p, x, B, L, layers=2):
    """Res_CNR_Stack (vqvae_modules.py:238-245): relu(BN(conv(layers(x))) + x)."""
    h = x
    for i in range(layers):
        h, _ = _cnr(P, f"{p}_layers.{i}.", h, B, L)
    c = T.conv(h, P[p + "conv.weight"], P[p + "conv.bias"], 1, B, L)
    y = T.bn_train(c, P[p + "norm.weight"], P[p + "norm.bias"], P[p + "norm.running_mean"], P[p + "norm.running_var"],
                   slope=0.0, res=x)
    if relu_trace is not None:
        relu_trace.append(y.detach())
    return y


def encoder_sc_train(P, x_rows, B, L, p="encoder."):
    """EncoderSC.forward (vqvae_1d_sc.py:220-235) in train mode: rows (B*T, 384) -> z rows (B*T/8, 512)."""
    h, L = _cnr(P, p + "project.", x_rows, B, L)
    h = _res_stack(P, p + "_enc_1.", h, B, L)
    for i in (1, 2, 3):
        h, L = _cnr(P, f"{p}_down_{i}.", h, B, L, "down", True)
        h = _res_stack(P, f"{p}_enc_{i + 1}.", h, B, L)
    return T.conv(h, P[p + "pre_vq_conv.weight"], P[p + "pre_vq_conv.bias"], 0, B, L), L


def decoder_sc_train(P, e_rows, B, L, p="decoder."):
    """DecoderSC.forward (vqvae_1d_sc.py:259-274) in train mode: rows (B*T', 512) -> rows (B*8T', 384)."""
    h = T.conv(e_rows, P[p + "aft_vq_conv.weight"], P[p + "aft_vq_conv.bias"], 0, B, L)
    h = _res_stack(P, p + "_dec_1.", h, B, L)
    for i in (2, 3, 4):
        h, L = _cnr(P, f"{p}_up_{i}.", h, B, L, "up", True)
        h = _res_stack(P, f"{p}_dec_{i}.", h, B, L)
    return T.conv(h, P[p + "project.weight"], P[p + "project.bias"], 0, B, L), L


def vqvae_train_forward(P, codebooks, x_rows, B, Tn, commitment_cost=0.25):
    """VQVAE_SC.forward(result_form='full') in train mode, without the EMA side effect (run `T.pq_ema_update` on the
    returned z / indices afterwards): -> (z rows, commitment loss, indices, pred rows, T_out)."""
    z, Tq = encoder_sc_train(P, x_rows, B, Tn)
    q, e_q_loss, idx = T.PQQuantizeTrain.apply(z, codebooks, commitment_cost)
    pred, Tout = decoder_sc_train(P, q, B, Tq)
    return z, e_q_loss, idx, pred, Tout


# ------------------------------------------------------------------------------------------- legacy VQ-VAE (s2g_body_vq)
def legacy_encoder_train(P, x_rows, B, Tn, p="encoder."):
    """Legacy Encoder.forward (nets/spg/vqvae_1d.py:66-92) in train mode: two stride-2 stages -> (z rows, T/4)."""
    h, L = _cnr(P, p + "project.", x_rows, B, Tn)
    h = _res_stack(P, p + "_enc_1.", h, B, L)
    for i in (1, 2):
        h, L = _cnr(P, f"{p}_down_{i}.", h, B, L, "down", True)
        h = _res_stack(P, f"{p}_enc_{i + 1}.", h, B, L)
    return T.conv(h, P[p + "pre_vq_conv.weight"], P[p + "pre_vq_conv.bias"], 0, B, L), L


def legacy_decoder_train(P, q_rows, B, L, p="decoder."):
    """Legacy Decoder.forward (nets/spg/vqvae_1d.py:115-144) in train mode -> (pred rows, 4 T')."""
    h = T.conv(q_rows, P[p + "aft_vq_conv.weight"], P[p + "aft_vq_conv.bias"], 0, B, L)
    h = _res_stack(P, p + "_dec_1.", h, B, L)
    for i in (2, 3):
        h, L = _cnr(P, f"{p}_up_{i}.", h, B, L, "up", True)
        h = _res_stack(P, f"{p}_dec_{i}.", h, B, L)
    return T.conv(h, P[p + "project.weight"], P[p + "project.bias"], 0, B, L), L


def legacy_vq_train_forward(P, codebook, x_rows, B, Tn, commitment_cost=0.25):
    """VQVAE.forward in train mode (nets/spg/vqvae_1d.py:163-173) without the EMA side effect: encoder, single-codebook
    nearest-code search + commitment loss + straight-through estimator (VectorQuantizerEMA, vqvae_modules.py:309-348),
    decoder.  -> (z rows, commitment loss, indices, pred rows, T_out)."""
    z, L = legacy_encoder_train(P, x_rows, B, Tn)
    q, e_q_loss, idx = T.PQQuantizeTrain.apply(z, codebook.unsqueeze(0), commitment_cost)
    pred, L = legacy_decoder_train(P, q, B, L)
    return z, e_q_loss, idx, pred, L


# ------------------------------------------------------------------------------------------- refiner
def rtransformer_train(P, x_rows, cond_rows, label, B, Tn, p_drop, seeds):
    """RTransformer.forward in train mode (RTransformer.py:209-230): x rows (B*T, 384) poses, cond rows (B*T, 800) =
    audio 768 | one-hot(mask) 2 -> rows (B*T, 384)."""
    pe = P["PE.position_encoding.weight"][1:Tn + 1].to(torch.float32).repeat(B, 1).contiguous()
    x = T.conv(x_rows, P["input_conv.weight"], P["input_conv.bias"], 0, 1, B * Tn, res=pe)
    mem = T.conv(cond_rows, P["audio_conv.weight"], P["audio_conv.bias"], 0, 1, B * Tn, res=pe)
    ce = P["class_embed.weight"]
    silu_ce = ce * torch.sigmoid(ce)
    n_layers = 0
    while f"transformer_decoder.layers.{n_layers}.norm1.weight" in P:
        n_layers += 1
    for i in range(n_layers):
        x = decoder_layer_train(P, f"transformer_decoder.layers.{i}.", x, mem, silu_ce, label, B, Tn, Tn, p_drop, seeds)
    h = T.conv(x, P["output_conv.0.weight"], P["output_conv.0.bias"], 0, 1, B * Tn, slope=0.0)
    return T.conv(h, P["output_conv.2.weight"], P["output_conv.2.bias"], 0, 1, B * Tn)


def refiner_loss(P, aud, input_poses, gt_poses, mask, label, p_drop=0.0, seeds=None, bounds=(0, 6, 96, 276, 376),
                 weights=(0.25, 0.25, 0.25, 0.25), vel_cols=276):
    """refiner.py:296-297: RTransformer forward + get_loss -> (loss, sums, pred rows)."""
    B, _, Tn = aud.shape
    x_rows = T.ncl_to_rows(input_poses)
    cond_rows = T.ncl_to_rows(torch.cat([aud, 1.0 - mask, mask], dim=1))          # F.one_hot(mask, 2): [mask==0 | mask==1]
    pred = rtransformer_train(P, x_rows, cond_rows, label.to(torch.int64).contiguous(), B, Tn, p_drop, seeds)
    nmask = (1.0 - mask).reshape(-1).contiguous()
    loss, sums = T.MaskedL1Losses.apply(pred, T.ncl_to_rows(gt_poses), nmask, float(nmask.sum()), B, Tn, list(bounds),
                                        list(weights), vel_cols)
    return loss, sums, pred
