"""CPU oracle of the predictor TRAIN step — TEST INFRASTRUCTURE, NOT PRODUCT (same rules as probtalk_oracle.py).

Restates predictor.py:232-278 on the functional oracle: AudEnc in train mode (batch-statistic BatchNorm,
vqvae_modules.py:193-200), PredictorNet train branch (predictornet.py:76-104), PTransformer.forward with
random_token('mask') (PTransformer.py:443-504,559-576), F.cross_entropy over all tokens (predictor.py:261), gradients by
torch.autograd on the CPU, clip_grad_norm_ per module (270-272) and AdamW (112-118, 278).  Dropout is not restated:
parity is checked with every dropout probability set to 0 (the reference's masks come from torch's RNG stream and
cannot be reproduced by any other implementation); the pin is tests/golden/predictor_train_b2_t88.npz, produced by
tools/make_golden_train.py from the imported reference with its Dropout modules set to p = 0.
"""
import contextlib
import math

import numpy as np

import torch
import torch.nn.functional as F

from oracle import probtalk_oracle as O


@contextlib.contextmanager
def bn_train_mode(running_out=None):
    """BatchNorm of the oracle's conv blocks uses batch statistics; updated running statistics land in running_out."""
    orig = O._bn

    def bn(sd, p, x):
        rm, rv = sd[p + "running_mean"].detach().clone(), sd[p + "running_var"].detach().clone()
        y = F.batch_norm(x, rm, rv, sd[p + "weight"], sd[p + "bias"], True, 0.1, O.BN_EPS)
        if running_out is not None:
            running_out[p + "running_mean"], running_out[p + "running_var"] = rm, rv
        return y

    O._bn = bn
    try:
        yield
    finally:
        O._bn = orig


def random_token(state, keep_scores, epoch_ratio, mask_id):
    """PTransformer.random_token('mask'), gamma 'linear' (PTransformer.py:559-582); keep_scores replaces torch.rand."""
    B = state.shape[0]
    flat = state.reshape(B, -1)
    r = math.floor((1 - epoch_ratio) * flat.shape[1])
    keep = torch.zeros_like(flat, dtype=torch.bool)
    if r > 0:
        keep.scatter_(1, keep_scores.topk(r, dim=1).indices, True)
    return (keep * flat + (~keep) * mask_id).reshape(state.shape)


def predictor_train_loss(sd_aud, sd_pred, aud, input_poses, mask, label, state, keep_scores, epoch_ratio, running_out=None):
    with bn_train_mode(running_out):
        audio = O.condition_encoder(sd_aud, aud)
    cond = O.predictor_condition(sd_pred, input_poses, mask, audio)
    ids = random_token(state, keep_scores, epoch_ratio, sd_pred["decoder.tok_emb.weight"].shape[0] - 1)
    logits = O.tokens_to_logits(sd_pred, ids, label, cond)
    return F.cross_entropy(logits.reshape(-1, logits.shape[-1]), state.reshape(-1).long()), logits


def predictor_train_grads(sd_aud, sd_pred, *args, **kw):
    """-> (loss, {('AudEnc'|'Predictor', key): grad}, running stats) with fresh leaf copies of every float parameter."""
    def leaf(sd):
        return {k: (v.detach().clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v)
                for k, v in sd.items()}
    la, lp = leaf(sd_aud), leaf(sd_pred)
    running = {}
    loss, logits = predictor_train_loss(la, lp, *args, running_out=running, **kw)
    loss.backward()
    grads = {}
    for tag, d in (("AudEnc", la), ("Predictor", lp)):
        for k, v in d.items():
            if torch.is_tensor(v) and v.requires_grad and v.grad is not None:
                grads[(tag, k)] = v.grad
    return loss.detach(), grads, running, logits.detach()


def clip_coef(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_: (total_norm, min(1, max_norm / (total_norm + 1e-6)))."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads]))
    return total, torch.clamp(max_norm / (total + 1e-6), max=1.0)


def adamw_step(p, g, m, v, step, lr, betas=(0.9, 0.99), eps=1e-8, wd=1e-2):
    """One torch.optim.AdamW update (amsgrad off), returns new (p, m, v)."""
    p = p * (1 - lr * wd)
    m = betas[0] * m + (1 - betas[0]) * g
    v = betas[1] * v + (1 - betas[1]) * g * g
    bc1, bc2 = 1 - betas[0] ** step, 1 - betas[1] ** step
    p = p - (lr / bc1) * m / (v.sqrt() / math.sqrt(bc2) + eps)
    return p, m, v


# ------------------------------------------------------------------------------------------------ PQ-VAE train step
def tile(x, K, noise):
    """ProductQuantization._tile (vqvae_modules.py:571-580): a batch of fewer chunks than codes is repeated and N(0,1) *
    0.01 / sqrt(D) is added; `noise` stands for torch.randn_like(out) (only its first K rows reach the caller)."""
    nb, D = x.shape
    if nb >= K:
        return x
    reps = (K + nb - 1) // nb
    out = x.repeat(reps, 1)
    std = 0.01 / np.sqrt(D)
    full = torch.zeros_like(out)
    full[:K] = noise                         # rows K.. of the reference's draw are never read (code_rand = out[:K])
    return out + full * std


def pq_train(sd, p, z, groups, commitment_cost=0.25, decay=0.99, eps=1e-5, counter=1, share_code=False, tile_noise=None):
    """ProductQuantization.forward, training branch (vqvae_modules.py:470-487,495-544; share_code: 496-512).
    -> (quantized with straight-through gradient (B,C,W), commitment loss, indices (BW, G), updated buffers).
    share_code: every group searches its own buffer embeddings_i (they alias one tensor only until the module is moved
    or loaded), the EMA state and embeddings_0 are the only buffers updated, and dead codes keep their EMA state."""
    B, C, W = z.shape
    D = C // groups
    x = z.permute(0, 2, 1)
    flat = x.reshape(-1, groups, D)
    K = sd[f"{p}embeddings_0"].shape[0]
    if flat.shape[0] * (groups if share_code else 1) < K:
        assert tile_noise is not None, "fewer chunks than codes: pass _tile's N(0,1) draws ([G,K,D], or [K,D] with share_code)"
    idx, q, new = [], [], {}
    for i in range(groups):
        emb = sd[f"{p}embeddings_{i}"]
        ind = torch.argmin(O.code_distances(flat[:, i].detach(), emb), dim=1)
        idx.append(ind)
        q.append(F.embedding(ind, emb).reshape(B, W, D))
    quantized = torch.cat(q, dim=-1)
    with torch.no_grad():
        corr = 1 - decay ** counter
        if share_code:
            all_idx = torch.stack(idx, dim=0).reshape(-1)                           # line 497: group-major
            all_x = flat.detach().permute(1, 0, 2).reshape(-1, D)                   # line 498
            enc = F.one_hot(all_idx, K).to(flat.dtype)
            cs_h = sd[f"{p}ema_cluster_size.hidden"].clone()
            cs_h.sub_((cs_h - enc.sum(0)) * (1 - decay))
            cs = cs_h / corr
            n = cs.sum()
            cs = (cs + eps) / (n + K * eps) * n
            dw_h = sd[f"{p}ema_dw.hidden"].clone()
            dw_h.sub_((dw_h - enc.t() @ all_x) * (1 - decay))
            normalised = (dw_h / corr) / cs.reshape(-1, 1)
            usage = (cs.view(K, 1) >= 1.0).float()
            new[f"{p}ema_dw.hidden"] = dw_h
            new[f"{p}ema_cluster_size.hidden"] = cs_h
            new[f"{p}embeddings_0"] = usage * normalised + (1 - usage) * tile(all_x, K, tile_noise)[:K]
        for i in range(groups if not share_code else 0):
            enc = F.one_hot(idx[i], K).to(flat.dtype)
            cs_h = sd[f"{p}ema_cluster_size.{i}.hidden"].clone()
            cs_h.sub_((cs_h - enc.sum(0)) * (1 - decay))
            cs = cs_h / corr
            n = cs.sum()
            cs = (cs + eps) / (n + K * eps) * n
            dw_h = sd[f"{p}ema_dw.{i}.hidden"].clone()
            dw_h.sub_((dw_h - enc.t() @ flat[:, i].detach()) * (1 - decay))
            normalised = (dw_h / corr) / cs.reshape(-1, 1)
            code_rand = tile(flat[:, i].detach(), K, None if tile_noise is None else tile_noise[i])[:K]
            usage = (cs.view(K, 1) >= 1.0).float()
            new[f"{p}ema_dw.{i}.hidden"] = usage * dw_h + (1 - usage) * code_rand
            new[f"{p}ema_cluster_size.{i}.hidden"] = cs_h * usage.view(-1) + (1 - usage).view(-1)
            new[f"{p}embeddings_{i}"] = usage * normalised + (1 - usage) * code_rand
    loss = commitment_cost * F.mse_loss(x, quantized.detach())
    quantized = x + (quantized - x).detach()
    return quantized.permute(0, 2, 1), loss, torch.stack(idx, dim=1), new


def vq_losses(pred, gt, dim_list=(0, 6, 6, 96, 276), exp_dim=100):
    """vq_teacher.get_loss, type 'fbhe' (vq_teacher.py:191-203,222-229), pred/gt (B,T,C) -> (rec + velocity, parts)."""
    d = dim_list
    sl1 = lambda a, b: F.smooth_l1_loss(a, b, beta=0.01)
    face = sl1(pred[..., :d[1]], gt[..., :d[1]])
    body = sl1(pred[..., d[1]:d[3]], gt[..., d[1]:d[3]])
    hand = sl1(pred[..., d[3]:d[4]], gt[..., d[3]:d[4]])
    exp = sl1(pred[..., d[4]:], gt[..., d[4]:])
    weight = 0.4
    rec = (face + exp) * (0.5 - weight) + (body + hand) * weight
    v_pr = pred[:, 1:, :-exp_dim] - pred[:, :-1, :-exp_dim]
    v_gt = gt[:, 1:, :-exp_dim] - gt[:, :-1, :-exp_dim]
    vel = torch.mean(torch.abs(v_pr - v_gt))
    return rec + vel, {"face_loss": face, "body_loss": body, "hand_loss": hand, "exp_loss": exp, "velocity_loss": vel}


def vq_train_grads(sd, x, groups=4, counter=1):
    """One vq_teacher step up to the gradients (vq_teacher.py:167-177): -> (loss, parts, grads, new buffers, pred)."""
    leaf = {k: (v.detach().clone().requires_grad_(True)
                if v.dtype == torch.float32 and "running" not in k and "embeddings_" not in k and "hidden" not in k else v)
            for k, v in sd.items()}
    running = {}
    with bn_train_mode(running):
        z = O.encoder_sc(leaf, "encoder.", x)
        e, e_q_loss, idx, new = pq_train(leaf, "vq_layer.", z, groups, counter=counter)
        pred = O.decoder_sc(leaf, "decoder.", e)
    rec, parts = vq_losses(pred.transpose(1, 2), x.transpose(1, 2))
    loss = rec + e_q_loss
    loss.backward()
    parts["e_q_loss"] = e_q_loss.detach()
    grads = {k: v.grad for k, v in leaf.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None}
    new.update(running)
    return loss.detach(), parts, grads, new, pred.detach(), idx


# ------------------------------------------------------------------------------------------------ refiner train step
def refiner_losses(pred, gt, mask, dim_list=(0, 6, 6, 96, 276), each_dim=(6, 90, 180, 100)):
    """refiner.get_loss, type 'fbhe' (refiner.py:321-349,371-391): pred/gt (B,C,T), mask (B,1,T)."""
    d = dim_list
    nmask = 1 - mask
    t_nmask = nmask.sum()
    part = lambda lo, hi, n: torch.abs(pred[:, lo:hi] * nmask - gt[:, lo:hi] * nmask).sum() / (t_nmask * n)
    face, body = part(0, d[1], each_dim[0]), part(d[1], d[3], each_dim[1])
    hand, exp = part(d[3], d[4], each_dim[2]), part(d[4], pred.shape[1], each_dim[3])
    rec = (face + body + hand + exp) / 4
    v_pr = pred[:, :-each_dim[3], 1:] - pred[:, :-each_dim[3], :-1]
    v_gt = gt[:, :-each_dim[3], 1:] - gt[:, :-each_dim[3], :-1]
    vel = torch.mean(torch.abs(v_pr - v_gt))
    return rec + vel, {"face_loss": face, "body_loss": body, "hand_loss": hand, "exp_loss": exp, "rec_loss": rec,
                       "velocity_loss": vel}


def refiner_train_grads(sd, aud, input_poses, gt_poses, mask, label):
    """refiner.py:296-300 with dropout off: -> (loss, parts, {key: grad}, pred)."""
    leaf = {k: (v.detach().clone().requires_grad_(True) if v.dtype == torch.float32 else v) for k, v in sd.items()}
    pred = O.rtransformer(leaf, input_poses, aud, mask, label)
    loss, parts = refiner_losses(pred, gt_poses, mask)
    loss.backward()
    grads = {k: v.grad for k, v in leaf.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None}
    return loss.detach(), parts, grads, pred.detach()


# ------------------------------------------------------------------------------------------------ legacy VQ train step
def vq_ema_train(sd, p, z, commitment_cost=0.25, decay=0.99, eps=1e-5, counter=1):
    """VectorQuantizerEMA.forward, training branch (vqvae_modules.py:309-348): single codebook, EMA update with Laplace
    smoothing and NO dead-code re-seeding.  -> (quantized with straight-through gradient (B,D,W), loss, indices, new buffers)."""
    B, D, W = z.shape
    x = z.permute(0, 2, 1)
    flat = x.reshape(-1, D)
    emb = sd[p + "embeddings"]
    K = emb.shape[0]
    ind = torch.argmin(O.code_distances(flat.detach(), emb), dim=1)
    quantized = F.embedding(ind, emb).reshape(B, W, D)
    with torch.no_grad():
        corr = 1 - decay ** counter
        enc = F.one_hot(ind, K).to(flat.dtype)
        cs_h = sd[p + "ema_cluster_size.hidden"].clone()
        cs_h.sub_((cs_h - enc.sum(0)) * (1 - decay))
        cs = cs_h / corr
        n = cs.sum()
        cs = (cs + eps) / (n + K * eps) * n
        dw_h = sd[p + "ema_dw.hidden"].clone()
        dw_h.sub_((dw_h - enc.t() @ flat.detach()) * (1 - decay))
        new = {p + "ema_cluster_size.hidden": cs_h, p + "ema_dw.hidden": dw_h, p + "embeddings": (dw_h / corr) / cs.reshape(-1, 1)}
    loss = commitment_cost * F.mse_loss(x, quantized.detach())
    quantized = x + (quantized - x).detach()
    return quantized.permute(0, 2, 1), loss, ind, new


def legacy_vq_losses(pred, gt):
    """smplx_body_vq.TrainWrapper.get_loss (smplx_body_vq.py:170-190) without the commitment term, pred/gt (B,C,T)."""
    rec = torch.mean(torch.abs(pred - gt))
    vel = torch.mean(torch.abs((pred[:, :, 1:] - pred[:, :, :-1]) - (gt[:, :, 1:] - gt[:, :, :-1])))
    return rec + vel, {"rec_loss": rec, "velocity_loss": vel}


def legacy_vq_train_grads(sd, x, counter=1):
    """One s2g_body_vq step up to the gradients (smplx_body_vq.py:146-163; VQVAE.forward train, vqvae_1d.py:163-173):
    -> (loss, parts, grads, new buffers incl. BatchNorm running statistics, pred, indices)."""
    leaf = {k: (v.detach().clone().requires_grad_(True)
                if v.dtype == torch.float32 and "running" not in k and "embeddings" not in k and "hidden" not in k else v)
            for k, v in sd.items()}
    running = {}
    with bn_train_mode(running):
        z = O.legacy_encoder(leaf, "encoder.", x)
        e, e_q_loss, idx, new = vq_ema_train(leaf, "vq_layer.", z, counter=counter)
        pred = O.legacy_decoder(leaf, "decoder.", e)
    rec, parts = legacy_vq_losses(pred, x)
    loss = rec + e_q_loss
    loss.backward()
    parts["e_q_loss"] = e_q_loss.detach()
    grads = {k: v.grad for k, v in leaf.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None}
    new.update(running)
    return loss.detach(), parts, grads, new, pred.detach(), idx
