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
