"""nn.Module mirror of the reference's hot-path networks, computing on libprobtalk_hip.so.

Every class keeps the reference's attribute names, constructor arguments and
`state_dict()` keys (checked against tests/golden/state_dict_manifest.json), so
checkpoints and callers carry over; `forward`/`encode`/`decode`/... enqueue the
hand-written HIP kernels through the C ABI.  The torch sub-modules below
(nn.Conv1d, nn.BatchNorm1d, nn.MultiheadAttention ...) are parameter containers
only: their own forward is never called.

Reference classes mirrored (relative to /root/reference):
  nets/spg/vqvae_modules.py:24-29,107-276,279-377,437-590
  nets/inpainting/vqvae_1d_sc.py:70-91,197-324     nets/spg/vqvae_1d.py:66-187
  nets/inpainting/gated_pixelcnn_1d.py:39-83       nets/inpainting/predictornet.py:13-112
  nets/inpainting/PTransformer.py:28-165,214-372,394-704
  nets/inpainting/RTransformer.py:181-230
In eval mode `forward`/`encode`/`decode`/... run the packed inference sequences of the library.  In train mode
(`module.train()`) the same calls run the training kernels through `probtalk_amd.train_ops` autograd Functions
(batch-statistic BatchNorm, Dropout, EMA codebook updates, straight-through estimator), so `loss.backward()` works on
the module API as it does on the reference's modules (probtalk_amd/training.py holds the sequences).
"""
import ctypes as C
import math
import os

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from .._lib import check, ptr, stream


def _seeds_of(mod):
    """The module's dropout seed stream (train_ops._Seeds); a wrapper may install a shared one as `_pt_seeds`."""
    s = mod.__dict__.get("_pt_seeds")
    if s is None:
        from ..train_ops import _Seeds
        s = mod.__dict__["_pt_seeds"] = _Seeds(0)
    return s


def _train_tensors(mod):
    """name -> parameter / buffer for the training kernels, which read them in place: they must live on the GPU."""
    from ..training import tensors_of
    P = tensors_of(mod)
    for k, v in P.items():
        if not v.is_cuda:
            raise RuntimeError(f"probtalk_amd: train mode reads the module's parameters in place; call .cuda() first "
                               f"({type(mod).__name__}.{k} is on {v.device})")
    return P


def _cuda_keep_grad(t):
    if not t.is_cuda:
        raise RuntimeError("probtalk_amd: inputs must be CUDA (HIP) tensors")
    return t.to(torch.float32)


def _cuda_f32(t):
    if not t.is_cuda:
        raise RuntimeError("probtalk_amd: inputs must be CUDA (HIP) tensors")
    return t.detach().to(torch.float32).contiguous()


class QuantizationOutput():
    """vqvae_modules.py:24-29 (eval mode: code indices travel in `.loss`)."""

    def __init__(self, quantized, loss, q_loss=0, lr_loss=0):
        self.quantized = quantized
        self.loss = loss
        self.q_loss = q_loss
        self.lr_loss = lr_loss


# ---------------------------------------------------------------------------
# parameter containers for the conv blocks
# ---------------------------------------------------------------------------
class ConvNormRelu(nn.Module):
    def __init__(self, in_channels, out_channels, leaky=False, slope=0.2, sample='none', p=0, groups=1,
                 residual=False, norm='bn'):
        super().__init__()
        assert norm == 'bn' and groups == 1, "only the BatchNorm, groups=1 form is on the hot path"
        self.residual = residual
        self.sample = sample
        self.slope = slope if leaky else 0.0
        padding = 1
        if sample == 'none':
            kernel_size, stride = 3, 1
        elif sample == 'one':
            padding, kernel_size, stride = 0, 1, 1
        else:
            kernel_size, stride = 4, 2
        conv = nn.ConvTranspose1d if sample == 'up' else nn.Conv1d
        if residual:
            if sample in ('down', 'up'):
                self.residual_layer = conv(in_channels, out_channels, kernel_size, stride, padding)
            elif in_channels == out_channels:
                self.residual_layer = nn.Identity()
            else:
                self.residual_layer = nn.Sequential(nn.Conv1d(in_channels, out_channels, kernel_size, stride, padding))
        self.conv = conv(in_channels, out_channels, kernel_size, stride, padding)
        self.norm = nn.BatchNorm1d(out_channels)
        self.dropout = nn.Dropout(p=p)
        self.relu = nn.LeakyReLU(negative_slope=slope) if leaky else nn.ReLU()

    def forward(self, x, **kwargs):
        raise NotImplementedError("ConvNormRelu is fused into its parent module's HIP sequence")


class Res_CNR_Stack(nn.Module):
    def __init__(self, channels, layers, sample='none', leaky=False, casual=False, leaky_out=False, norm='bn'):
        super().__init__()
        assert not casual and sample == 'none' and not leaky_out
        self._layers = nn.ModuleList(
            [ConvNormRelu(channels, channels, leaky=leaky, sample=sample, norm=norm) for _ in range(layers)])
        self.conv = nn.Conv1d(channels, channels, 3, 1, 1)
        self.norm = nn.BatchNorm1d(channels)
        self.relu = nn.ReLU()

    def forward(self, x, pre_state=None):
        raise NotImplementedError("Res_CNR_Stack is fused into its parent module's HIP sequence")


class ExponentialMovingAverage(nn.Module):
    def __init__(self, init_value, decay):
        super().__init__()
        self.decay = decay
        self.counter = 0
        self.register_buffer("hidden", torch.zeros_like(init_value))


class _HandleOwner:
    """Lazy creation of the packed device model from the module's current state_dict."""

    _kind = None
    _bf16_ok = False

    def _cfg(self):
        raise NotImplementedError

    def _handle(self):
        h = self.__dict__.get("_pt_handle")
        if h is None:
            h = _lib.Handle(self._kind, self._cfg(), self.state_dict(),
                            precision=self.__dict__.get("_pt_precision", "fp32"))
            self.__dict__["_pt_handle"] = h
        return h

    def set_precision(self, precision):
        """'fp32' (default, parity path) or 'bf16' (inference only: bf16 MFMA operands, fp32 accumulation and LayerNorm
        statistics; the transformers' residual stream is bf16 too unless option resid_bf16 = 0; the PQ-VAE keeps
        encode + nearest-code search in fp32).  Train mode always computes in fp32."""
        if precision not in ("fp32", "bf16"):
            raise ValueError(f"precision must be 'fp32' or 'bf16', got {precision!r}")
        if precision == "bf16" and not self._bf16_ok:
            raise NotImplementedError(f"{type(self).__name__} has no bf16 mode")
        if precision != self.__dict__.get("_pt_precision", "fp32"):
            self.__dict__["_pt_precision"] = precision
            self.__dict__["_pt_handle"] = None
        return self

    def invalidate(self):
        """Drop the packed weights (call after mutating parameters in place)."""
        self.__dict__["_pt_handle"] = None

    def _load_from_state_dict(self, *a, **k):
        self.__dict__["_pt_handle"] = None
        return super()._load_from_state_dict(*a, **k)


# ---------------------------------------------------------------------------
# quantizers
# ---------------------------------------------------------------------------
class _QuantizerBase(nn.Module):
    def _codebooks(self, device):
        cache = self.__dict__.get("_pt_cb")
        if cache is None or cache.device != device:
            names = self._codebook_names()
            cache = torch.stack([getattr(self, n).detach().to(torch.float32) for n in names]).to(device).contiguous()
            self.__dict__["_pt_cb"] = cache
        return cache

    def _load_from_state_dict(self, *a, **k):
        self.__dict__["_pt_cb"] = None
        return super()._load_from_state_dict(*a, **k)

    def _search(self, x):
        """x (B,C,W) -> quantized (B,C,W), indices (B,W,G) int64 via pt_pq_search_cl."""
        x = _cuda_f32(x)
        B, Cc, W = x.shape
        cb = self._codebooks(x.device)
        G, K, D = cb.shape
        if Cc != G * D:
            raise RuntimeError(f"quantizer: input has {Cc} channels, codebooks cover {G}x{D}")
        L = _lib.lib()
        z_cl = torch.empty(B * W, Cc, device=x.device)
        q_cl = torch.empty(B * W, Cc, device=x.device)
        idx = torch.empty(B, W, G, dtype=torch.int64, device=x.device)
        q = torch.empty(B, Cc, W, device=x.device)
        check(L.pt_ncl_to_cl(ptr(x), ptr(z_cl), B, Cc, W, Cc, stream()))
        check(L.pt_pq_search_cl(ptr(z_cl), Cc, ptr(cb), B * W, G, D, K, ptr(idx), ptr(q_cl), Cc, stream()))
        check(L.pt_cl_to_ncl(ptr(q_cl), Cc, ptr(q), B, Cc, W, stream()))
        return q, idx


class ProductQuantization(_QuantizerBase):
    """vqvae_modules.py:437-590: eval path (indices in `.loss`) and train path (EMA update, commitment loss in `.loss`,
    straight-through estimator)."""

    def __init__(self, embedding_dim, num_embeddings, commitment_cost, decay, epsilon=1e-5, num_chunks=8,
                 share_code=False, **kwargs):
        super().__init__()
        self.num_embeddings = num_embeddings
        self.commitment_cost = commitment_cost
        self.epsilon = epsilon
        self.num_chunks = num_chunks
        self.share_code = share_code
        self.embedding_dim = embedding_dim // num_chunks
        self.ema_dw = nn.ModuleList()
        self.ema_cluster_size = nn.ModuleList()
        if share_code:
            # vqvae_modules.py:452-459: ONE tensor registered under all G buffer names (G state_dict entries), one EMA
            # pair.  As in the reference, the names stop aliasing once the module is moved to a device or loaded from a
            # checkpoint (nn.Module handles every buffer name on its own); training then updates embeddings_0 only
            # (line 512) while group i keeps searching embeddings_i.
            emb = torch.empty(num_embeddings, self.embedding_dim)
            nn.init.xavier_uniform_(emb)
            for i in range(num_chunks):
                self.register_buffer("embeddings_{}".format(i), emb)
            self.ema_dw = ExponentialMovingAverage(emb, decay)
            self.ema_cluster_size = ExponentialMovingAverage(torch.zeros((num_embeddings,)), decay)
            return
        for i in range(num_chunks):
            emb = torch.empty(num_embeddings, self.embedding_dim)
            nn.init.xavier_uniform_(emb)
            self.register_buffer("embeddings_{}".format(i), emb)
            self.ema_dw.append(ExponentialMovingAverage(emb, decay))
            self.ema_cluster_size.append(ExponentialMovingAverage(torch.zeros((num_embeddings,)), decay))

    def _codebook_names(self):
        return [f"embeddings_{i}" for i in range(self.num_chunks)]

    def forward(self, x, r=0., **kwargs):
        if r != 0.:
            raise NotImplementedError("random token replacement (r>0) is off in every config of the path (vq.json)")
        if self.training:
            return self._forward_train(x)
        q, idx = self._search(x)
        return QuantizationOutput(q, idx)

    def _forward_train(self, x):
        """vqvae_modules.py:470-544: nearest codes, EMA codebook update with dead-code re-seeding (no gradient; the
        returned codes are the OLD ones, as in the reference), commitment loss, straight-through estimator."""
        from .. import train_ops as T
        x = _cuda_keep_grad(x)
        B, Cc, W = x.shape
        G = self.num_chunks
        if Cc != G * self.embedding_dim:
            raise RuntimeError(f"quantizer: input has {Cc} channels, codebooks cover {G}x{self.embedding_dim}")
        _train_tensors(self)
        # `_tile`'s N(0,1) draws for a batch with fewer chunks than codes (vqvae_modules.py:573-577): tests inject them
        # ([G,K,D], or [K,D] with share_code) as they inject the sampler's noise; otherwise the kernel draws its own
        tile_noise = self.__dict__.pop("_pt_tile_noise", None)
        cb = torch.stack([getattr(self, n) for n in self._codebook_names()]).contiguous()
        z = T.to_rows(x, Cc)
        q, loss, idx = T.PQQuantizeTrain.apply(z, cb, self.commitment_cost)
        if self.share_code:
            self.ema_cluster_size.counter += 1
            self.ema_dw.counter += 1
            T.pq_ema_update_shared(z.detach(), idx, G, self.ema_cluster_size.hidden, self.ema_dw.hidden, self.embeddings_0,
                                   self.ema_dw.decay, self.epsilon, self.ema_dw.counter, tile_noise=tile_noise)
            self.__dict__["_pt_cb"] = None
            self.__dict__["_pt_last_indices"] = idx.view(B, W, G)
            return QuantizationOutput(T.RowsToNCL.apply(q, B, Cc, W), loss)
        cs = torch.stack([m.hidden for m in self.ema_cluster_size]).contiguous()
        dw = torch.stack([m.hidden for m in self.ema_dw]).contiguous()
        for m in list(self.ema_cluster_size) + list(self.ema_dw):
            m.counter += 1
        T.pq_ema_update(z.detach(), idx, cs, dw, cb, self.ema_dw[0].decay, self.epsilon, self.ema_dw[0].counter,
                        tile_noise=tile_noise)
        with torch.no_grad():
            for i in range(G):
                self.ema_cluster_size[i].hidden.copy_(cs[i])
                self.ema_dw[i].hidden.copy_(dw[i])
                getattr(self, f"embeddings_{i}").copy_(cb[i])
        self.__dict__["_pt_cb"] = None
        self.__dict__["_pt_last_indices"] = idx.view(B, W, G)        # not part of the reference's return value
        return QuantizationOutput(T.RowsToNCL.apply(q, B, Cc, W), loss)

    def quantize_all(self, encoding_indices):
        """(..., G) int -> (..., G*D)  (vqvae_modules.py:557-564)."""
        if not encoding_indices.is_cuda:
            raise RuntimeError("probtalk_amd: inputs must be CUDA (HIP) tensors")
        cb = self._codebooks(encoding_indices.device)
        G, K, D = cb.shape
        idx = encoding_indices
        if idx.dtype not in (torch.int64, torch.int32):
            idx = idx.to(torch.int64)
        idx = idx.contiguous()
        if idx.shape[-1] != G:
            raise RuntimeError(f"quantize_all: last dim {idx.shape[-1]} != groups {G}")
        rows = idx.numel() // G
        out = torch.empty(*idx.shape[:-1], G * D, device=idx.device)
        check(_lib.lib().pt_pq_gather_cl(ptr(idx), int(idx.dtype == torch.int64), ptr(cb), rows, G, D, K, ptr(out),
                                         G * D, stream()))
        return out

    def quantize(self, encoding_indices, i):
        cb = self._codebooks(encoding_indices.device)
        G, K, D = cb.shape
        idx = encoding_indices.to(torch.int64).contiguous()
        out = torch.empty(*idx.shape, D, device=idx.device)
        check(_lib.lib().pt_pq_gather_cl(ptr(idx), 1, ptr(cb[i]), idx.numel(), 1, D, K, ptr(out), D, stream()))
        return out

    def get_code_indices(self, flat_x, i):
        """flat_x (N,D) -> (N,) int64 for sub-codebook i (vqvae_modules.py:546-555)."""
        x = _cuda_f32(flat_x)
        cb = self._codebooks(x.device)
        G, K, D = cb.shape
        idx = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
        check(_lib.lib().pt_pq_search_cl(ptr(x), D, ptr(cb[i]), x.shape[0], 1, D, K, ptr(idx), None, D, stream()))
        return idx


class VectorQuantizerEMA(_QuantizerBase):
    """vqvae_modules.py:279-377 (legacy single codebook): eval -> (quantized, indices); train -> (quantized with the
    straight-through estimator, commitment loss) after the EMA codebook update."""

    def __init__(self, embedding_dim, num_embeddings, commitment_cost, decay, epsilon=1e-5):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.num_embeddings = num_embeddings
        self.commitment_cost = commitment_cost
        self.epsilon = epsilon
        emb = torch.empty(num_embeddings, embedding_dim)
        nn.init.xavier_uniform_(emb)
        self.register_buffer("embeddings", emb)
        self.ema_dw = ExponentialMovingAverage(self.embeddings, decay)
        self.ema_cluster_size = ExponentialMovingAverage(torch.zeros((num_embeddings,)), decay)

    def _codebook_names(self):
        return ["embeddings"]

    def forward(self, x):
        if self.training:
            from .. import train_ops as T
            x = _cuda_keep_grad(x)
            B, Cc, W = x.shape
            _train_tensors(self)
            z = T.to_rows(x, Cc)
            q, loss, idx = T.PQQuantizeTrain.apply(z, self.embeddings.unsqueeze(0), self.commitment_cost)
            self.ema_cluster_size.counter += 1
            self.ema_dw.counter += 1
            T.vq_ema_update(z.detach(), idx, self.ema_cluster_size.hidden, self.ema_dw.hidden, self.embeddings,
                            self.ema_dw.decay, self.epsilon, self.ema_dw.counter)
            self.__dict__["_pt_cb"] = None
            return T.RowsToNCL.apply(q, B, Cc, W), loss
        q, idx = self._search(x)
        return q, idx.reshape(q.shape[0], q.shape[2])

    def quantize(self, encoding_indices):
        cb = self._codebooks(encoding_indices.device)
        _, K, D = cb.shape
        idx = encoding_indices.to(torch.int64).contiguous()
        out = torch.empty(*idx.shape, D, device=idx.device)
        check(_lib.lib().pt_pq_gather_cl(ptr(idx), 1, ptr(cb), idx.numel(), 1, D, K, ptr(out), D, stream()))
        return out

    def quantize_all(self, encoding_indices):
        return self.quantize(encoding_indices[:, :, 0])


# ---------------------------------------------------------------------------
# PQ-VAE (vqvae_1d_sc.py:197-324) and legacy VQ-VAE (vqvae_1d.py:66-187)
# ---------------------------------------------------------------------------
class EncoderSC(nn.Module):
    def __init__(self, in_dim, embedding_dim, num_hiddens, num_residual_layers, num_residual_hiddens):
        super().__init__()
        nh, n = num_hiddens, num_residual_layers
        self.project = ConvNormRelu(in_dim, nh // 4, leaky=True)
        self._enc_1 = Res_CNR_Stack(nh // 4, n, leaky=True)
        self._down_1 = ConvNormRelu(nh // 4, nh // 4, leaky=True, residual=True, sample='down')
        self._enc_2 = Res_CNR_Stack(nh // 4, n, leaky=True)
        self._down_2 = ConvNormRelu(nh // 4, nh // 2, leaky=True, residual=True, sample='down')
        self._enc_3 = Res_CNR_Stack(nh // 2, n, leaky=True)
        self._down_3 = ConvNormRelu(nh // 2, nh, leaky=True, residual=True, sample='down')
        self._enc_4 = Res_CNR_Stack(nh, n, leaky=True)
        self.pre_vq_conv = nn.Conv1d(nh, embedding_dim, 1, 1)


class DecoderSC(nn.Module):
    def __init__(self, out_dim, embedding_dim, num_hiddens, num_residual_layers, num_residual_hiddens, ae=False):
        super().__init__()
        nh, n = num_hiddens, num_residual_layers
        self.aft_vq_conv = nn.Conv1d(embedding_dim, nh, 1, 1)
        self._dec_1 = Res_CNR_Stack(nh, n, leaky=True)
        self._up_2 = ConvNormRelu(nh, nh // 2, leaky=True, residual=True, sample='up')
        self._dec_2 = Res_CNR_Stack(nh // 2, n, leaky=True)
        self._up_3 = ConvNormRelu(nh // 2, nh // 4, leaky=True, residual=True, sample='up')
        self._dec_3 = Res_CNR_Stack(nh // 4, n, leaky=True)
        self._up_4 = ConvNormRelu(nh // 4, nh // 4, leaky=True, residual=True, sample='up')
        self._dec_4 = Res_CNR_Stack(nh // 4, n, leaky=True)
        self.project = nn.Conv1d(nh // 4, out_dim, 1, 1)


class _VQBase(_HandleOwner, nn.Module):
    _levels = 4

    def _latent_len(self, T):
        L = T
        for _ in range(self._levels - 1):
            L = (L - 2) // 2 + 1
        return L

    def _encode_train(self, x):
        """Encoder in train mode (batch-statistic BatchNorm, running statistics updated): (B,C,T) -> z (B,emb,T')."""
        from .. import train_ops as T, training
        x = _cuda_keep_grad(x)
        B, Cc, Tn = x.shape
        if Cc != self.in_dim:
            raise RuntimeError(f"expected {self.in_dim} pose channels, got {Cc}")
        enc = training.encoder_sc_train if self._levels == 4 else training.legacy_encoder_train
        z_rows, Tq = enc(_train_tensors(self), T.to_rows(x), B, Tn)
        self.__dict__["_pt_handle"] = None          # BatchNorm statistics moved: the packed inference weights are stale
        return T.RowsToNCL.apply(z_rows, B, self.embedding_dim, Tq)

    def _decode_train(self, e):
        from .. import train_ops as T, training
        e = _cuda_keep_grad(e)
        B, Cc, Tq = e.shape
        dec = training.decoder_sc_train if self._levels == 4 else training.legacy_decoder_train
        rows, Tout = dec(_train_tensors(self), T.to_rows(e, Cc), B, Tq)
        self.__dict__["_pt_handle"] = None
        return T.RowsToNCL.apply(rows, B, self.in_dim, Tout)

    def _encode_raw(self, x, want_z=True):
        x = _cuda_f32(x)
        B, Cc, T = x.shape
        if Cc != self.in_dim:
            raise RuntimeError(f"expected {self.in_dim} pose channels, got {Cc}")
        h = self._handle()
        L = _lib.lib()
        Tq = self._latent_len(T)
        G = self._groups()
        z = torch.empty(B, self.embedding_dim, Tq, device=x.device) if want_z else None
        q = torch.empty(B, self.embedding_dim, Tq, device=x.device)
        idx = torch.empty(B, Tq, G, dtype=torch.int64, device=x.device)
        nb = L.pt_vqvae_workspace_bytes(h.h, B, T)
        ws = h.workspace(nb)
        check(L.pt_vqvae_encode(h.h, ptr(x), B, T, ptr(z), ptr(q), ptr(idx), ptr(ws), ws.numel(), stream()))
        return z, q, idx

    def _decode_raw(self, e):
        if self.training:
            return self._decode_train(e)
        e = _cuda_f32(e)
        B, Cc, Tq = e.shape
        h = self._handle()
        L = _lib.lib()
        T = Tq << (self._levels - 1)
        out = torch.empty(B, self.in_dim, T, device=e.device)
        ws = h.workspace(L.pt_vqvae_workspace_bytes(h.h, B, T))
        check(L.pt_vqvae_decode(h.h, ptr(e), B, Tq, ptr(out), ptr(ws), ws.numel(), stream()))
        return out

    def decode_codes(self, codes):
        """quantize_all + transpose + decode in one call (predictor.py:439-442): codes (B,T',G) -> (B,C,8T')."""
        if self.training:
            q = self.vq_layer.quantize_all(codes)
            return self._decode_train(q.reshape(codes.shape[0], codes.shape[1], -1).transpose(1, 2))
        if codes.dtype not in (torch.int32, torch.int64):
            codes = codes.to(torch.int64)
        codes = codes.contiguous()
        B, Tq = codes.shape[0], codes.shape[1]
        h = self._handle()
        L = _lib.lib()
        T = Tq << (self._levels - 1)
        out = torch.empty(B, self.in_dim, T, device=codes.device)
        ws = h.workspace(L.pt_vqvae_workspace_bytes(h.h, B, T))
        check(L.pt_vqvae_decode_codes(h.h, ptr(codes), int(codes.dtype == torch.int64), B, Tq, ptr(out), ptr(ws),
                                      ws.numel(), stream()))
        return out


class VQVAE_SC(_VQBase):
    """vqvae_1d_sc.py:277-324.  bf16 mode (inference): the conv DECODER runs on bf16 MFMA; encode + code search stay
    fp32.  Train mode: encode / decode / forward(result_form='full') run the training kernels with autograd."""
    _kind = "vqvae_sc"
    _levels = 4
    _bf16_ok = True

    def __init__(self, in_dim, embedding_dim, num_embeddings, num_hiddens, num_residual_layers, num_residual_hiddens,
                 commitment_cost=0.25, decay=0.99, q_type='pro', groups=1, share_code=False):
        super().__init__()
        if q_type != 'pro':
            raise NotImplementedError("q_type='res' is not on the configured path (config/vq.json:46)")
        if num_residual_layers != 2:
            raise NotImplementedError("the HIP sequence is built for num_residual_layers=2 (vq_teacher.py:66)")
        self.in_dim = in_dim
        self.embedding_dim = embedding_dim
        self.num_embeddings = num_embeddings
        self.num_hiddens = num_hiddens
        self.q_type = q_type
        self.encoder = EncoderSC(in_dim, embedding_dim, num_hiddens, num_residual_layers, num_residual_hiddens)
        self.vq_layer = ProductQuantization(embedding_dim, num_embeddings, commitment_cost, decay, num_chunks=groups,
                                            share_code=share_code)
        self.decoder = DecoderSC(in_dim, embedding_dim, num_hiddens, num_residual_layers, num_residual_hiddens)

    def _groups(self):
        return self.vq_layer.num_chunks

    def _cfg(self):
        return [self.in_dim, self.embedding_dim, self.num_embeddings, self.num_hiddens, self.vq_layer.num_chunks, 4]

    def encode(self, gt_poses, r=0., **kwargs):
        if r != 0.:
            raise NotImplementedError("random token replacement (r>0) is off in every config of the path (vq.json)")
        if self.training:
            z = self._encode_train(gt_poses)
            return z, self.vq_layer(z, r), {1: 0, 2: 0, 3: 0, 4: 0}
        z, q, idx = self._encode_raw(gt_poses)
        # the reference also returns the per-level encoder features; every caller zeroes or ignores them
        return z, QuantizationOutput(q, idx), {1: 0, 2: 0, 3: 0, 4: 0}

    def decode(self, e, enc_feats=None):
        return self._decode_raw(e)

    def forward(self, gt_poses, result_form='pose', r=0., **kwargs):
        z, qo, enc_feats = self.encode(gt_poses, r, **kwargs)
        pred_poses = self.decode(qo.quantized, enc_feats)
        if result_form == 'full':
            return z, qo, pred_poses
        elif result_form == 'part':
            return qo.loss, pred_poses
        return pred_poses


class _LegacyEncoder(nn.Module):
    def __init__(self, in_dim, embedding_dim, nh, n):
        super().__init__()
        self.project = ConvNormRelu(in_dim, nh // 4, leaky=True)
        self._enc_1 = Res_CNR_Stack(nh // 4, n, leaky=True)
        self._down_1 = ConvNormRelu(nh // 4, nh // 2, leaky=True, residual=True, sample='down')
        self._enc_2 = Res_CNR_Stack(nh // 2, n, leaky=True)
        self._down_2 = ConvNormRelu(nh // 2, nh, leaky=True, residual=True, sample='down')
        self._enc_3 = Res_CNR_Stack(nh, n, leaky=True)
        self.pre_vq_conv = nn.Conv1d(nh, embedding_dim, 1, 1)


class _LegacyDecoder(nn.Module):
    def __init__(self, out_dim, embedding_dim, nh, n):
        super().__init__()
        self.aft_vq_conv = nn.Conv1d(embedding_dim, nh, 1, 1)
        self._dec_1 = Res_CNR_Stack(nh, n, leaky=True)
        self._up_2 = ConvNormRelu(nh, nh // 2, leaky=True, residual=True, sample='up')
        self._dec_2 = Res_CNR_Stack(nh // 2, n, leaky=True)
        self._up_3 = ConvNormRelu(nh // 2, nh // 4, leaky=True, residual=True, sample='up')
        self._dec_3 = Res_CNR_Stack(nh // 4, n, leaky=True)
        self.project = nn.Conv1d(nh // 4, out_dim, 1, 1)


class VQVAE(_VQBase):
    """Legacy single-codebook VQ-VAE (vqvae_1d.py:147-187), the model behind `s2g_body_vq`."""
    _kind = "vqvae_legacy"
    _levels = 3

    def __init__(self, in_dim, embedding_dim, num_embeddings, num_hiddens, num_residual_layers, num_residual_hiddens,
                 commitment_cost=0.25, decay=0.99, share=False):
        super().__init__()
        self.in_dim = in_dim
        self.embedding_dim = embedding_dim
        self.num_embeddings = num_embeddings
        self.num_hiddens = num_hiddens
        self.share_code_vq = share
        self.encoder = _LegacyEncoder(in_dim, embedding_dim, num_hiddens, num_residual_layers)
        self.vq_layer = VectorQuantizerEMA(embedding_dim, num_embeddings, commitment_cost, decay)
        self.decoder = _LegacyDecoder(in_dim, embedding_dim, num_hiddens, num_residual_layers)

    def _groups(self):
        return 1

    def _cfg(self):
        return [self.in_dim, self.embedding_dim, self.num_embeddings, self.num_hiddens, 1, 3]

    def forward(self, gt_poses, id=None, pre_state=None, **kwargs):
        if self.training:                                   # vqvae_1d.py:170-173 -> (e_q_loss, gt_recon)
            e, e_q_loss = self.vq_layer(self._encode_train(gt_poses))
            return e_q_loss, self._decode_train(e)
        _, e, _ = self._encode_raw(gt_poses, want_z=False)
        return e, self._decode_raw(e)

    def encode(self, gt_poses):
        if self.training:
            return self.vq_layer(self._encode_train(gt_poses))
        _, e, idx = self._encode_raw(gt_poses, want_z=False)
        return e, idx.reshape(idx.shape[0], idx.shape[1])

    def decode(self, b, w, e=None, latents=None, pre_state=None):
        if e is None:
            e = self.vq_layer.quantize(latents).view(b, w, -1).permute(0, 2, 1).contiguous()
        return self._decode_raw(e), None


# ---------------------------------------------------------------------------
# audio condition encoder (vqvae_1d_sc.py:70-91)
# ---------------------------------------------------------------------------
class ConditionEncoder(_HandleOwner, nn.Module):
    _kind = "cond_encoder"
    _bf16_ok = True

    def __init__(self, in_dim, num_hiddens, num_residual_layers, num_residual_hiddens, dp=0.1):
        super().__init__()
        self.in_dim = in_dim
        self._num_hiddens = num_hiddens
        self.feature_map = nn.Sequential(nn.Conv1d(in_dim, num_hiddens, 1, 1, padding=0), nn.BatchNorm1d(num_hiddens),
                                         nn.LeakyReLU(0.1), nn.Dropout(dp))
        self._down_1 = ConvNormRelu(num_hiddens, num_hiddens, leaky=True, residual=True, sample='down', p=dp)
        self._down_2 = ConvNormRelu(num_hiddens, num_hiddens, leaky=True, residual=True, sample='down', p=dp)
        self._down_3 = ConvNormRelu(num_hiddens, num_hiddens, leaky=True, residual=True, sample='down', p=dp)

    def _cfg(self):
        return [self.in_dim, self._num_hiddens]

    def forward(self, x):
        if self.training:
            from .. import train_ops as T, training
            x = _cuda_keep_grad(x)
            B, Cc, Tn = x.shape
            rows, Tq = training.cond_encoder_train(_train_tensors(self), T.to_rows(x), B, Tn, self.feature_map[3].p,
                                                   _seeds_of(self))
            self.__dict__["_pt_handle"] = None
            return T.RowsToNCL.apply(rows, B, self._num_hiddens, Tq)
        x = _cuda_f32(x)
        B, Cc, T = x.shape
        h = self._handle()
        L = _lib.lib()
        Tq = T
        for _ in range(3):
            Tq = (Tq - 2) // 2 + 1
        out = torch.empty(B, self._num_hiddens, Tq, device=x.device)
        ws = h.workspace(L.pt_cond_encoder_workspace_bytes(h.h, B, T))
        check(L.pt_cond_encoder_forward(h.h, ptr(x), B, T, ptr(out), ptr(ws), ws.numel(), stream()))
        return out
