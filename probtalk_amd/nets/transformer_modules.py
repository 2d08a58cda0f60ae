"""nn.Module mirror of the predictor / refiner networks on libprobtalk_hip.so.

Reference classes mirrored (relative to /root/reference):
  nets/inpainting/gated_pixelcnn_1d.py:39-83     GatedConv
  nets/inpainting/predictornet.py:13-112         PredictorNet
  nets/inpainting/PTransformer.py:28-165         StylizationBlock_V3 (type 'sb')
  nets/inpainting/PTransformer.py:214-268        PositionalEncoding
  nets/inpainting/PTransformer.py:284-372        DeepSupervisionLayer
  nets/inpainting/PTransformer.py:394-704        PTransformer (tokens_to_logits, sample_good)
  nets/inpainting/RTransformer.py:181-230        RTransformer
The torch sub-modules are parameter containers (same state_dict keys as the reference); compute
goes through the C ABI.  Eval mode runs the packed inference sequences (MaskGIT sampling, refiner forward); train mode
(`.train()`) runs the training kernels through `probtalk_amd.train_ops` autograd Functions, so the reference's training
calls (`self.Predictor(input_poses, state, mask, id, audio, text, epoch_ratio)` -> {"logits": ...}, predictor.py:254;
`self.RTransformer(input_poses, aud, mask, id)`, refiner.py:296) work on the module API.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
from torch.nn.init import xavier_uniform_

from .. import _lib
from .._lib import check, ptr, stream
from .modules import _HandleOwner, _cuda_f32, _cuda_keep_grad, _seeds_of, _train_tensors


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


class StylizationBlock_V3(nn.Module):
    def __init__(self, type, in_dim, groups, dropout=0.1):
        super().__init__()
        if list(type) != ['sb']:
            raise NotImplementedError("only style_type=['sb'] is on the configured path (PTransformer.py:430)")
        self.type = type
        self.emb_layers_2 = nn.Sequential(nn.SiLU(), nn.Linear(in_dim, in_dim * 2))
        self.norm_2 = nn.LayerNorm(in_dim)
        self.out_layers_2 = nn.Sequential(nn.SiLU(), nn.Dropout(p=dropout), zero_module(nn.Linear(in_dim, in_dim)))


class PositionalEncoding(nn.Module):
    """Sinusoid table with a zero PAD row 0, kept in float64 like the reference (PTransformer.py:227-245)."""

    def __init__(self, d_model, max_seq_len, sine=True):
        super().__init__()
        self.position_encoding = nn.Embedding(max_seq_len + 1, d_model)
        if sine:
            pe = np.array([[pos / np.power(10000, 2.0 * (j // 2) / d_model) for j in range(d_model)]
                           for pos in range(max_seq_len)])
            pe[:, 0::2] = np.sin(pe[:, 0::2])
            pe[:, 1::2] = np.cos(pe[:, 1::2])
            table = torch.cat((torch.zeros([1, d_model]), torch.from_numpy(pe)))
            self.position_encoding.weight = nn.Parameter(table, requires_grad=False)
        else:
            self.position_encoding.weight = nn.Parameter(torch.zeros(max_seq_len + 1, d_model), requires_grad=True)


class DeepSupervisionLayer(nn.TransformerDecoderLayer):
    def __init__(self, d_model, nhead, groups, sb_every_layer=True, dropout=0.1, maskgit=False, style_type=['sb']):
        super().__init__(d_model, nhead, dropout=dropout)
        if not sb_every_layer:
            raise NotImplementedError("identity=False (no style blocks) is not on the configured path")
        self.d_model = d_model
        self.groups = groups
        self.sb_every_layer = sb_every_layer
        self.maskgit = maskgit
        self.style_type = style_type
        self.style_gourp_1 = StylizationBlock_V3(style_type, d_model, groups, dropout=0.)
        self.style_gourp_2 = StylizationBlock_V3(style_type, d_model, groups, dropout=0.)
        self.style_gourp_3 = StylizationBlock_V3(style_type, d_model, groups, dropout=0.)
        if groups > 1:
            self.pad_1 = nn.Parameter(torch.zeros([1, d_model, groups - 1]))
            self.pad_2 = nn.Parameter(torch.zeros([1, d_model, groups - 1]))
            self.pad_3 = nn.Parameter(torch.zeros([1, d_model, groups - 1]))
            xavier_uniform_(self.pad_1)
            xavier_uniform_(self.pad_2)
            xavier_uniform_(self.pad_3)
        else:
            self.pad_1 = self.pad_2 = self.pad_3 = 0

    def forward(self, *a, **k):
        raise NotImplementedError("DeepSupervisionLayer is fused into its parent module's HIP sequence")


class GatedConv(nn.Module):
    def __init__(self, dim, kernel, stride, padding, residual, upsample=False, double=False):
        super().__init__()
        assert residual and not upsample and not double
        self.residual = residual
        self.in_dim = dim // 2 if (stride == 2 and dim > 128) else dim
        self.vert_stack = nn.Conv1d(self.in_dim, dim * 2, kernel, stride, padding)
        self.conv = nn.Conv1d(self.in_dim, dim, kernel, stride, padding) if stride == 2 else nn.Identity()
        self.res = nn.Conv1d(dim, dim, 1)


class PTransformer(nn.Module):
    """MaskGIT token transformer.  Owns no handle: it computes through its PredictorNet parent
    (set by PredictorNet.__init__), because the packed model also holds the motion-context encoder."""

    def __init__(self, groups=4, input_dim=512, condi_dim=256, intermediate_dim=512, num_hidden_layers=12,
                 num_code=512, n_classes=4, identity=False, maskgit=False):
        super().__init__()
        if not maskgit or not identity:
            raise NotImplementedError("only the maskgit=True, identity=True form is on the configured path")
        if intermediate_dim != 512:
            raise NotImplementedError("d_model must be 512")
        self.groups = groups
        self.num_code = num_code
        self.mask_id = num_code
        self.identity = identity
        self.maskgit = maskgit
        self.choice_temperature = 4.5
        self.tp_pe = True
        self.n_classes = n_classes
        self.num_hidden_layers = num_hidden_layers
        self.condi_dim = condi_dim
        try:
            sine = bool(int(os.environ['sine']))
        except Exception:
            sine = True
        self.tok_emb = nn.Embedding(num_code + 1, intermediate_dim)
        self.class_embed = nn.Embedding(n_classes, intermediate_dim)
        self.condi_emb = nn.Conv1d(condi_dim, intermediate_dim, 1)
        self.PE = PositionalEncoding(intermediate_dim, 1800, sine=sine)
        self.product_PE = PositionalEncoding(intermediate_dim, groups, sine=sine)
        layer = DeepSupervisionLayer(d_model=intermediate_dim, nhead=8, groups=groups, sb_every_layer=identity,
                                     dropout=0.1, maskgit=maskgit, style_type=['sb'])
        self.transformer_decoder = nn.TransformerDecoder(layer, num_layers=num_hidden_layers)
        self.output_conv = nn.Sequential(nn.Conv1d(intermediate_dim, 512, 1), nn.ReLU(True), nn.Conv1d(512, num_code, 1))
        self.epoch_ratio = 0

    def _owner(self):
        o = self.__dict__.get("_pt_owner")
        if o is None:
            raise RuntimeError("PTransformer computes through its PredictorNet (construct it via PredictorNet)")
        return o

    def tokens_to_logits(self, x, label, condition):
        """x (B,T',G) int (mask id = num_code), label (B,), condition (B,768,T') -> (B,T',G,num_code)."""
        if self.training:
            raise RuntimeError("tokens_to_logits is the inference entry (no dropout, no autograd): call .eval() first, or "
                               "use forward(x, label, condition, input_codes, epoch_ratio) to train")
        cond = _cuda_f32(condition)
        B, _, Tq = cond.shape
        ids = x.to(device=cond.device, dtype=torch.int32).contiguous()
        lab = label.to(device=cond.device, dtype=torch.int64).contiguous()
        if lab.numel() != B:
            raise RuntimeError(f"label must hold one speaker id per clip ({B}), got {lab.numel()}")
        h = self._owner()._handle()
        L = _lib.lib()
        logits = torch.empty(B, Tq, self.groups, self.num_code, device=cond.device)
        ws = h.workspace(L.pt_predictor_workspace_bytes(h.h, B, Tq * 8))
        check(L.pt_predictor_logits(h.h, ptr(ids), ptr(lab), ptr(cond), B, Tq, ptr(logits), ptr(ws), ws.numel(),
                                    stream()))
        return logits

    @torch.no_grad()
    def sample_good(self, label=None, condition=None, input_codes=None, T=10, mode="cosine", noise=None,
                    return_logits=False, seed=None):
        """MaskGIT decoding (PTransformer.py:590-642) -> codes (B,T',G) int32.

        `noise=(q, g)` injects the two random draws (q ~ Exp(1) of shape (T,B,N,num_code) for the categorical sample,
        g ~ Gumbel(0,1) of shape (T,B,N)): the parity mode.  By default both are generated inside the MaskGIT step kernel
        from a 64-bit seed (`seed=`, else drawn from torch's CPU generator, so torch.manual_seed makes runs repeatable):
        no noise tensors and no host RNG call per pass (the reference draws the Gumbel noise on the CPU and copies it
        each pass, PTransformer.py:697-698).  A one-element int64 CUDA tensor as `seed` is read by the kernels at run
        time: nothing of the call depends on a host value then, so it can be captured into a HIP graph and replayed with a
        new seed written to that tensor (probtalk_amd/graphs.py)."""
        if self.training:
            raise RuntimeError("sample_good is the inference entry (the reference calls it under eval()): call .eval() first")
        if mode != "cosine":
            raise NotImplementedError("sample_good is only called with the cosine schedule")
        cond = _cuda_f32(condition)
        B, _, Tq = cond.shape
        N = Tq * self.groups
        lab = label.to(device=cond.device, dtype=torch.int64).contiguous()
        h = self._owner()._handle()
        L = _lib.lib()
        codes = torch.empty(B, Tq, self.groups, dtype=torch.int32, device=cond.device)
        logits = torch.empty(T, B, N, self.num_code, device=cond.device) if return_logits else None
        ws = h.workspace(L.pt_predictor_workspace_bytes(h.h, B, Tq * 8))
        if noise is None and torch.is_tensor(seed):
            if not (seed.is_cuda and seed.dtype == torch.int64 and seed.numel() == 1):
                raise RuntimeError("sample_good: a tensor seed must be one int64 element on the GPU")
            check(L.pt_predictor_sample_seed_dev(h.h, ptr(lab), ptr(cond), ptr(seed), T, B, Tq, ptr(codes), ptr(logits), ptr(ws),
                                                 ws.numel(), stream()))
        elif noise is None:
            if seed is None:
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            check(L.pt_predictor_sample_seeded(h.h, ptr(lab), ptr(cond), int(seed), T, B, Tq, ptr(codes), ptr(logits), ptr(ws),
                                               ws.numel(), stream()))
        else:
            q, g = (_cuda_f32(t) for t in noise)
            if tuple(q.shape) != (T, B, N, self.num_code) or tuple(g.shape) != (T, B, N):
                raise RuntimeError("sample_good: injected noise has the wrong shape")
            check(L.pt_predictor_sample(h.h, ptr(lab), ptr(cond), ptr(q), ptr(g), T, B, Tq, ptr(codes), ptr(logits),
                                        ptr(ws), ws.numel(), stream()))
        return (codes, logits) if return_logits else codes

    def gamma(self, r):
        """PTransformer.py:437,578-580: the 'linear' schedule, share of tokens left unmasked."""
        return 1 - r

    def random_token(self, state, type, epoch_ratio, keep_scores=None):
        """PTransformer.py:559-576 ('mask'): keep r = floor(gamma(ratio) * N) random tokens per clip, the rest become the
        mask id.  keep_scores (B, N) stands in for torch.rand (tests)."""
        if type != 'mask':
            raise NotImplementedError("only the 'mask' replacement is used on the maskgit path")
        B = state.shape[0]
        flat = state.reshape(B, -1)
        r = math.floor(self.gamma(epoch_ratio) * flat.shape[1])
        if keep_scores is None:
            keep_scores = torch.rand(flat.shape, device=flat.device)
        keep = torch.zeros_like(flat, dtype=torch.bool)
        if r > 0:
            keep.scatter_(1, keep_scores.topk(r, dim=1).indices, True)
        return torch.where(keep, flat, torch.full_like(flat, self.mask_id)).reshape(state.shape)

    def _forward_rows(self, x, label, cond_rows, B, Tq, epoch_ratio):
        from .. import training
        owner = self._owner()
        ids = self.random_token(x, 'mask', epoch_ratio, self.__dict__.pop("_pt_keep_scores", None))
        p_drop = self.transformer_decoder.layers[0].dropout.p
        lab = label.to(device=cond_rows.device, dtype=torch.int64).contiguous().reshape(-1)
        rows = training.ptransformer_train(_train_tensors(owner), ids, lab, cond_rows, B, Tq, self.groups, p_drop,
                                           _seeds_of(owner))
        owner.__dict__["_pt_handle"] = None         # parameters are about to move: the packed inference weights are stale
        return {"logits": rows[:, :self.num_code].reshape(B, Tq, self.groups, self.num_code)}

    def forward(self, x, label, condition, input_codes, epoch_ratio):
        """PTransformer.py:443-504 (training): x (B,T',G) code indices, condition (B,768,T') -> {"logits": (B,T',G,num_code)}."""
        if input_codes is not None:
            raise NotImplementedError("input_codes (motion_block) is not on the configured path")
        from .. import train_ops as T
        cond = _cuda_keep_grad(condition)
        B, Cc, Tq = cond.shape
        return self._forward_rows(x, label, T.to_rows(cond, Cc), B, Tq, epoch_ratio)


class PredictorNet(_HandleOwner, nn.Module):
    """predictornet.py:13-112."""
    _kind = "predictor"
    _bf16_ok = True

    def __init__(self, knn=False, mot_dim=256, sta_dim=2048, dim=64, enc_layers=10, ar_layers=15, n_classes=10,
                 groups=1, identity=False, maskgit=False, maskgit_T=8, transformer=False, text=False, audio=True,
                 motion_context=True):
        super().__init__()
        if not transformer:
            raise NotImplementedError("the PixelCNN decoder (transformer=False) is not on the configured path")
        if not motion_context or audio is None or knn:
            raise NotImplementedError("only motion_context=True, audio set, knn=False is on the configured path")
        if dim != 512 or enc_layers != 10:
            raise NotImplementedError("the HIP sequence is built for dim=512, enc_layers=10 (predictor.py:72)")
        self.dim = dim
        self.knn = knn
        self.groups = groups
        self.ar_layers = ar_layers
        self.num_code = sta_dim
        self.mask_id = self.num_code
        self.maskgit = maskgit
        self.T = maskgit_T
        self.motion_context = motion_context
        self.is_transformer = transformer
        self.mot_dim = mot_dim
        self.n_classes = n_classes
        self.text = text
        self.motion_embedding = nn.Conv1d(mot_dim + 1, dim // 4, 1, 1, padding=0)
        self.enc = nn.ModuleList()
        d = dim // 4
        for i in range(enc_layers):
            if i == 0:
                kernel, stride, padding = 7, 1, 3
            elif i in [2, 4, 6]:
                kernel, stride, padding = 4, 2, 1
                if i in [4, 6]:
                    d = d * 2
            else:
                kernel, stride, padding = 3, 1, 1
            self.enc.append(GatedConv(d, kernel, stride, padding, True))
        condi_dim = d + 256 * (audio is not None) + 256 * (text is not None)
        self.decoder = PTransformer(groups, 512, condi_dim, 512, 6, self.num_code, n_classes, identity, maskgit)
        self.decoder.__dict__["_pt_owner"] = self
        self.epoch_ratio = 0

    def _cfg(self):
        return [self.mot_dim, self.num_code, self.groups, self.n_classes, self.decoder.num_hidden_layers,
                self.decoder.condi_dim - 512]

    def _features(self, audio, text):
        """Audio (and, with p_text set, text) condition features side by side, as predictornet.py:93-97 concatenates them."""
        if (text is not None) != (self.text is not None):
            raise RuntimeError("PredictorNet: text features must be passed exactly when the net was built with text=<name>")
        return audio if text is None else torch.cat([audio, text], dim=1)

    def condition(self, masked_motion, mask, audio, text=None):
        """The part of forward before the decoder call (predictornet.py:87-100): (B, 768 | 1024 with text, T')
        (inference sequence)."""
        audio = self._features(audio, text)
        mm, mk, au = _cuda_f32(masked_motion), _cuda_f32(mask), _cuda_f32(audio)
        B, Cc, T = mm.shape
        if Cc != self.mot_dim or tuple(mk.shape) != (B, 1, T):
            raise RuntimeError("PredictorNet: masked_motion must be (B,%d,T) and mask (B,1,T)" % self.mot_dim)
        Tq = T
        for _ in range(3):
            Tq = (Tq - 2) // 2 + 1
        if tuple(au.shape) != (B, self.decoder.condi_dim - 512, Tq):
            raise RuntimeError(f"PredictorNet: audio features must be (B,{self.decoder.condi_dim - 512},{Tq}), "
                               f"got {tuple(au.shape)}")
        h = self._handle()
        L = _lib.lib()
        cond = torch.empty(B, self.decoder.condi_dim, Tq, device=mm.device)
        ws = h.workspace(L.pt_predictor_workspace_bytes(h.h, B, T))
        check(L.pt_predictor_condition(h.h, ptr(mm), ptr(mk), ptr(au), B, T, ptr(cond), ptr(ws), ws.numel(), stream()))
        return cond

    def forward(self, masked_motion, state, mask, label, audio, text, epoch_ratio, input_codes=None,
                distribution=False, noise=None, seed=None):
        if distribution:
            raise NotImplementedError("distribution=True calls a method the reference does not define")
        if self.training:
            # predictornet.py:87-103: motion-context encoder, cat with the audio features, decoder(state, ...) -> logits
            from .. import train_ops as T, training
            mm, mk, au = _cuda_keep_grad(masked_motion), _cuda_keep_grad(mask), _cuda_keep_grad(self._features(audio, text))
            B, _, Tn = mm.shape
            motion, Tq = training.motion_context_train(_train_tensors(self), T.to_rows(torch.cat([mm, mk], dim=1)), B, Tn)
            if tuple(au.shape) != (B, self.decoder.condi_dim - 512, Tq):
                raise RuntimeError(f"PredictorNet: audio features must be (B,{self.decoder.condi_dim - 512},{Tq}), "
                                   f"got {tuple(au.shape)}")
            cond = torch.cat([motion, T.to_rows(au, au.shape[1])], dim=1).contiguous()
            return self.decoder._forward_rows(state, label, cond, B, Tq, epoch_ratio)
        cond = self.condition(masked_motion, mask, audio, text)
        return self.decoder.sample_good(label, cond, input_codes, T=self.T, noise=noise, seed=seed)


class RTransformer(_HandleOwner, nn.Module):
    """RTransformer.py:181-230."""
    _kind = "refiner"
    _bf16_ok = True

    def __init__(self, input_dim=512, condi_dim=512, intermediate_dim=512, num_hidden_layers=6, use_label=True,
                 n_classes=4, motion_context=False):
        super().__init__()
        if not (use_label and motion_context) or intermediate_dim != 512:
            raise NotImplementedError("only use_label=True, motion_context=True, d_model=512 is on the configured path")
        self.input_dim = input_dim
        self.cond_dim = condi_dim
        self.n_classes = n_classes
        self.num_hidden_layers = num_hidden_layers
        self.win_size = 7
        self.motion_context = motion_context
        self.class_embed = nn.Embedding(n_classes, intermediate_dim)
        self.PE = PositionalEncoding(intermediate_dim, 1800)
        self.input_conv = nn.Conv1d(input_dim, intermediate_dim, 1, 1, 0)
        self.audio_conv = nn.Conv1d(condi_dim + 2, intermediate_dim, 1, 1, 0)
        layer = DeepSupervisionLayer(d_model=intermediate_dim, nhead=8, groups=1, sb_every_layer=use_label)
        self.transformer_decoder = nn.TransformerDecoder(layer, num_layers=num_hidden_layers)
        self.output_conv = nn.Sequential(nn.Conv1d(intermediate_dim, 1024, 1), nn.ReLU(True),
                                         nn.Conv1d(1024, input_dim, 1))

    def _cfg(self):
        return [self.input_dim, self.cond_dim, self.n_classes, self.num_hidden_layers]

    def forward(self, x, condition, mask, label):
        if self.training:
            # RTransformer.py:209-230 in train mode (dropout 0.1 in every decoder layer)
            from .. import train_ops as T, training
            xg, cg, mg = _cuda_keep_grad(x), _cuda_keep_grad(condition), _cuda_keep_grad(mask)
            B, Cc, Tn = xg.shape
            cond_rows = T.to_rows(torch.cat([cg, 1.0 - mg, mg], dim=1))       # F.one_hot(mask, 2): [mask==0 | mask==1]
            lab = label.to(device=xg.device, dtype=torch.int64).contiguous().reshape(-1)
            rows = training.rtransformer_train(_train_tensors(self), T.to_rows(xg), cond_rows, lab, B, Tn,
                                               self.transformer_decoder.layers[0].dropout.p, _seeds_of(self))
            self.__dict__["_pt_handle"] = None
            return T.RowsToNCL.apply(rows, B, self.input_dim, Tn)
        x, cond, mk = _cuda_f32(x), _cuda_f32(condition), _cuda_f32(mask)
        B, Cc, T = x.shape
        if Cc != self.input_dim or tuple(cond.shape) != (B, self.cond_dim, T) or tuple(mk.shape) != (B, 1, T):
            raise RuntimeError("RTransformer: expected x (B,%d,T), condition (B,%d,T), mask (B,1,T)"
                               % (self.input_dim, self.cond_dim))
        lab = label.to(device=x.device, dtype=torch.int64).contiguous().reshape(-1)
        h = self._handle()
        L = _lib.lib()
        out = torch.empty(B, self.input_dim, T, device=x.device)
        ws = h.workspace(L.pt_refiner_workspace_bytes(h.h, B, T))
        check(L.pt_refiner_forward(h.h, ptr(x), ptr(cond), ptr(mk), ptr(lab), lab.numel(), B, T, ptr(out), ptr(ws),
                                   ws.numel(), stream()))
        return out


def interp_blend(pred, size, gt=None, mask=None, premask=False):
    """F.interpolate(pred, size, mode='linear') [+ gt*mask(*mask) + pred*(1-mask)] on the GPU
    (predictor.py:445-446, refiner.py:528-529,553)."""
    g = m = None
    if gt is not None:
        g, m = _cuda_f32(gt), _cuda_f32(mask)
    if pred is None:                       # plain gt * mask
        B, Cc, Tin = g.shape
        dev = g.device
    else:
        pred = _cuda_f32(pred)
        B, Cc, Tin = pred.shape
        dev = pred.device
    if g is not None and (tuple(g.shape) != (B, Cc, size) or tuple(m.shape) != (B, 1, size)):
        raise RuntimeError("interp_blend: gt must be (B,C,size) and mask (B,1,size)")
    out = torch.empty(B, Cc, size, device=dev)
    check(_lib.lib().pt_interp_blend(ptr(pred), B, Cc, Tin, size, ptr(g), ptr(m), int(premask), ptr(out), stream()))
    return out
