"""Autograd glue of the training path: every Function's forward AND backward run the library's HIP kernels through
the C ABI (`pt_conv1d_cl`, `pt_attention_cl`, `pt_embed_pe`, `pt_train_*`); torch.autograd is only the tape, torch
itself only moves memory (permute / pad / cat of weights and rows).  No CPU or PyTorch-kernel fallback exists: the ops
raise if the library or the GPU is missing.

All activations are channels-last 2-D fp32 tensors `[B*L, C]` with C a multiple of 32 (pad columns zero), the layout
of the inference library.  Conv `kind`: 0 = Linear / 1x1, 1 = k3 s1 p1, 2 = k7 s1 p3, 3 = k4 s2 p1, 4 = ConvTranspose
k4 s2 p1 (reference shapes: nets/spg/vqvae_modules.py:107-200, nets/inpainting/gated_pixelcnn_1d.py:39-83).
"""
import torch

from . import _lib
from ._lib import check, ptr, stream

_KT = {0: 1, 1: 3, 2: 7, 3: 4, 4: 4}


def _ru(x, m):
    return (x + m - 1) // m * m


def _zeros(*shape):
    return torch.zeros(*shape, device="cuda", dtype=torch.float32)


def _empty(*shape):
    return torch.empty(*shape, device="cuda", dtype=torch.float32)


def pack_taps(w_nck):
    """(N, Cin, kt) -> contiguous [kt][Npad][Kpad] fp32 (the tap-GEMM's weight layout), zero padded."""
    N, Cin, kt = w_nck.shape
    Np, Kp = _ru(N, 128), _ru(Cin, 32)
    if kt == 1 and Np == N and Kp == Cin:
        return w_nck.reshape(1, N, Cin).contiguous(), Np, Kp
    out = _zeros(kt, Np, Kp)
    out[:, :N, :Cin] = w_nck.permute(2, 0, 1)
    return out, Np, Kp


def conv_raw(x2d, w_nck, bias, kind, B, Lin, act=0, slope=0.0, res=None):
    """y = act(conv(x) + bias + res) on channels-last rows through the fp32 tap-GEMM."""
    L = _lib.lib()
    N = w_nck.shape[0]
    wp, Np, Kp = pack_taps(w_nck)
    if x2d.shape[1] < Kp:
        raise RuntimeError(f"conv: input rows have {x2d.shape[1]} columns, packed K is {Kp}")
    Lout = {0: Lin, 1: Lin, 2: Lin, 3: (Lin - 2) // 2 + 1, 4: 2 * Lin}[kind]
    ldc = _ru(N, 32)
    y = _empty(B * Lout, ldc)
    bp = None
    if bias is not None:
        if Np == N and bias.is_contiguous() and bias.dtype == torch.float32:
            bp = bias                       # no padding to add: skips a fill and a copy launch per Linear
        else:
            bp = _zeros(Np)
            bp[:N] = bias
    check(L.pt_conv1d_cl(ptr(x2d), x2d.shape[1], ptr(wp), ptr(bp), N, Np, Kp, kind, B, Lin, ptr(y), ldc, act, slope,
                         ptr(res), 0 if res is None else res.shape[1], stream()))
    return y, Lout


class ConvCL(torch.autograd.Function):
    """Conv1d / ConvTranspose1d / Linear with optional fused residual and (leaky-)ReLU; weights in the reference layout
    ((N, Cin[, kt]) or, for kind 4, (Cin, N, kt))."""

    @staticmethod
    def forward(ctx, x, w, bias, res, kind, B, Lin, slope):
        w3 = w if w.dim() == 3 else w.unsqueeze(-1)
        w_nck = w3.permute(1, 0, 2) if kind == 4 else w3
        act = 0 if slope < 0 else 1
        y, Lout = conv_raw(x, w_nck.contiguous(), bias, kind, B, Lin, act, max(slope, 0.0), res)
        ctx.save_for_backward(x, w, y if act else None)
        ctx.meta = (kind, B, Lin, Lout, slope, bias is not None, res is not None, w.dim())
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        kind, B, Lin, Lout, slope, has_bias, has_res, wdim = ctx.meta
        L = _lib.lib()
        dy = dy.contiguous()
        N = w.shape[1] if kind == 4 else w.shape[0]
        Cin = w.shape[0] if kind == 4 else w.shape[1]
        if slope >= 0:
            dz = _empty(dy.shape)
            check(L.pt_train_act_bwd(ptr(dy), dy.shape[1], ptr(y), y.shape[1], slope, ptr(dz), dz.shape[1], dy.shape[0],
                                     dy.shape[1], stream()))
            dy = dz
        w3 = w if w.dim() == 3 else w.unsqueeze(-1)
        dx = dw = db = dres = None
        if ctx.needs_input_grad[0]:
            # input gradient = the same tap-GEMM on dY with re-laid-out weights (see train.hip header)
            if kind in (0, 1, 2):
                wt = w3.permute(1, 0, 2).flip(2).contiguous()            # (Cin, N, kt), taps reversed
                dx, _ = conv_raw(dy, wt, None, kind, B, Lout)
            elif kind == 3:
                wt = w3.permute(1, 0, 2).contiguous()                    # ConvTranspose form, W_k^T per tap
                if 2 * Lout < Lin:                                       # odd input length: one more (zero) output row
                    dyp = _zeros(B, Lout + 1, dy.shape[1])
                    dyp[:, :Lout] = dy.view(B, Lout, -1)
                    full, _ = conv_raw(dyp.view(B * (Lout + 1), -1), wt, None, 4, B, Lout + 1)
                    dx = full.view(B, 2 * (Lout + 1), -1)[:, :Lin].reshape(B * Lin, -1).contiguous()
                else:
                    dx, _ = conv_raw(dy, wt, None, 4, B, Lout)
                    if 2 * Lout > Lin:
                        dx = dx.view(B, 2 * Lout, -1)[:, :Lin].reshape(B * Lin, -1).contiguous()
            else:
                wt = w3.contiguous()                                     # (Cin, N, kt) is already (out=Cin, in=N, kt)
                dx, _ = conv_raw(dy, wt, None, 3, B, Lout)
            if dx.shape[1] != x.shape[1]:
                t = _zeros(x.shape)
                t[:, :min(dx.shape[1], x.shape[1])] = dx[:, :min(dx.shape[1], x.shape[1])]
                dx = t
        want_db = has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            kt = _KT[kind]
            dwp = _zeros(kt, N, Cin)
            if want_db:
                # the bias gradient rides on the weight-gradient kernel (it stages the dY tiles anyway)
                db = _zeros(N)
                check(L.pt_train_wgrad_bias(ptr(dy), dy.shape[1], ptr(x), x.shape[1], ptr(dwp), ptr(db), N, Cin, kind, B, Lin,
                                            stream()))
                want_db = False
            else:
                check(L.pt_train_wgrad(ptr(dy), dy.shape[1], ptr(x), x.shape[1], ptr(dwp), N, Cin, kind, B, Lin, stream()))
            dw = dwp.permute(2, 1, 0) if kind == 4 else dwp.permute(1, 2, 0)
            dw = dw.contiguous() if wdim == 3 else dw.reshape(N, Cin).contiguous()
        if want_db:
            db = _zeros(N)
            check(L.pt_train_colsum(ptr(dy), dy.shape[1], None, 0, dy.shape[0], N, ptr(db), stream()))
        if has_res and ctx.needs_input_grad[3]:
            dres = dy
        return dx, dw, db, dres, None, None, None, None


def conv(x, w, bias, kind, B, Lin, slope=-1.0, res=None):
    return ConvCL.apply(x, w, bias, res, kind, B, Lin, slope)


class LayerNormRes(torch.autograd.Function):
    """y = LayerNorm(x + res) over 512-wide rows (PTransformer.py:329-331 post-LN residual; norm_2 of the style block)."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta):
        L = _lib.lib()
        rows = x.shape[0]
        y, stats = _empty(rows, 512), _empty(rows, 2)
        check(L.pt_train_ln_fwd(ptr(x), ptr(res), ptr(gamma), ptr(beta), ptr(y), ptr(stats), rows, stream()))
        ctx.save_for_backward(x, res, gamma, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, res, gamma, stats = ctx.saved_tensors
        L = _lib.lib()
        rows = x.shape[0]
        dy = dy.contiguous()
        dx, dg, db = _empty(rows, 512), _zeros(512), _zeros(512)
        check(L.pt_train_ln_bwd(ptr(dy), ptr(x), ptr(res), ptr(stats), ptr(gamma), ptr(dx), ptr(dg), ptr(db), rows, stream()))
        return dx, (dx if res is not None else None), dg, db


def layer_norm(x, gamma, beta, res=None):
    if x.shape[1] != 512:
        raise RuntimeError("layer_norm: the path's LayerNorms are 512 wide")
    return LayerNormRes.apply(x, res, gamma, beta)


class ModSiLU(torch.autograd.Function):
    """h = SiLU(y * (1 + scale[c]) + shift[c]) with table (n_classes, 1024) = scale | shift (PTransformer.py:134-139)."""

    @staticmethod
    def forward(ctx, y, table, label, rows_per_clip):
        L = _lib.lib()
        h = _empty(y.shape)
        check(L.pt_train_modsilu_fwd(ptr(y), ptr(table), ptr(label), label.numel(), rows_per_clip, table.shape[0], ptr(h),
                                     y.shape[0], stream()))
        ctx.save_for_backward(y, table, label)
        ctx.rpc = rows_per_clip
        return h

    @staticmethod
    def backward(ctx, dh):
        y, table, label = ctx.saved_tensors
        L = _lib.lib()
        dh = dh.contiguous()
        dy, dt = _empty(y.shape), _zeros(table.shape)
        check(L.pt_train_modsilu_bwd(ptr(dh), ptr(y), ptr(table), ptr(label), label.numel(), ctx.rpc, table.shape[0], ptr(dy),
                                     ptr(dt), y.shape[0], stream()))
        return dy, dt, None, None


class Attention(torch.autograd.Function):
    """softmax(QK^T/8)V per (clip, head), 8 heads x 64, on row tensors: q [B*Nq, ldq], kv rows [B*Nk, ldk] with K at
    column k_off and V at v_off.  Self-attention passes the packed in-proj output (q | k | v) as both tensors.
    p > 0: dropout on the attention weights (nn.MultiheadAttention(dropout) in train mode), mask replayed from the seed."""

    @staticmethod
    def forward(ctx, q, kv, q_off, k_off, v_off, B, Nq, Nk, p, seed):
        import ctypes as C
        L = _lib.lib()
        o = _empty(B * Nq, 512)
        qb, kb = q.data_ptr() + 4 * q_off, kv.data_ptr()
        if p > 0:
            check(L.pt_train_attention_fwd(C.c_void_p(qb), q.shape[1], C.c_void_p(kb + 4 * k_off), kv.shape[1],
                                           C.c_void_p(kb + 4 * v_off), kv.shape[1], ptr(o), 512, B, Nq, Nk, 8, p, seed, stream()))
        else:
            check(L.pt_attention_cl(C.c_void_p(qb), q.shape[1], C.c_void_p(kb + 4 * k_off), kv.shape[1],
                                    C.c_void_p(kb + 4 * v_off), kv.shape[1], ptr(o), 512, B, Nq, Nk, 8, stream()))
        ctx.save_for_backward(q, kv)
        ctx.meta = (q_off, k_off, v_off, B, Nq, Nk, p, seed, q.data_ptr() == kv.data_ptr())
        return o

    @staticmethod
    def backward(ctx, do):
        import ctypes as C
        q, kv = ctx.saved_tensors
        q_off, k_off, v_off, B, Nq, Nk, p, seed, packed = ctx.meta
        L = _lib.lib()
        do = do.contiguous()
        dq = _zeros(q.shape)
        dkv = dq if packed else _zeros(kv.shape)
        lse, delta = _empty(B * 8 * Nq), _empty(B * 8 * Nq)
        qb, kb, dqb, dkb = q.data_ptr() + 4 * q_off, kv.data_ptr(), dq.data_ptr() + 4 * q_off, dkv.data_ptr()
        check(L.pt_train_attention_bwd(C.c_void_p(qb), q.shape[1], C.c_void_p(kb + 4 * k_off), kv.shape[1],
                                       C.c_void_p(kb + 4 * v_off), kv.shape[1], ptr(do), 512, C.c_void_p(dqb), dq.shape[1],
                                       C.c_void_p(dkb + 4 * k_off), dkv.shape[1], C.c_void_p(dkb + 4 * v_off), dkv.shape[1],
                                       ptr(lse), ptr(delta), B, Nq, Nk, 8, p, seed, stream()))
        return dq, (None if packed else dkv), None, None, None, None, None, None, None, None


def attention_self(qkv, B, N, p=0.0, seeds=None):
    p = float(p) if seeds is not None else 0.0
    return Attention.apply(qkv, qkv, 0, 512, 1024, B, N, N, p, seeds.next() if p > 0 else 0)


def attention_cross(q, kv, B, Nq, Nk, p=0.0, seeds=None):
    p = float(p) if seeds is not None else 0.0
    return Attention.apply(q, kv, 0, 0, 512, B, Nq, Nk, p, seeds.next() if p > 0 else 0)


class CrossEntropy(torch.autograd.Function):
    """F.cross_entropy(logits [rows, V], target [rows]) with mean reduction (predictor.py:261)."""

    @staticmethod
    def forward(ctx, logits, target):
        L = _lib.lib()
        rows, V = logits.shape
        loss, dl = _zeros(1), _empty(rows, V)
        check(L.pt_train_cross_entropy(ptr(logits), ptr(target), V, rows, ptr(loss), ptr(dl), stream()))
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None


class BNTrain(torch.autograd.Function):
    """y = act(BatchNorm1d_train(x) + res) on channels-last rows, batch statistics over all rows; running statistics
    are updated in place like nn.BatchNorm1d (momentum 0.1, unbiased running variance)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, slope, run_mean, run_var, momentum):
        L = _lib.lib()
        rows, cols = x.shape[0], gamma.numel()
        y = _zeros(rows, x.shape[1]) if x.shape[1] != cols else _empty(rows, cols)
        mr = _empty(2, cols)
        acc = torch.empty(2 * cols, device="cuda", dtype=torch.float64)
        check(L.pt_train_bn_fwd(ptr(x), x.shape[1], rows, cols, ptr(gamma), ptr(beta), ptr(res),
                                0 if res is None else res.shape[1], slope, ptr(y), y.shape[1], ptr(mr), ptr(acc), ptr(run_mean),
                                ptr(run_var), momentum, stream()))
        ctx.save_for_backward(x, gamma, y, mr)
        ctx.meta = (slope, res is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, y, mr = ctx.saved_tensors
        slope, has_res = ctx.meta
        L = _lib.lib()
        rows, cols = x.shape[0], gamma.numel()
        dy = dy.contiguous()
        dz = _zeros(dy.shape) if dy.shape[1] != cols else _empty(dy.shape)
        dx = _zeros(x.shape) if x.shape[1] != cols else _empty(x.shape)
        sums = _empty(2, cols)
        check(L.pt_train_bn_bwd(ptr(dy), dy.shape[1], ptr(y), y.shape[1], slope, ptr(x), x.shape[1], ptr(mr), ptr(gamma), ptr(dz),
                                dz.shape[1], ptr(dx), dx.shape[1], ptr(sums), rows, cols, stream()))
        return dx, sums[1].clone(), sums[0].clone(), (dz if has_res else None), None, None, None, None


def bn_train(x, gamma, beta, run_mean, run_var, slope=-1.0, res=None, momentum=0.1):
    return BNTrain.apply(x, gamma, beta, res, slope, run_mean, run_var, momentum)


class Gate(torch.autograd.Function):
    """tanh(a) * sigmoid(b) on hv = [a | b] (gated_pixelcnn_1d.py:74-76)."""

    @staticmethod
    def forward(ctx, hv):
        L = _lib.lib()
        rows, C2 = hv.shape
        out = _empty(rows, C2 // 2)
        check(L.pt_train_gate_fwd(ptr(hv), rows, C2 // 2, ptr(out), stream()))
        ctx.save_for_backward(hv)
        return out

    @staticmethod
    def backward(ctx, dout):
        (hv,) = ctx.saved_tensors
        L = _lib.lib()
        dhv = _empty(hv.shape)
        check(L.pt_train_gate_bwd(ptr(dout.contiguous()), ptr(hv), hv.shape[0], hv.shape[1] // 2, ptr(dhv), stream()))
        return dhv


class EmbedPE(torch.autograd.Function):
    """x[row] = float(double(tok_emb[id]) + (PE[t+1] + PPE[g+1])) (PTransformer.py:447-474); gradient to tok_emb only
    (the sinusoid tables are frozen, PTransformer.py:243-245)."""

    @staticmethod
    def forward(ctx, ids32, tok, pe, ppe, N, G):
        L = _lib.lib()
        rows = ids32.numel()
        x = _empty(rows, 512)
        check(L.pt_embed_pe(ptr(ids32), ptr(tok), ptr(pe), ptr(ppe), ptr(x), rows, N, G, tok.shape[0], stream()))
        ctx.save_for_backward(ids32)
        ctx.vocab = tok.shape[0]
        return x

    @staticmethod
    def backward(ctx, dx):
        (ids32,) = ctx.saved_tensors
        L = _lib.lib()
        dE = _zeros(ctx.vocab, 512)
        check(L.pt_train_embed_bwd(ptr(dx.contiguous()), ptr(ids32), ids32.numel(), ctx.vocab, ptr(dE), stream()))
        return None, dE, None, None, None, None


class Dropout(torch.autograd.Function):
    """Inverted dropout with a counter-based mask that the backward pass replays from the seed."""

    @staticmethod
    def forward(ctx, x, p, seed):
        L = _lib.lib()
        y = _empty(x.shape)
        check(L.pt_train_dropout(ptr(x), ptr(y), x.numel(), p, seed, stream()))
        ctx.meta = (p, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed = ctx.meta
        L = _lib.lib()
        dy = dy.contiguous()
        dx = _empty(dy.shape)
        check(L.pt_train_dropout(ptr(dy), ptr(dx), dy.numel(), p, seed, stream()))
        return dx, None, None


class _Seeds:
    """Per-step dropout seed stream (host counter; the device never sees torch's RNG).  The rank of the process is folded
    into every seed when it is drawn, so data-parallel replicas draw independent masks (as nn.DataParallel's replicas do)
    whenever the process group was initialised; the position `n` is checkpointed by the wrappers."""

    def __init__(self, seed=0):
        self.base, self.n = int(seed) & 0xFFFFFFFF, 0

    def next(self):
        import torch.distributed as dist
        self.n += 1
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        return (((self.base ^ (rank << 16)) & 0xFFFFFFFF) << 20) + self.n

    def state_dict(self):
        return {"base": self.base, "n": self.n}

    def load_state_dict(self, sd):
        self.base, self.n = int(sd["base"]), int(sd["n"])


def dropout(x, p, seeds):
    if p <= 0.0 or seeds is None:
        return x
    return Dropout.apply(x.contiguous(), float(p), seeds.next())


def ncl_to_rows(x_ncl, width=None):
    """(B, C, L) fp32 CUDA -> rows [B*L, round_up(C,32)] (pad columns zero); no gradient (inputs)."""
    L = _lib.lib()
    B, C, Ln = x_ncl.shape
    ld = width or _ru(C, 32)
    out = _empty(B * Ln, ld)
    check(L.pt_ncl_to_cl(ptr(x_ncl.contiguous()), ptr(out), B, C, Ln, ld, stream()))
    return out


class NCLToRows(torch.autograd.Function):
    """(B, C, L) -> rows [B*L, ld] (pad columns zero) with a gradient: the module API's tensors are channels-first, the
    kernels' are channels-last rows."""

    @staticmethod
    def forward(ctx, x, ld):
        B, C, Ln = x.shape
        out = _empty(B * Ln, ld)
        check(_lib.lib().pt_ncl_to_cl(ptr(x.contiguous()), ptr(out), B, C, Ln, ld, stream()))
        ctx.meta = (B, C, Ln, ld)
        return out

    @staticmethod
    def backward(ctx, dy):
        B, C, Ln, ld = ctx.meta
        dx = _empty(B, C, Ln)
        check(_lib.lib().pt_cl_to_ncl(ptr(dy.contiguous()), ld, ptr(dx), B, C, Ln, stream()))
        return dx, None


class RowsToNCL(torch.autograd.Function):
    """rows [B*L, ld] -> (B, C, L) (the first C columns) with a gradient."""

    @staticmethod
    def forward(ctx, rows, B, C, Ln):
        out = _empty(B, C, Ln)
        ld = rows.shape[1]
        check(_lib.lib().pt_cl_to_ncl(ptr(rows.contiguous()), ld, ptr(out), B, C, Ln, stream()))
        ctx.meta = (B, C, Ln, ld)
        return out

    @staticmethod
    def backward(ctx, dy):
        B, C, Ln, ld = ctx.meta
        d = _empty(B * Ln, ld)
        check(_lib.lib().pt_ncl_to_cl(ptr(dy.contiguous()), ptr(d), B, C, Ln, ld, stream()))
        return d, None, None, None


def to_rows(x_ncl, width=None):
    """ncl_to_rows for tensors that may carry a gradient."""
    ld = width or _ru(x_ncl.shape[1], 32)
    x = x_ncl.to(torch.float32)
    return NCLToRows.apply(x, ld) if x.requires_grad else ncl_to_rows(x, ld)


class PQQuantizeTrain(torch.autograd.Function):
    """ProductQuantization.forward in train mode, the differentiable part (vqvae_modules.py:470-487,536-544): nearest
    codes per sub-codebook (same fp32 search kernel as inference), commitment loss cc * mse(z, q), straight-through
    estimator.  z rows [B*W, G*D]; returns (q rows, loss, indices (B*W, G) int64)."""

    @staticmethod
    def forward(ctx, z, codebooks, commitment_cost):
        L = _lib.lib()
        rows, Cc = z.shape
        G, K, D = codebooks.shape
        q = _empty(rows, Cc)
        idx = torch.empty(rows, G, dtype=torch.int64, device=z.device)
        check(L.pt_pq_search_cl(ptr(z), Cc, ptr(codebooks), rows, G, D, K, ptr(idx), ptr(q), Cc, stream()))
        ss = _zeros(1)
        check(L.pt_train_sqdiff(ptr(z), ptr(q), z.numel(), ptr(ss), stream()))
        ctx.save_for_backward(z, q)
        ctx.cc = float(commitment_cost)
        ctx.mark_non_differentiable(idx)
        return q, ss[0] * (ctx.cc / z.numel()), idx

    @staticmethod
    def backward(ctx, dq, dloss, _didx):
        z, q = ctx.saved_tensors
        L = _lib.lib()
        dz = _empty(z.shape)
        g = dloss.reshape(1).contiguous() if dloss is not None else None
        dqc = dq.contiguous() if dq is not None else None
        check(L.pt_train_commit_bwd(ptr(dqc), ptr(z), ptr(q), ptr(g), 2.0 * ctx.cc / z.numel(), z.numel(), ptr(dz), stream()))
        return dz, None, None


def _tile_args(tile_noise, shape, need):
    """(noise pointer tensor or None, seed) for `_tile`'s noise (vqvae_modules.py:573-577): injected N(0,1) draws in the
    parity mode, else a seed for the kernel's own generator drawn from torch's CPU generator."""
    if tile_noise is not None:
        tn = tile_noise.to(device="cuda", dtype=torch.float32).contiguous()
        if tuple(tn.shape) != tuple(shape):
            raise RuntimeError(f"_tile noise must have shape {tuple(shape)}, got {tuple(tn.shape)}")
        return tn, 0
    return None, (int(torch.randint(0, 2 ** 62, (1,)).item()) if need else 0)


def pq_ema_update(z, idx, cs_hidden, dw_hidden, emb, decay, eps, counter, tile_noise=None):
    """EMA codebook update + dead-code re-seeding (vqvae_modules.py:513-534) in place on the stacked buffers
    cs_hidden [G,K], dw_hidden [G,K,D], emb [G,K,D]; z rows are the encoder output of this step, idx its codes.
    Fewer rows than codes: the seed rows are the noise-tiled batch (`_tile`); tile_noise [G,K,D] injects its N(0,1)."""
    L = _lib.lib()
    G, K, D = emb.shape
    counts, dw = _empty(G, K), _empty(G, K, D)
    tn, seed = _tile_args(tile_noise, (G, K, D), z.shape[0] < K)
    check(L.pt_train_pq_ema(ptr(z), z.shape[1], ptr(idx), z.shape[0], G, D, K, ptr(counts), ptr(dw), ptr(cs_hidden),
                            ptr(dw_hidden), ptr(emb), float(decay), float(eps), int(counter), ptr(tn), seed, stream()))


def pq_ema_update_shared(z, idx, G, cs_hidden, dw_hidden, emb, decay, eps, counter, tile_noise=None):
    """share_code=True (vqvae_modules.py:496-512): the chunks of all G groups update the ONE code book in place:
    cs_hidden [K], dw_hidden [K,D], emb [K,D]; z rows [B*W, G*D], idx (B*W, G)."""
    L = _lib.lib()
    K, D = emb.shape
    counts, dw = _empty(K), _empty(K, D)
    tn, seed = _tile_args(tile_noise, (K, D), z.shape[0] * G < K)
    check(L.pt_train_pq_ema_shared(ptr(z), z.shape[1], ptr(idx), z.shape[0], G, D, K, ptr(counts), ptr(dw), ptr(cs_hidden),
                                   ptr(dw_hidden), ptr(emb), float(decay), float(eps), int(counter), ptr(tn), seed, stream()))


def vq_ema_update(z, idx, cs_hidden, dw_hidden, emb, decay, eps, counter):
    """VectorQuantizerEMA's EMA codebook update (vqvae_modules.py:323-334, no re-seeding) in place on cs_hidden [K],
    dw_hidden [K,D], emb [K,D]; z rows [B*W, ld] are the encoder output of this step, idx (B*W, 1) its codes."""
    L = _lib.lib()
    K, D = emb.shape
    counts, dw = _empty(1, K), _empty(1, K, D)
    check(L.pt_train_vq_ema(ptr(z), z.shape[1], ptr(idx), z.shape[0], D, K, ptr(counts), ptr(dw), ptr(cs_hidden), ptr(dw_hidden),
                            ptr(emb), float(decay), float(eps), int(counter), stream()))


class VQLosses(torch.autograd.Function):
    """vq_teacher.get_loss (vq_teacher.py:184-232) without the commitment term: weighted smooth-L1 part losses +
    velocity L1.  Returns (total, sums) with sums = part means | velocity mean | total (device tensor, no gradient)."""

    @staticmethod
    def forward(ctx, pred, gt, B, Tn, bounds, weights, vel_cols, beta):
        import ctypes as C
        L = _lib.lib()
        nparts = len(weights)
        sums = _zeros(nparts + 2)
        dpred = _zeros(pred.shape)
        check(L.pt_train_vq_losses(ptr(pred), pred.shape[1], ptr(gt), gt.shape[1], B, Tn, nparts,
                                   (C.c_int * (nparts + 1))(*bounds), (C.c_float * nparts)(*weights), vel_cols, beta,
                                   ptr(sums), ptr(dpred), dpred.shape[1], stream()))
        ctx.save_for_backward(dpred)
        ctx.mark_non_differentiable(sums)
        return sums[nparts + 1].clone(), sums

    @staticmethod
    def backward(ctx, g, _gs):
        (dpred,) = ctx.saved_tensors
        return dpred * g, None, None, None, None, None, None, None


class MaskedL1Losses(torch.autograd.Function):
    """refiner.get_loss (refiner.py:313-391): per-part L1 over the unmasked frames, weighted, + velocity L1.
    nmask: (B*T,) = 1 - mask per frame row.  Returns (total, sums) like VQLosses."""

    @staticmethod
    def forward(ctx, pred, gt, nmask, t_nmask, B, Tn, bounds, weights, vel_cols):
        import ctypes as C
        L = _lib.lib()
        nparts = len(weights)
        sums = _zeros(nparts + 2)
        dpred = _zeros(pred.shape)
        check(L.pt_train_masked_l1_losses(ptr(pred), pred.shape[1], ptr(gt), gt.shape[1], ptr(nmask), float(t_nmask), B, Tn,
                                          nparts, (C.c_int * (nparts + 1))(*bounds), (C.c_float * nparts)(*weights),
                                          vel_cols, ptr(sums), ptr(dpred), dpred.shape[1], stream()))
        ctx.save_for_backward(dpred)
        ctx.mark_non_differentiable(sums)
        return sums[nparts + 1].clone(), sums

    @staticmethod
    def backward(ctx, g, _gs):
        (dpred,) = ctx.saved_tensors
        return dpred * g, None, None, None, None, None, None, None, None


# ---------------------------------------------------------------------------------------------------------------
# optimizer side: torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW semantics on HIP kernels
# ---------------------------------------------------------------------------------------------------------------
def clip_grad_norm(params, max_norm):
    """Scales the gradients' effect by min(1, max_norm / (total_norm + 1e-6)); returns (total_norm tensor, coef tensor).
    The coefficient stays on the device and is applied inside the AdamW kernel (no extra pass over the gradients)."""
    L = _lib.lib()
    ss, coef, norm = _zeros(1), _empty(1), _empty(1)
    for p in params:
        if p.grad is not None:
            g = p.grad.contiguous()
            check(L.pt_train_sumsq(ptr(g), g.numel(), ptr(ss), stream()))
    check(L.pt_train_clip_coef(ptr(ss), float(max_norm), ptr(coef), ptr(norm), stream()))
    return norm, coef


class AdamW:
    """torch.optim.AdamW (amsgrad off) on the library's kernel; state_dict()/load_state_dict() use torch's layout."""

    def __init__(self, param_groups, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.param_groups = [dict(params=list(g["params"]), lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
                             for g in param_groups]
        self.state = {}

    def zero_grad(self):
        for g in self.param_groups:
            for p in g["params"]:
                p.grad = None

    def step(self, coefs=None):
        """coefs: optional {param: device scalar} of clip coefficients."""
        L = _lib.lib()
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = self.state.get(p)
                if st is None:
                    st = self.state[p] = {"step": 0, "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}
                if not p.is_cuda:
                    raise RuntimeError("AdamW.step: parameters must be on the GPU")
                for k in ("exp_avg", "exp_avg_sq"):        # state restored before the module moved to the GPU
                    if st[k].device != p.device:
                        st[k] = st[k].to(p.device)
                st["step"] += 1
                c = None if coefs is None else coefs.get(p)
                gr = p.grad.contiguous()
                check(L.pt_train_adamw(ptr(p.data), ptr(gr), ptr(st["exp_avg"]), ptr(st["exp_avg_sq"]), p.numel(), ptr(c),
                                       g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], st["step"], stream()))

    def state_dict(self):
        idx, packed, groups = {}, {}, []
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                idx[p] = len(idx)
                ids.append(idx[p])
            groups.append({k: v for k, v in g.items() if k != "params"} | {"params": ids})
        for p, st in self.state.items():
            packed[idx[p]] = {"step": torch.tensor(float(st["step"])), "exp_avg": st["exp_avg"], "exp_avg_sq": st["exp_avg_sq"]}
        return {"state": packed, "param_groups": groups}

    def load_state_dict(self, sd):
        flat = [p for g in self.param_groups for p in g["params"]]
        for i, st in sd.get("state", {}).items():
            p = flat[int(i)]
            self.state[p] = {"step": int(float(st["step"])), "exp_avg": st["exp_avg"].to(p.device).float().clone(),
                             "exp_avg_sq": st["exp_avg_sq"].to(p.device).float().clone()}
        for g, sg in zip(self.param_groups, sd.get("param_groups", [])):
            for k in ("lr", "betas", "eps", "weight_decay"):
                if k in sg:
                    g[k] = tuple(sg[k]) if k == "betas" else sg[k]
