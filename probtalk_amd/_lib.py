"""ctypes binding of libprobtalk_hip.so (C ABI: include/probtalk_hip.h).

The product path has no CPU fallback: if the shared library is missing or fails
to load, every entry point raises.  PyTorch is used only for device memory,
the current stream and (elsewhere) torch.distributed.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PROBTALK_HIP_LIB") or os.path.join(_HERE, "libprobtalk_hip.so")   # env: kernel A/B experiments

_lib = None

c_fp = C.c_void_p  # device pointers travel as void*


def _proto(lib):
    i, sz, vp, f = C.c_int, C.c_size_t, C.c_void_p, C.c_float
    lib.pt_last_error.restype = C.c_char_p
    lib.pt_version.restype = i
    lib.pt_model_create.restype = vp
    lib.pt_model_create.argtypes = [C.c_char_p, C.POINTER(C.c_int), i]
    lib.pt_model_set_tensor.argtypes = [vp, C.c_char_p, vp, C.POINTER(C.c_int64), i, i]
    lib.pt_model_finalize.argtypes = [vp]
    lib.pt_model_set_precision.argtypes = [vp, i]
    lib.pt_model_destroy.argtypes = [vp]
    lib.pt_model_destroy.restype = None
    for name in ("pt_vqvae_workspace_bytes", "pt_cond_encoder_workspace_bytes", "pt_predictor_workspace_bytes",
                 "pt_refiner_workspace_bytes"):
        getattr(lib, name).restype = sz
        getattr(lib, name).argtypes = [vp, i, i]
    lib.pt_vqvae_encode.argtypes = [vp, vp, i, i, vp, vp, vp, vp, sz, vp]
    lib.pt_vqvae_decode.argtypes = [vp, vp, i, i, vp, vp, sz, vp]
    lib.pt_vqvae_decode_codes.argtypes = [vp, vp, i, i, i, vp, vp, sz, vp]
    lib.pt_pq_quantize.argtypes = [vp, vp, i, i, vp, vp, vp, sz, vp]
    lib.pt_pq_quantize_static.argtypes = [vp, vp, i, i, vp, vp, vp]
    lib.pt_pq_quantize_all.argtypes = [vp, vp, i, i, vp, vp]
    lib.pt_cond_encoder_forward.argtypes = [vp, vp, i, i, vp, vp, sz, vp]
    lib.pt_predictor_condition.argtypes = [vp, vp, vp, vp, i, i, vp, vp, sz, vp]
    lib.pt_predictor_logits.argtypes = [vp, vp, vp, vp, i, i, vp, vp, sz, vp]
    lib.pt_predictor_sample.argtypes = [vp, vp, vp, vp, vp, i, i, i, vp, vp, vp, sz, vp]
    lib.pt_predictor_sample_seeded.argtypes = [vp, vp, vp, C.c_uint64, i, i, i, vp, vp, vp, sz, vp]
    lib.pt_predictor_sample_seed_dev.argtypes = [vp, vp, vp, vp, i, i, i, vp, vp, vp, sz, vp]
    lib.pt_maskgit_noise.argtypes = [C.c_uint64, i, i, i, i, vp, vp, vp]
    lib.pt_maskgit_noise_from_bits.argtypes = [vp, i, vp, vp, vp, vp]
    lib.pt_refiner_forward.argtypes = [vp, vp, vp, vp, vp, i, i, i, vp, vp, sz, vp]
    lib.pt_interp_blend.argtypes = [vp, i, i, i, i, vp, vp, i, vp, vp]
    lib.pt_conv1d_cl.argtypes = [vp, i, vp, vp, i, i, i, i, i, i, vp, i, i, f, vp, i, vp]
    lib.pt_ncl_to_cl.argtypes = [vp, vp, i, i, i, i, vp]
    lib.pt_cl_to_ncl.argtypes = [vp, i, vp, i, i, i, vp]
    lib.pt_pq_search_cl.argtypes = [vp, i, vp, i, i, i, i, vp, vp, i, vp]
    lib.pt_pq_gather_cl.argtypes = [vp, i, vp, i, i, i, i, vp, i, vp]
    lib.pt_prof_enable.argtypes = [i]
    lib.pt_prof_read.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.pt_gemm_bf16.argtypes = [vp, i, vp, vp, i, i, i, i, vp, i, vp, i, i, vp]
    lib.pt_gemm_bf16_lnmod.argtypes = [vp, i, vp, vp, i, i, i, vp, vp, i, i, i, vp, vp]
    lib.pt_gemm_bf16_addln.argtypes = [vp, i, vp, vp, i, i, i, vp, vp, vp, vp, vp]
    lib.pt_gemm_bf16_sublayer.argtypes = [vp, i, i, vp, vp, vp, vp, i, i, i, vp, vp, vp, vp, vp, i, vp]
    lib.pt_permute_k32.argtypes = [vp, vp, i, i, vp]
    lib.pt_conv1d_cl_bf16.argtypes = [vp, i, vp, vp, i, i, i, i, i, i, vp, i, vp, i, i, C.c_float, vp, i, vp]
    lib.pt_postprocess_poses.argtypes = [vp, i, i, vp, vp, vp, i, vp, vp, i, vp]
    f, u64, sz = C.c_float, C.c_uint64, C.c_size_t
    lib.pt_embed_pe.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, vp]
    lib.pt_train_wgrad.argtypes = [vp, i, vp, i, vp, i, i, i, i, i, vp]
    lib.pt_train_wgrad_bias.argtypes = [vp, i, vp, i, vp, vp, i, i, i, i, i, vp]
    lib.pt_train_colsum.argtypes = [vp, i, vp, i, i, i, vp, vp]
    lib.pt_train_ln_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i, vp]
    lib.pt_train_ln_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i, vp]
    lib.pt_train_modsilu_fwd.argtypes = [vp, vp, vp, i, i, i, vp, i, vp]
    lib.pt_train_modsilu_bwd.argtypes = [vp, vp, vp, vp, i, i, i, vp, vp, i, vp]
    lib.pt_train_attention_fwd.argtypes = [vp, i, vp, i, vp, i, vp, i, i, i, i, i, f, u64, vp]
    lib.pt_train_attention_bwd.argtypes = [vp, i, vp, i, vp, i, vp, i, vp, i, vp, i, vp, i, vp, vp, i, i, i, i, f, u64, vp]
    lib.pt_train_cross_entropy.argtypes = [vp, vp, i, i, vp, vp, vp]
    lib.pt_train_bn_fwd.argtypes = [vp, i, i, i, vp, vp, vp, i, f, vp, i, vp, vp, vp, vp, f, vp]
    lib.pt_train_bn_bwd.argtypes = [vp, i, vp, i, f, vp, i, vp, vp, vp, i, vp, i, vp, i, i, vp]
    lib.pt_train_act_bwd.argtypes = [vp, i, vp, i, f, vp, i, i, i, vp]
    lib.pt_train_gate_fwd.argtypes = [vp, i, i, vp, vp]
    lib.pt_train_gate_bwd.argtypes = [vp, vp, i, i, vp, vp]
    lib.pt_train_embed_bwd.argtypes = [vp, vp, i, i, vp, vp]
    lib.pt_train_dropout.argtypes = [vp, vp, sz, f, u64, vp]
    lib.pt_train_sumsq.argtypes = [vp, sz, vp, vp]
    lib.pt_train_clip_coef.argtypes = [vp, f, vp, vp, vp]
    lib.pt_train_adamw.argtypes = [vp, vp, vp, vp, sz, vp, f, f, f, f, f, i, vp]
    lib.pt_train_add.argtypes = [vp, vp, vp, sz, vp]
    lib.pt_train_pq_ema.argtypes = [vp, i, vp, i, i, i, i, vp, vp, vp, vp, vp, f, f, i, vp, C.c_uint64, vp]
    lib.pt_train_pq_ema_shared.argtypes = lib.pt_train_pq_ema.argtypes
    lib.pt_train_vq_ema.argtypes = [vp, i, vp, i, i, i, vp, vp, vp, vp, vp, f, f, i, vp]
    lib.pt_train_vq_losses.argtypes = [vp, i, vp, i, i, i, i, vp, vp, i, f, vp, vp, i, vp]
    lib.pt_train_masked_l1_losses.argtypes = [vp, i, vp, i, vp, f, i, i, i, vp, vp, i, vp, vp, i, vp]
    lib.pt_train_sqdiff.argtypes = [vp, vp, sz, vp, vp]
    lib.pt_train_commit_bwd.argtypes = [vp, vp, vp, vp, f, sz, vp, vp]
    lib.pt_set_option.argtypes = [C.c_char_p, i]
    lib.pt_get_option.argtypes = [C.c_char_p, vp]
    lib.pt_attention_cl.argtypes = [vp, i, vp, i, vp, i, vp, i, i, i, i, i, vp]
    lib.pt_attention_bf16.argtypes = [vp, i, vp, i, vp, i, vp, i, i, i, i, i, vp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int and name != "pt_version":
            fn.restype = C.c_int


EXPORTS = [
    "pt_last_error", "pt_version", "pt_model_create", "pt_model_set_tensor", "pt_model_finalize", "pt_model_destroy",
    "pt_model_set_precision",
    "pt_vqvae_workspace_bytes", "pt_vqvae_encode", "pt_vqvae_decode", "pt_vqvae_decode_codes", "pt_pq_quantize", "pt_pq_quantize_static",
    "pt_pq_quantize_all", "pt_cond_encoder_workspace_bytes", "pt_cond_encoder_forward",
    "pt_predictor_workspace_bytes", "pt_predictor_condition", "pt_predictor_logits", "pt_predictor_sample", "pt_predictor_sample_seeded", "pt_predictor_sample_seed_dev", "pt_maskgit_noise", "pt_maskgit_noise_from_bits",
    "pt_refiner_workspace_bytes", "pt_refiner_forward", "pt_interp_blend", "pt_conv1d_cl", "pt_ncl_to_cl",
    "pt_cl_to_ncl", "pt_pq_search_cl", "pt_pq_gather_cl", "pt_attention_cl", "pt_attention_bf16", "pt_gemm_bf16", "pt_gemm_bf16_lnmod",
    "pt_gemm_bf16_addln", "pt_gemm_bf16_sublayer", "pt_permute_k32", "pt_conv1d_cl_bf16", "pt_postprocess_poses", "pt_prof_enable",
    "pt_prof_read", "pt_embed_pe", "pt_train_wgrad", "pt_train_wgrad_bias", "pt_train_colsum", "pt_train_ln_fwd", "pt_train_ln_bwd",
    "pt_train_modsilu_fwd", "pt_train_modsilu_bwd", "pt_train_attention_fwd", "pt_train_attention_bwd",
    "pt_train_cross_entropy", "pt_train_bn_fwd", "pt_train_bn_bwd", "pt_train_act_bwd", "pt_train_gate_fwd",
    "pt_train_gate_bwd", "pt_train_embed_bwd", "pt_train_dropout", "pt_train_sumsq", "pt_train_clip_coef",
    "pt_train_adamw", "pt_train_add", "pt_train_pq_ema", "pt_train_pq_ema_shared", "pt_train_vq_ema", "pt_train_vq_losses", "pt_train_masked_l1_losses", "pt_train_sqdiff", "pt_train_commit_bwd",
    "pt_set_option", "pt_get_option",
]


def lib():
    """The loaded library; raises (never falls back) if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(probtalk_amd has no CPU fallback)")
        _lib = C.CDLL(LIB_PATH)
        _proto(_lib)
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError("probtalk_hip: " + lib().pt_last_error().decode())


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev_f32(t, device=None):
    """Contiguous fp32 tensor on the GPU (the reference passes fp32 CUDA tensors already)."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class Handle:
    """Owner of one pt_model plus a growing device workspace."""

    def __init__(self, kind, cfg, state_dict, prefix="", precision="fp32"):
        L = lib()
        if not torch.cuda.is_available():
            raise RuntimeError("probtalk_amd needs a HIP device (no CPU fallback)")
        arr = (C.c_int * len(cfg))(*[int(c) for c in cfg])
        self.kind = kind
        self.h = L.pt_model_create(kind.encode(), arr, len(cfg))
        if not self.h:
            raise RuntimeError("probtalk_hip: " + L.pt_last_error().decode())
        for k, v in state_dict.items():
            if not torch.is_tensor(v) or not v.dtype.is_floating_point:
                continue
            t = v.detach().cpu().contiguous()
            if t.dtype == torch.float64:
                dt = 1
            else:
                t, dt = t.to(torch.float32), 0
            shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
            check(L.pt_model_set_tensor(self.h, (prefix + k).encode(), C.c_void_p(t.data_ptr()), shape, t.dim(), dt))
        if precision not in ("fp32", "bf16"):
            raise ValueError(f"precision must be 'fp32' or 'bf16', got {precision!r}")
        if precision == "bf16":
            check(L.pt_model_set_precision(self.h, 1))
        check(L.pt_model_finalize(self.h))
        self.precision = precision
        self._ws = None

    def workspace(self, nbytes):
        """Scratch arena of the library calls on this model, one per HIP stream: calls on one stream run in order and
        share it, calls on different streams (probtalk_amd.synthetic.holistic_step(streams=...)) must not."""
        key = torch.cuda.current_stream().cuda_stream
        pool = self.__dict__.setdefault("_ws_by_stream", {})
        ws = pool.get(key)
        if ws is None or ws.numel() < nbytes:
            pool[key] = None
            for k in list(pool)[:-8]:          # streams that are gone: their arenas go back to the allocator
                if k != key:
                    del pool[k]
            ws = pool[key] = torch.empty(int(nbytes), dtype=torch.uint8, device="cuda")
        self._ws = ws
        return ws

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().pt_model_destroy(self.h)
                self.h = None
        except Exception:
            pass
