"""Deterministic name-keyed synthetic weights (SURVEY.md §8c golden-vector rules).

No checkpoint of the reference exists in the build or on the GPU box, so every
parity test and the benchmark run on synthetic weights.  A tensor's values
depend only on (seed, tensor name, shape): raw counter-based Philox bits mapped
to floats with numpy, never `torch.manual_seed`, so this container (where the
reference can be imported) and the GPU box regenerate identical tensors.

Zero-initialised tensors of the reference (the `zero_module` style linears,
PTransformer.py:75-79) and default BatchNorm statistics are re-randomised,
otherwise every sub-layer output is exactly 0 and parity is vacuous.
"""
import zlib

import numpy as np
import torch

_LN_TOKENS = ("norm1.", "norm2.", "norm3.", "norm_2.", "norm.")


def _rng(name, seed):
    key = (zlib.crc32(name.encode()) << 32) | (seed & 0xFFFFFFFF)
    return np.random.Generator(np.random.Philox(key=key))


def _fan_in(shape):
    n = 1
    for s in shape[1:]:
        n *= s
    return max(n, 1)


def make_tensor(name, shape, dtype=torch.float32, seed=0):
    """Value of tensor `name` (a state_dict key) under `seed`."""
    shape = tuple(int(s) for s in shape)
    g = _rng(name, seed)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    if dtype in (torch.int64, torch.int32):
        return torch.zeros(shape, dtype=dtype)
    if leaf == "running_var":
        a = g.uniform(0.5, 1.5, shape)
    elif leaf == "running_mean":
        a = g.normal(0.0, 0.1, shape)
    elif leaf.startswith("embeddings_") or leaf == "embeddings":
        # codebooks: xavier-uniform bound like the reference's initialiser
        bound = np.sqrt(6.0 / (shape[0] + shape[1]))
        a = g.uniform(-bound, bound, shape) * 4.0
    elif leaf == "hidden":
        a = g.uniform(0.0, 1.0, shape)
    elif len(shape) == 1 and leaf == "weight":
        a = g.uniform(0.8, 1.2, shape)  # BatchNorm / LayerNorm gains
    elif leaf in ("bias", "in_proj_bias"):
        a = g.uniform(-0.05, 0.05, shape)
    elif "tok_emb" in name or "class_embed" in name or "start_emb" in name:
        a = g.normal(0.0, 1.0, shape)
    elif leaf.startswith("pad_"):
        a = g.uniform(-0.1, 0.1, shape)
    elif len(shape) >= 2:
        bound = np.sqrt(3.0 / _fan_in(shape))
        if "_up_" in name and len(shape) == 3:
            # ConvTranspose1d weights are (C_in, C_out, k); two taps feed each output
            bound = np.sqrt(3.0 / max(shape[0] * shape[2] // 2, 1))
        a = g.uniform(-bound, bound, shape)
    else:
        a = g.uniform(-0.05, 0.05, shape)
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


# Tensors that keep their constructed value (sinusoid tables are part of the
# algorithm, PTransformer.py:227-245, not random state).
def is_constant(name):
    return "position_encoding" in name


def generate_state_dict(template, seed=0, prefix=""):
    """Return {key: tensor} for every key of `template` (a state_dict), leaving
    positional-encoding tables untouched."""
    out = {}
    for k, v in template.items():
        if is_constant(k):
            out[k] = v.detach().clone()
        else:
            out[k] = make_tensor(prefix + k, v.shape, v.dtype, seed)
    return out


def fill_module(module, seed=0, prefix=""):
    """Overwrite `module`'s parameters and buffers with generated values."""
    sd = generate_state_dict(module.state_dict(), seed, prefix)
    module.load_state_dict(sd)
    return module


def make_input(name, shape, kind="normal", seed=0, low=0, high=1):
    """Deterministic synthetic *input* tensor (same counter-based scheme as the weights)."""
    g = _rng("input." + name, seed)
    shape = tuple(int(s) for s in shape)
    if kind == "normal":
        return torch.from_numpy(g.normal(0.0, 1.0, shape)).to(torch.float32)
    if kind == "uniform":
        return torch.from_numpy(g.uniform(low, high, shape)).to(torch.float32)
    if kind == "int":
        return torch.from_numpy(g.integers(low, high, shape, dtype=np.int64))
    if kind == "exp":  # Exp(1) draws for the injected categorical noise
        return torch.from_numpy(g.exponential(1.0, shape)).to(torch.float32)
    if kind == "gumbel":
        return torch.from_numpy(g.gumbel(0.0, 1.0, shape)).to(torch.float32)
    raise ValueError(kind)
