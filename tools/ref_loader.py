"""Loader for the upstream reference modules (this container only).

Test infrastructure: imports the reference's hot-path modules from
/root/reference on CPU without executing `nets/__init__.py` (which pulls every
model family and therefore librosa/smplx/textgrid...).  Nothing here is used
by the product path; the reference never travels to the GPU box.

Shims, none of which modifies a reference file:
 1. empty namespace packages `nets`, `nets.spg`, `nets.inpainting`;
 2. a stub `torchvision` (vqvae_modules.py:6 imports it and never uses it);
 3. `Tensor.to("cuda")` -> cpu while importing (PTransformer.py:16 moves a
    constant to "cuda" at module scope);
 4. `nn.TransformerDecoder.forward` replaced by the torch-1.12 plain layer
    loop (the reference pins torch 1.12.1; torch>=2 passes tgt_is_causal=,
    which DeepSupervisionLayer.forward rejects).
"""
import importlib
import importlib.machinery
import os
import sys
import types

REF = os.environ.get("PROBTALK_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF, "nets", "inpainting"))


def load():
    import torch
    import torch.nn as nn
    import transformers  # noqa: F401  (must precede the torchvision stub)

    if "nets.inpainting.predictornet" in sys.modules:
        return sys.modules["nets"]
    for name, sub in (("nets", "nets"), ("nets.spg", "nets/spg"), ("nets.inpainting", "nets/inpainting")):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, sub)]
        m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
        sys.modules[name] = m
    for name in ("torchvision", "torchvision.datasets", "torchvision.transforms"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__spec__ = importlib.machinery.ModuleSpec(name, None)
            m.__path__ = []
            sys.modules[name] = m
    sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]

    orig_to = torch.Tensor.to

    def to_cpu(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, str) and x.startswith("cuda")) else x for x in a)
        return orig_to(self, *a, **k)

    torch.Tensor.to = to_cpu
    try:
        for mod in ("nets.spg.vqvae_modules", "nets.inpainting.vqvae_1d_sc", "nets.inpainting.PTransformer",
                    "nets.inpainting.predictornet", "nets.inpainting.RTransformer", "nets.spg.vqvae_1d"):
            importlib.import_module(mod)
    finally:
        torch.Tensor.to = orig_to

    def plain_decoder_forward(self, tgt, memory, tgt_mask=None, memory_mask=None,
                              tgt_key_padding_mask=None, memory_key_padding_mask=None, **_):
        out = tgt
        for layer in self.layers:
            out = layer(out, memory, tgt_mask=tgt_mask, memory_mask=memory_mask,
                        tgt_key_padding_mask=tgt_key_padding_mask,
                        memory_key_padding_mask=memory_key_padding_mask)
        if self.norm is not None:
            out = self.norm(out)
        return out

    nn.TransformerDecoder.forward = plain_decoder_forward
    return sys.modules["nets"]
