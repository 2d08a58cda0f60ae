"""CPU checks of the boundary: C-ABI exports, state_dict key parity with the reference, hygiene."""
import os
import re
import subprocess

import pytest
import torch

from tests.helpers import manifest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "probtalk_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    so = os.path.join(ROOT, "probtalk_amd", "libprobtalk_hip.so")
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    have = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    decl = _declared()
    assert len(decl) >= 25
    missing = [s for s in decl if s not in have]
    assert not missing, missing
    from probtalk_amd import _lib
    assert sorted(_lib.EXPORTS) == decl, "ctypes prototype list and header disagree"


def test_product_never_touches_the_oracle_or_reference():
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "probtalk_amd")):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, fn)).read()
                if re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M) or "/root/reference" in txt.replace(
                        "relative to /root/reference", ""):
                    bad.append(fn)
    assert not bad, bad


def _check_keys(module, name):
    sd = module.state_dict()
    man = manifest()[name]
    assert set(sd) == set(man), sorted(set(sd) ^ set(man))[:10]
    for k, v in sd.items():
        assert list(v.shape) == man[k][0] and str(v.dtype) == "torch." + man[k][1], k


def test_state_dict_keys_match_reference():
    from probtalk_amd.nets.modules import VQVAE, VQVAE_SC, ConditionEncoder
    from probtalk_amd.nets.transformer_modules import PredictorNet, RTransformer
    _check_keys(VQVAE_SC(376, 512, 128, 1024, 2, 512, q_type="pro", groups=4), "VQ")
    _check_keys(VQVAE(270, 64, 128, 1024, 2, 512), "g")
    _check_keys(ConditionEncoder(768, 256, 2, 256), "AudEnc")
    pn = PredictorNet(False, 376, 128, 512, 10, 6, 4, groups=4, identity=True, maskgit=True, maskgit_T=8,
                      transformer=True, text=None, audio="wav2vec", motion_context=True)
    _check_keys(pn, "Predictor")
    assert sum(p.numel() for p in pn.parameters()) == manifest()["_param_counts"]["Predictor"]
    _check_keys(RTransformer(376, 768, 512, 6, use_label=True, motion_context=True), "HFTransformer")


def test_weight_generator_is_deterministic_and_name_keyed():
    from probtalk_amd.weights import make_tensor
    a = make_tensor("VQ.encoder.project.conv.weight", (256, 376, 3))
    b = make_tensor("VQ.encoder.project.conv.weight", (256, 376, 3))
    c = make_tensor("VQ.encoder.project.conv.bias", (256,))
    assert torch.equal(a, b) and a.dtype == torch.float32
    assert float(a.abs().max()) > 0 and not torch.equal(a.flatten()[:256], c)
    assert float(make_tensor("x.norm.running_var", (64,)).min()) >= 0.5


def test_no_cpu_fallback_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from probtalk_amd.nets.modules import VQVAE_SC
    vq = VQVAE_SC(376, 512, 128, 1024, 2, 512, q_type="pro", groups=4).eval()
    with pytest.raises(RuntimeError):
        vq.encode(torch.zeros(1, 376, 88))
