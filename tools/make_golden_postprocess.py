"""Golden fixture of the pose post-process from the IMPORTED REFERENCE functions (build container only):
data_utils/rotation_conversion.py (rotation_6d_to_matrix, matrix_to_axis_angle) and data_utils/lower_body.py (c_index_6d,
part2full, lower_pose), applied in the order of scripts/demo.py:183-205.  nets/utils.py (denormalize / tofbhe) imports
`textgrid`, which this image lacks, so those two steps (a gather of the statistics and one multiply-add) are the oracle's
restatement; the statistics are the reference's own data file data_utils/norm_stats.npy, read with numpy's non-pickle
loader and stored in the fixture as data.  Data only.

    python tools/make_golden_postprocess.py
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("PROBTALK_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden", "postprocess.npz")


def edge_case_6d():
    """6-D forms (first two ROWS of the matrix, rotation_conversion.py:533) of rotations the formula treats specially:
    identity and near-identity (small-angle branch), angles close to pi (1 + trace -> 0), unnormalised / skewed inputs."""
    from oracle.postprocess import matrix_to_axis_angle  # noqa: F401  (oracle and reference are compared in the test)
    rows = []
    axes = torch.nn.functional.normalize(torch.tensor([[1.0, 0, 0], [0, 1.0, 0], [0.3, -0.5, 0.8], [-0.6, 0.64, 0.48]]), dim=-1)
    for ang in (0.0, 1e-8, 3e-7, 1e-5, 1e-3, 0.5, 2.0, 3.0, math.pi - 1e-1, math.pi - 1e-2, math.pi - 1e-3, math.pi):
        for ax in axes:
            K = torch.tensor([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]], dtype=torch.float64)
            R = torch.eye(3, dtype=torch.float64) + math.sin(ang) * K + (1 - math.cos(ang)) * (K @ K)
            rows.append(R[:2].reshape(6).to(torch.float32))
    d = torch.stack(rows)
    # unnormalised and skewed variants of the same rotations (Gram-Schmidt has to fix them)
    g = torch.Generator().manual_seed(3)
    skew = d.clone()
    skew[:, :3] *= 0.2 + 3 * torch.rand(d.shape[0], 1, generator=g)
    skew[:, 3:] = skew[:, 3:] * (0.5 + torch.rand(d.shape[0], 1, generator=g)) + 0.3 * skew[:, :3]
    return torch.cat([d, skew])


def main():
    # data_utils/__init__.py pulls the data loaders (librosa, textgrid ...): register the package as a bare namespace so
    # that only the two modules named here are executed (the same device tools/ref_loader.py uses for `nets`)
    import importlib.machinery
    import types
    pkg = types.ModuleType("data_utils")
    pkg.__path__ = [os.path.join(REF, "data_utils")]
    pkg.__spec__ = importlib.machinery.ModuleSpec("data_utils", None, is_package=True)
    sys.modules["data_utils"] = pkg
    from data_utils.lower_body import c_index_6d, part2full
    from data_utils.rotation_conversion import matrix_to_axis_angle, rotation_6d_to_matrix
    from oracle import postprocess as OP
    assert np.array_equal(c_index_6d, OP.c_index())
    stats = np.load(os.path.join(REF, "data_utils", "norm_stats.npy"))          # (2, 1, 1, 430) float32, no pickle
    assert stats.shape == (2, 1, 1, 430) and stats.dtype == np.float32
    g = torch.Generator().manual_seed(11)
    B, T = 3, 32
    pred = torch.randn(B, 376, T, generator=g)                                  # network output in normalised units
    ns = torch.from_numpy(stats)
    den = OP.denormalize(pred, ns[0], ns[1], c_index_6d)
    # plant the edge cases into clip 2 AFTER denormalisation (so they reach the rotation code exactly), and store the
    # normalised values that produce them
    edge = edge_case_6d()                                                       # (96, 6)
    n_e = edge.shape[0]
    flat = den[2].transpose(0, 1)[:, :276].reshape(T * 46, 6)
    flat[:n_e] = edge
    den[2, :276] = flat.reshape(T, 276).transpose(0, 1)
    outs = []
    for b in range(B):
        p = den[b].transpose(0, 1)
        face, p = p[:, -100:], p[:, :-100]
        p = matrix_to_axis_angle(rotation_6d_to_matrix(p.reshape(p.shape[0], -1, 6))).reshape(p.shape[0], -1)
        outs.append(part2full(torch.cat([p, face], dim=-1), False))
    out = torch.stack(outs)
    assert tuple(out.shape) == (B, T, 265) and bool(torch.isfinite(out).all())
    np.savez_compressed(OUT, denormalized=den.numpy(), norm_stats=stats, out=out.numpy(), n_edge=np.int64(n_e))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
