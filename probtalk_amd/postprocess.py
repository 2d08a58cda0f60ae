"""Pose post-process of generated motion on the GPU: the host-side mirror of scripts/demo.py:183-205 (denormalize,
rotation_6d_to_matrix, matrix_to_axis_angle, part2full) over libprobtalk_hip.so's pt_postprocess_poses.

The reference does this per sample on the CPU after `.to('cpu')`; here the whole batch stays on the GPU (one kernel) and
only the (B, T, 265) result is copied if the caller wants it on the host.  Constants restated from the reference:
data_utils/lower_body.py:4-9 (lower_pose), :43-63 (c_index_3d / c_index_6d)."""
import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream

exp_dim = 100

lower_pose = torch.tensor(
    [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 3.0747, -0.0158, -0.0152, -1.1826512813568115, 0.23866955935955048, 0.15146760642528534,
     -1.2604516744613647, -0.3160211145877838, -0.1603458970785141, 1.1654603481292725, 0.0, 0.0,
     1.2521806955337524, 0.041598282754421234, -0.06312154978513718, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])

_fix = set(list(range(18)) + [21, 22, 23, 24, 25, 26, 30, 31, 32, 33, 34, 35])
c_index_3d = np.asarray([i for i in range(165) if i not in _fix])
c_index_6d = np.asarray([v for i in c_index_3d for v in (2 * i, 2 * i + 1)])

_dev_cache = {}


def _on(device, key, make):
    k = (key, str(device))
    if k not in _dev_cache:
        _dev_cache[k] = make().to(device)
    return _dev_cache[k]


def poses_to_axis_angle(pred, norm_stats=None, c_index=None, full=True, stand=False):
    """pred (B, 376, T) fp32 CUDA, as `infer_on_batch` returns it -> (B, T, 265) SMPL-X pose rows (axis-angle of jaw,
    the fixed lower body, body and hands, then the 100 expression coefficients); (B, T, 238) with full=False (no
    part2full).  norm_stats: None when Data.pose.normalization is off, else the (2, 1, 1, 430) mean/std array of
    data_utils/norm_stats.npy (tensor or ndarray)."""
    if not (torch.is_tensor(pred) and pred.is_cuda and pred.dim() == 3 and pred.shape[1] == 376):
        raise RuntimeError("poses_to_axis_angle: pred must be a CUDA tensor of shape (B, 376, T) (vq_type 'fbhe', 6-D poses)")
    if stand:
        raise NotImplementedError("stand=True (the standing lower body) is not used by scripts/demo.py's defaults")
    x = pred.detach().to(torch.float32).contiguous()
    B, _, T = x.shape
    dev = x.device
    mean = std = cidx = None
    n_stats = 0
    if norm_stats is not None:
        ns = torch.as_tensor(norm_stats, dtype=torch.float32)
        if ns.shape[0] != 2 or ns[0].numel() != 430:
            raise RuntimeError(f"poses_to_axis_angle: norm_stats must be (2, ..., 430) = mean, std; got {tuple(ns.shape)}")
        ns = ns.reshape(2, -1).to(dev).contiguous()
        mean, std, n_stats = ns[0], ns[1], ns.shape[1]
        ci = c_index_6d if c_index is None else np.asarray(c_index)
        if ci.shape != (270,) or ci.min() < 0 or ci.max() >= n_stats:
            raise RuntimeError("poses_to_axis_angle: c_index must hold 270 indices into the statistics")
        cidx = _on(dev, ("cidx", ci.tobytes()), lambda: torch.from_numpy(ci.astype(np.int32)))
    lp = _on(dev, "lower", lambda: lower_pose.clone()) if full else None
    W = 265 if full else 238
    out = torch.empty(B, T, W, device=dev)
    check(_lib.lib().pt_postprocess_poses(ptr(x), B, T, ptr(mean), ptr(std), ptr(cidx), n_stats, ptr(lp), ptr(out), W, stream()))
    return out
