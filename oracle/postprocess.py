"""CPU oracle of the pose post-process behind the hot path (TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline may import this; the product path is probtalk_amd/postprocess.py on the HIP kernel).

Restates, in torch fp32 with the reference's operation order:
  nets/utils.py:16-43                         tofbhe / denormalize ('fbhe': jaw | c_index body+hands | expression statistics)
  data_utils/rotation_conversion.py:512-533   rotation_6d_to_matrix (Gram-Schmidt, F.normalize eps 1e-12)
  data_utils/rotation_conversion.py:86-119    _sqrt_positive_part, _copysign, matrix_to_quaternion
  data_utils/rotation_conversion.py:481-509   quaternion_to_axis_angle (|angle| < 1e-6 -> 0.5 - angle^2 / 48)
  data_utils/lower_body.py:4-9,43-63,66-83    lower_pose, c_index_3d / c_index_6d, part2full
  scripts/demo.py:183-205                     the call sequence per generated sample
Pinned by tests/golden/postprocess.npz, which tools/make_golden_postprocess.py produced from the imported reference
functions (rotation_conversion.py and lower_body.py import with torch alone; nets/utils.py needs `textgrid`, which is
absent, so denormalize/tofbhe are pinned through this restatement only: a gather and one multiply-add)."""
import numpy as np
import torch

EXP_DIM = 100

# data_utils/lower_body.py:4-9 (axis-angle of the 9 lower-body joints kept fixed)
LOWER_POSE = torch.tensor(
    [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 3.0747, -0.0158, -0.0152, -1.1826512813568115, 0.23866955935955048, 0.15146760642528534,
     -1.2604516744613647, -0.3160211145877838, -0.1603458970785141, 1.1654603481292725, 0.0, 0.0,
     1.2521806955337524, 0.041598282754421234, -0.06312154978513718, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])


def c_index(six_d=True):
    """lower_body.py:43-63: the 135 axis-angle scalars kept out of 165 (as 270 6-D scalars when six_d)."""
    fix = set(list(range(18)) + [21, 22, 23, 24, 25, 26, 30, 31, 32, 33, 34, 35])
    keep = [i for i in range(165) if i not in fix]
    return np.asarray([v for i in keep for v in (2 * i, 2 * i + 1)] if six_d else keep)


def tofbhe(kps, cidx):
    jaw_dim = 3 if cidx.shape[0] == 129 else 6
    return torch.cat([kps[:, :jaw_dim, :], kps[:, cidx, :], kps[:, -EXP_DIM:, :]], 1)


def denormalize(kps, data_mean, data_std, cidx):
    """kps (B, 376, T); statistics of the full 430-channel pose vector, any shape with 430 elements."""
    std = tofbhe(data_std.reshape(1, -1, 1), cidx)
    mean = tofbhe(data_mean.reshape(1, -1, 1), cidx)
    return kps * std + mean


def rotation_6d_to_matrix(d6):
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = torch.nn.functional.normalize(a1, dim=-1)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = torch.nn.functional.normalize(b2, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


def _sqrt_positive_part(x):
    return torch.where(x > 0, torch.sqrt(torch.clamp_min(x, 0)), torch.zeros_like(x))


def _copysign(a, b):
    return torch.where((a < 0) != (b < 0), -a, a)


def matrix_to_quaternion(m):
    m00, m11, m22 = m[..., 0, 0], m[..., 1, 1], m[..., 2, 2]
    o0 = 0.5 * _sqrt_positive_part(1 + m00 + m11 + m22)
    x = 0.5 * _sqrt_positive_part(1 + m00 - m11 - m22)
    y = 0.5 * _sqrt_positive_part(1 - m00 + m11 - m22)
    z = 0.5 * _sqrt_positive_part(1 - m00 - m11 + m22)
    o1 = _copysign(x, m[..., 2, 1] - m[..., 1, 2])
    o2 = _copysign(y, m[..., 0, 2] - m[..., 2, 0])
    o3 = _copysign(z, m[..., 1, 0] - m[..., 0, 1])
    return torch.stack((o0, o1, o2, o3), -1)


def quaternion_to_axis_angle(q):
    norms = torch.norm(q[..., 1:], p=2, dim=-1, keepdim=True)
    half = torch.atan2(norms, q[..., :1])
    ang = 2 * half
    small = ang.abs() < 1e-6
    s = torch.where(small, 0.5 - (ang * ang) / 48, torch.sin(half) / torch.where(small, torch.ones_like(ang), ang))
    return q[..., 1:] / s


def matrix_to_axis_angle(m):
    return quaternion_to_axis_angle(matrix_to_quaternion(m))


def part2full(x, lower=LOWER_POSE):
    lp = lower.to(x.dtype).unsqueeze(0).repeat(x.shape[0], 1)
    return torch.cat([x[:, :3], lp[:, :15], x[:, 3:6], lp[:, 15:21], x[:, 6:9], lp[:, 21:27], x[:, 9:]], dim=1)


def postprocess(pred, norm_stats=None, cidx=None, full=True):
    """scripts/demo.py:183-205 for a batch: pred (B, 376, T) network output -> (B, T, 265) SMPL-X axis-angle pose
    (jaw 3 | 27 fixed lower-body | ... 135 body+hands) + 100 expression coefficients; (B, T, 238) when full is False.
    norm_stats: None (Data.pose.normalization off) or an array/tensor of shape (2, ..., 430) = mean, std."""
    if norm_stats is not None:
        ns = torch.as_tensor(norm_stats, dtype=torch.float32)
        pred = denormalize(pred, ns[0], ns[1], c_index() if cidx is None else cidx)
    outs = []
    for b in range(pred.shape[0]):
        p = pred[b].transpose(0, 1)
        face, p = p[:, -EXP_DIM:], p[:, :-EXP_DIM]
        p = matrix_to_axis_angle(rotation_6d_to_matrix(p.reshape(p.shape[0], -1, 6))).reshape(p.shape[0], -1)
        p = torch.cat([p, face], dim=-1)
        outs.append(part2full(p) if full else p)
    return torch.stack(outs)
