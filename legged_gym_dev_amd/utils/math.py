"""Torch helpers with the reference's names (legged_gym/utils/math.py:38-56 and the
isaacgym.torch_utils functions the env used).  Not on the device hot path
(the kernels carry their own copies); kept for user code that imports them."""
import numpy as np
import torch


def normalize(x, eps: float = 1e-9):
    return x / x.norm(p=2, dim=-1).clamp(min=eps).unsqueeze(-1)


def quat_apply(q, v):
    shape = v.shape
    q, v = q.reshape(-1, 4), v.reshape(-1, 3)
    xyz = q[:, :3]
    t = 2 * torch.cross(xyz, v, dim=-1)
    return (v + q[:, 3:] * t + torch.cross(xyz, t, dim=-1)).view(shape)


def quat_rotate_inverse(q, v):
    w, xyz = q[:, -1:], q[:, :3]
    return v * (2.0 * w ** 2 - 1.0) - 2.0 * w * torch.cross(xyz, v, dim=-1) \
        + 2.0 * xyz * (xyz * v).sum(-1, keepdim=True)


def quat_apply_yaw(quat, vec):
    qy = quat.clone().view(-1, 4)
    qy[:, :2] = 0.0
    return quat_apply(normalize(qy), vec)


def wrap_to_pi(angles):
    angles %= 2 * np.pi
    angles -= 2 * np.pi * (angles > np.pi)
    return angles


def torch_rand_float(lower, upper, shape, device):
    return (upper - lower) * torch.rand(*shape, device=device) + lower


def torch_rand_sqrt_float(lower, upper, shape, device):
    r = 2 * torch.rand(*shape, device=device) - 1
    r = torch.where(r < 0.0, -torch.sqrt(-r), torch.sqrt(r))
    return (upper - lower) * (r + 1.0) / 2.0 + lower
