"""Small torch helpers with the semantics of reference legged_gym/utils/math.py:38-55 and of the
isaacgym.torch_utils functions the env relies on.  Host-side conveniences for
scripts; the hot path computes the same quantities inside the HIP step kernel."""
import numpy as np
import torch


def normalize(x, eps: float = 1e-9):
    return x / x.norm(p=2, dim=-1).clamp(min=eps, max=None).unsqueeze(-1)


def quat_apply(a, b):
    shape = b.shape
    a = a.reshape(-1, 4)
    b = b.reshape(-1, 3)
    xyz = a[:, :3]
    t = torch.cross(xyz, b, dim=-1) * 2
    return (b + a[:, 3:] * t + torch.cross(xyz, t, dim=-1)).view(shape)


def _quat_rot(q, v, sign):
    q_w, q_vec = q[:, -1], q[:, :3]
    a = v * (2.0 * q_w ** 2 - 1.0).unsqueeze(-1)
    b = torch.cross(q_vec, v, dim=-1) * q_w.unsqueeze(-1) * 2.0
    c = q_vec * (q_vec * v).sum(-1, keepdim=True) * 2.0
    return a + sign * b + c


def quat_rotate(q, v):
    return _quat_rot(q, v, 1.0)


def quat_rotate_inverse(q, v):
    return _quat_rot(q, v, -1.0)


def torch_rand_float(lower, upper, shape, device):
    return (upper - lower) * torch.rand(*shape, device=device) + lower


def quat_apply_yaw(quat, vec):
    quat_yaw = quat.clone().view(-1, 4)
    quat_yaw[:, :2] = 0.
    return quat_apply(normalize(quat_yaw), vec)


def wrap_to_pi(angles):
    """In place, like the reference (math.py:45-48)."""
    angles %= 2 * np.pi
    angles -= 2 * np.pi * (angles > np.pi)
    return angles


def torch_rand_sqrt_float(lower, upper, shape, device):
    r = 2 * torch.rand(*shape, device=device) - 1
    r = torch.where(r < 0., -torch.sqrt(-r), torch.sqrt(r))
    return (upper - lower) * (r + 1.) / 2. + lower
