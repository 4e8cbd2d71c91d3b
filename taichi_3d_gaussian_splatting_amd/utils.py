"""Host-side pose helpers the callers of the operator use to build its inputs.

Written from the conventions of the reference (quaternions are (x,y,z,w); ``T_pointcloud_camera``
maps camera -> pointcloud; docs/RawDataFormat.md:72-92).  Reference counterparts:
``SE3_to_quaternion_and_translation_torch`` UTL:485-492 (callers: RENDER:107-110, T_RAS:136-137),
``quaternion_to_rotation_matrix_torch`` UTL:596-632, ``inverse_SE3_qt_torch`` UTL:426-432.
These run on tiny (K,4)/(K,3) tensors outside the hot path; inside the operator the pose inverse
is the HIP kernel ``gs_pose_inverse``.
"""
from __future__ import annotations

from typing import Tuple

import torch


def quaternion_to_rotation_matrix_torch(q: torch.Tensor) -> torch.Tensor:
    """(…,4) xyzw -> (…,3,3).  The quaternion is used as given (assumed unit)."""
    x, y, z, w = q.unbind(-1)
    rows = [
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], dim=-1),
        torch.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], dim=-1),
        torch.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=-1),
    ]
    return torch.stack(rows, dim=-2)


def rotation_matrix_to_quaternion_torch(R: torch.Tensor) -> torch.Tensor:
    """(B,3,3) -> (B,4) xyzw, branch on the largest diagonal term for numerical stability."""
    m00, m11, m22 = R[..., 0, 0], R[..., 1, 1], R[..., 2, 2]
    trace = m00 + m11 + m22
    q = torch.zeros(R.shape[:-2] + (4,), dtype=R.dtype, device=R.device)
    c0 = trace > 0
    c1 = (~c0) & (m00 > m11) & (m00 > m22)
    c2 = (~c0) & (~c1) & (m11 > m22)
    c3 = ~(c0 | c1 | c2)
    eps = torch.finfo(R.dtype).tiny

    s0 = torch.sqrt(torch.clamp(trace + 1.0, min=eps)) * 2.0  # 4w
    s1 = torch.sqrt(torch.clamp(1.0 + m00 - m11 - m22, min=eps)) * 2.0  # 4x
    s2 = torch.sqrt(torch.clamp(1.0 + m11 - m00 - m22, min=eps)) * 2.0  # 4y
    s3 = torch.sqrt(torch.clamp(1.0 + m22 - m00 - m11, min=eps)) * 2.0  # 4z
    cand0 = torch.stack([(R[..., 2, 1] - R[..., 1, 2]) / s0, (R[..., 0, 2] - R[..., 2, 0]) / s0,
                         (R[..., 1, 0] - R[..., 0, 1]) / s0, 0.25 * s0], dim=-1)
    cand1 = torch.stack([0.25 * s1, (R[..., 0, 1] + R[..., 1, 0]) / s1, (R[..., 0, 2] + R[..., 2, 0]) / s1,
                         (R[..., 2, 1] - R[..., 1, 2]) / s1], dim=-1)
    cand2 = torch.stack([(R[..., 0, 1] + R[..., 1, 0]) / s2, 0.25 * s2, (R[..., 1, 2] + R[..., 2, 1]) / s2,
                         (R[..., 0, 2] - R[..., 2, 0]) / s2], dim=-1)
    cand3 = torch.stack([(R[..., 0, 2] + R[..., 2, 0]) / s3, (R[..., 1, 2] + R[..., 2, 1]) / s3, 0.25 * s3,
                         (R[..., 1, 0] - R[..., 0, 1]) / s3], dim=-1)
    for cond, cand in ((c0, cand0), (c1, cand1), (c2, cand2), (c3, cand3)):
        q = torch.where(cond.unsqueeze(-1), cand, q)
    return q


def SE3_to_quaternion_and_translation_torch(transform: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(B,4,4) -> ((B,4) xyzw, (B,3))."""
    return rotation_matrix_to_quaternion_torch(transform[..., :3, :3]), transform[..., :3, 3]


def inverse_SE3_qt_torch(q: torch.Tensor, t: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Inverse of the rigid transform (q,t): q_inv = conj(q); t_inv = -R(q_inv/|q_inv|) t."""
    q_inv = torch.cat([-q[..., :3], q[..., 3:4]], dim=-1)
    qn = q_inv / q_inv.norm(dim=-1, keepdim=True)
    t_inv = -(quaternion_to_rotation_matrix_torch(qn) @ t.unsqueeze(-1)).squeeze(-1)
    return q_inv, t_inv
