#!/usr/bin/env python
"""Generates tests/golden/reference_utils_vectors.npz by EXECUTING the reference's own pure-PyTorch functions
(taichi_3d_gaussian_splatting/utils.py under /root/reference) on seeded inputs.

Run in the build container only (the GPU box has no /root/reference):  python tests/golden/make_reference_vectors.py

The reference's utils.py imports Taichi at module level and decorates its Taichi functions at import time; Taichi is
not installed here, so a stub module named ``taichi`` (every attribute / call returns the stub, decorators return the
function unchanged) is put in sys.modules first.  Only functions that never touch Taichi are then called -- the
comparator the reference's rasteriser tests use (torch_single_point_alpha_forward, UTL:513-558, with its rotation
helper UTL:596-632), the torch SH basis (UTL:635-660) and the SE(3)/quaternion helpers (UTL:386-493).  Nothing is
copied from the reference: the script imports it where it lies and stores inputs and outputs.
"""
import contextlib
import importlib
import io
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return self

    def __call__(self, *args, **kwargs):
        if len(args) == 1 and not kwargs and callable(args[0]) and not isinstance(args[0], _Stub):
            return args[0]          # used as a decorator
        return self


def load_reference_utils():
    stub = _Stub("taichi")
    sys.modules["taichi"] = stub
    sys.modules["taichi.math"] = stub
    pkg = types.ModuleType("taichi_3d_gaussian_splatting")     # package shell: do not run the reference's __init__
    pkg.__path__ = [os.path.join(REF, "taichi_3d_gaussian_splatting")]
    sys.modules["taichi_3d_gaussian_splatting"] = pkg
    return importlib.import_module("taichi_3d_gaussian_splatting.utils")


def random_pose(g, dtype):
    q = torch.randn(4, generator=g, dtype=dtype)
    q = q / q.norm()
    x, y, z, w = q.tolist()
    R = torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=dtype)
    T = torch.eye(4, dtype=dtype)
    T[:3, :3] = R
    T[:3, 3] = torch.randn(3, generator=g, dtype=dtype)
    return T


def main():
    U = load_reference_utils()
    dt = torch.float64
    g = torch.Generator().manual_seed(2024)
    out = {}

    # --- single-Gaussian alpha + autograd gradients (the comparator of T_RAS:441-548) ------------------------
    n = 48
    K = torch.tensor([[32.0, 0, 16], [0, 32.0, 16], [0, 0, 1]], dtype=dt)
    cases = dict(xyz=[], q=[], s=[], logit=[], T=[], uv=[], alpha=[], g_xyz=[], g_q=[], g_s=[], g_logit=[])
    # case 0 is the reference's own test vector (T_RAS:346-372): T = I with t_z = 2, pixel (3,3); its quaternion
    # (norm 0.99999) is normalised first, as the operator does in place before using it (RAS:196-205)
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from tests.test_oracle_pins import FEATURES
    ref_xyz = [-0.4325, -0.7224, -0.4733]
    ref_feat = list(FEATURES[:8])
    qn = float(np.linalg.norm(ref_feat[0:4]))
    ref_feat[0:4] = [c / qn for c in ref_feat[0:4]]
    while len(cases["alpha"]) < n:
        i = len(cases["alpha"])
        if i == 0:
            xyz = torch.tensor(ref_xyz, dtype=dt)
            q, s, logit = (torch.tensor(ref_feat[0:4], dtype=dt), torch.tensor(ref_feat[4:7], dtype=dt),
                           torch.tensor(ref_feat[7:8], dtype=dt))
            T = torch.eye(4, dtype=dt); T[2, 3] = 2.0
            pix = torch.tensor([3, 3])
        else:
            xyz = torch.randn(3, generator=g, dtype=dt) * 0.5
            q = torch.randn(4, generator=g, dtype=dt)
            q = q / q.norm()
            s = torch.rand(3, generator=g, dtype=dt) * 1.5 - 2.0
            logit = torch.randn(1, generator=g, dtype=dt)
            T = random_pose(g, dt)
            T[:3, 3] = torch.tensor([0.0, 0.0, 3.0], dtype=dt) + 0.3 * torch.randn(3, generator=g, dtype=dt)
            pix = torch.randint(0, 32, (2,), generator=g)
        xyz.requires_grad_(True); q.requires_grad_(True); s.requires_grad_(True); logit.requires_grad_(True)
        with contextlib.redirect_stdout(io.StringIO()):      # the comparator prints its intermediates
            alpha = U.torch_single_point_alpha_forward(point_xyz=xyz, point_q=q, point_s=s, T_camera_pointcloud=T,
                                                       camera_intrinsics=K, point_alpha=logit, pixel_uv=pix)
            alpha.sum().backward()
        if not torch.isfinite(alpha).all() or float(alpha) < 1e-6:
            continue                                          # keep cases with a usable signal
        for k, v in (("xyz", xyz), ("q", q), ("s", s), ("logit", logit), ("T", T), ("uv", pix.to(dt)),
                     ("alpha", alpha.reshape(1)), ("g_xyz", xyz.grad), ("g_q", q.grad), ("g_s", s.grad),
                     ("g_logit", logit.grad)):
            cases[k].append(v.detach().numpy().copy())
    for k, v in cases.items():
        out["sp_" + k] = np.stack(v)
    out["sp_K"] = K.numpy()

    # --- spherical-harmonics basis (UTL:635-660) ----------------------------------------------------------------
    dirs = torch.randn(64, 3, generator=g, dtype=dt)
    out["sh_dirs"] = dirs.numpy().copy()
    out["sh_basis"] = np.stack([U.get_spherical_harmonic_from_xyz_torch(d.clone()).numpy() for d in dirs])

    # --- SE(3) / quaternion helpers (UTL:386-493, 596-632) -------------------------------------------------------
    qs = torch.randn(100, 4, generator=g, dtype=dt)
    qs = qs / qs.norm(dim=1, keepdim=True)
    ts = torch.randn(100, 3, generator=g, dtype=dt)
    qi, ti = U.inverse_SE3_qt_torch(qs, ts)
    out.update(pose_q=qs.numpy(), pose_t=ts.numpy(), pose_q_inv=qi.numpy(), pose_t_inv=ti.numpy())
    out["rot_from_q"] = U.quaternion_to_rotation_matrix_torch(qs).numpy()
    Ts = torch.stack([random_pose(g, dt) for _ in range(50)])
    q_from_T, t_from_T = U.SE3_to_quaternion_and_translation_torch(Ts)
    out.update(se3_T=Ts.numpy(), se3_q=q_from_T.numpy(), se3_t=t_from_T.numpy())
    out["inverse_SE3"] = np.stack([U.inverse_SE3(T).numpy() for T in Ts])
    v = torch.randn(100, 3, generator=g, dtype=dt)
    out["rotated_v"] = U.quaternion_rotate_torch(qs, v).numpy()
    out["rotate_v_in"] = v.numpy()

    path = os.path.join(HERE, "reference_utils_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
