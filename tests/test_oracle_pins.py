"""Pins the CPU oracle against every known-answer vector the reference's own
tests hold for the hot path.  CPU only.

T_RAS = /root/reference/tests/GaussianPointCloudRasterisation_test.py
T_GP3 = /root/reference/tests/GaussianPoint3D_test.py
T_UTL = /root/reference/tests/utils_test.py
UTL   = /root/reference/taichi_3d_gaussian_splatting/utils.py
"""
import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

from oracle import gs_oracle as O

PRECISIONS = ["f32", "f64"]


def test_find_tile_start_and_end_known_answer():
    # T_RAS:18-51
    keys = np.array([0x100000000, 0x100000001, 0x200000000, 0x200000001, 0x200000002,
                     0x300000000, 0x300000001], dtype=np.int64)
    start, end = O.tile_ranges(keys, 4)
    assert start.tolist() == [0, 0, 2, 5]
    assert end.tolist() == [0, 2, 5, 7]


@pytest.mark.parametrize("precision", PRECISIONS)
def test_rotation_matrix_from_quaternion_vs_scipy(precision):
    # T_GP3:56-67, atol 1e-2 in the reference (q is not exactly unit)
    q = np.array([0.0229, 0.9774, 0.1204, 0.1725])
    R = O.rotation_matrix_from_quaternion(q, precision)
    assert np.allclose(R, Rotation.from_quat(q).as_matrix(), atol=1e-2)
    qn = q / np.linalg.norm(q)
    Rn = O.rotation_matrix_from_quaternion(qn, precision)
    assert np.allclose(Rn, Rotation.from_quat(qn).as_matrix(), atol=1e-6)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_project_to_camera_covariance_vs_numpy(precision):
    # T_GP3:12-54 (rtol 1e-2 there because scipy normalises q and Taichi does not)
    K = np.array([[32, 0, 16], [0, 32, 16], [0, 0, 1]], dtype=np.float64)
    xyz = np.array([-0.1316, -0.2471, 1.0090])
    s = np.log(np.array([0.7606, 0.9650, 0.1946]))
    q = np.array([0.0229, 0.9774, 0.1204, 0.1725])
    R = Rotation.from_quat(q).as_matrix()
    S = np.diag(np.exp(s))
    x, y, z = xyz
    J = np.array([[32 / z, 0, -32 * x / (z * z)], [0, 32 / z, -32 * y / (z * z)]])
    cov_np = J @ R @ S @ S @ R.T @ J.T
    cov = O.project_covariance(q, s, np.eye(3), K, xyz, precision)
    assert np.allclose(cov, cov_np, rtol=1e-2)
    # regression values derived in the survey (float64 evaluation with the un-normalised q)
    assert np.allclose(cov, [[565.169, 30.840], [30.840, 992.793]], rtol=2e-4)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_inverse_se3_qt_vs_numpy(precision):
    # T_UTL:138-157
    rng = np.random.default_rng(0)
    q = rng.random((100, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    t = rng.random((100, 3))
    T = np.tile(np.eye(4), (100, 1, 1))
    T[:, :3, :3] = Rotation.from_quat(q).as_matrix(); T[:, :3, 3] = t
    qi, ti = O.inverse_se3_qt(q, t, precision)
    Ti = np.tile(np.eye(4), (100, 1, 1))
    for i in range(100):
        Ti[i, :3, :3] = O.rotation_matrix_from_quaternion(qi[i], precision)
    Ti[:, :3, 3] = ti
    assert np.allclose(np.linalg.inv(T), Ti, atol=1e-5 if precision == "f32" else 1e-12)


def test_sh_basis_constants():
    # SPH:16-31 evaluated in float64 numpy, independent of the C code
    d = np.array([0.3, -0.5, 0.81])
    x, y, z = d / np.linalg.norm(d)
    expect = [0.28209479177387814, -0.48860251190291987 * y, 0.48860251190291987 * z,
              -0.48860251190291987 * x, 1.0925484305920792 * x * y, -1.0925484305920792 * y * z,
              0.94617469575755997 * z * z - 0.31539156525251999, -1.0925484305920792 * x * z,
              0.54627421529603959 * (x * x - y * y), 0.59004358992664352 * y * (-3 * x * x + y * y),
              2.8906114426405538 * x * y * z, 0.45704579946446572 * y * (1 - 5 * z * z),
              0.3731763325901154 * z * (5 * z * z - 3), 0.45704579946446572 * x * (1 - 5 * z * z),
              1.4453057213202769 * z * (x * x - y * y), 0.59004358992664352 * x * (-x * x + 3 * y * y)]
    assert np.allclose(O.sh_basis(d, "f64"), expect, atol=1e-14)
    assert np.allclose(O.sh_basis(d, "f32"), expect, atol=1e-6)


# ---- single Gaussian known answer, T_RAS:353-548 ---------------------------------------------
FEATURES = [0.0115, 0.5507, 0.6920, 0.4666, np.log(0.6306), np.log(0.0871), np.log(0.0112), 1.7667,
            2.2963, 0.1560, 0.8710, 0.3418, 0.3658, 0.1913, 0.8727, 0.3608,
            0.6874, 0.7516, 0.9281, 0.5649, 0.9469, 0.9090, 0.7356, 0.5436,
            1.7886, 0.7542, 0.9568, 0.2868, 0.3552, 0.3872, 0.0827, 0.4101,
            0.7783, 0.6266, 0.9601, 0.8252, 0.7846, 0.0183, 0.6635, 0.4688,
            -1.4012, 0.1584, 0.3252, 0.5403, 0.4992, 0.2780, 0.7412, 0.5056,
            0.8236, 0.9722, 0.5467, 0.6644, 0.2583, 0.0953, 0.3986, 0.2265]


def _torch_single_point_alpha(xyz, feat, T_cp, K, pixel_uv):
    """The reference's pure-torch comparator restated (UTL:513-558 with the
    rotation helper UTL:596-632): NON-conic Gaussian weight, J detached."""
    xyz1 = torch.cat([xyz, torch.ones_like(xyz[:1])])
    c = (T_cp @ xyz1)[:3]
    uv1 = K @ c
    uv = uv1[:2] / uv1[2]
    # like UTL:613-632, x,y,z,w are unbound BEFORE the normalisation, i.e. q is used as given
    x, y, z, w = feat[:4].unbind(-1)
    R = torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)]),
                     torch.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)]),
                     torch.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)])])
    S = torch.diag(torch.exp(feat[4:7]))
    Sigma = R @ S @ S @ R.T
    cd = c.detach()
    J = torch.tensor([[K[0, 0] / cd[2], 0, -K[0, 0] * cd[0] / cd[2] ** 2],
                      [0, K[1, 1] / cd[2], -K[1, 1] * cd[1] / cd[2] ** 2]], dtype=xyz.dtype)
    W = T_cp[:3, :3]
    cov = J @ W @ Sigma @ W.T @ J.T
    d = pixel_uv.to(xyz.dtype) + 0.5 - uv
    p = torch.exp(-0.5 * d @ torch.inverse(cov) @ d)
    return torch.sigmoid(feat[7]) * p, uv, cov


def test_single_point_alpha_and_projection_pin():
    dt = torch.float64
    T_cp = torch.eye(4, dtype=dt); T_cp[2, 3] = 2.0
    K = torch.tensor([[32., 0., 16.], [0., 32., 16.], [0., 0., 1.]], dtype=dt)
    xyz = torch.tensor([-0.4325, -0.7224, -0.4733], dtype=dt, requires_grad=True)
    feat = torch.tensor(FEATURES, dtype=dt, requires_grad=True)
    alpha, uv, cov = _torch_single_point_alpha(xyz, feat, T_cp, K, torch.tensor([3, 3]))
    # values recorded in SURVEY.md 8(c)
    assert np.allclose(uv.detach().numpy(), [6.9347, 0.8583], atol=1e-4)
    assert np.allclose(cov.detach().numpy(), [[87.5116, -52.6297], [-52.6297, 31.8323]], rtol=1e-4)
    assert abs(alpha.item() - 0.318772) < 1e-5
    # oracle: projection + covariance of the same point through the production entry points.
    # q_pointcloud_camera = identity, t_pointcloud_camera = (0,0,-2)  <=>  T_camera_pointcloud t_z = +2
    for precision, tol in (("f32", 1e-4), ("f64", 1e-9)):
        f = O.forward(xyz.detach().numpy()[None], feat.detach().numpy()[None], np.zeros(1, np.int8),
                      np.zeros(1, np.int32), K.numpy(), np.array([[0, 0, 0, 1.]]), np.array([[0, 0, -2.]]),
                      32, 32, near_plane=0.1, precision=precision)
        assert f["ids"].tolist() == [0]
        assert np.allclose(f["uv"][0], uv.detach().numpy(), atol=tol * 10)
        # un-filtered covariance recovered from the conic: inv([[A,B],[B,C]]) - 0.3 I
        A, B, C, _ = f["conic"][0].astype(np.float64)
        cov_o = np.linalg.inv(np.array([[A, B], [B, C]])) - 0.3 * np.eye(2)
        # the operator normalises q in place first (|q| = 0.99999 here), the comparator uses q as given
        assert np.allclose(cov_o, cov.detach().numpy(), rtol=2e-3 if precision == "f32" else 1e-4)
        assert abs(f["alpha"][0] - torch.sigmoid(feat[7]).item()) < tol


def test_single_point_gradients_pin():
    """dalpha/d{xyz,q,s,logit} of the oracle's Jacobian chain vs torch autograd on the
    reference's comparator (T_RAS:441-548: tolerances 1e-4 forward/xyz, 1e-2 features).
    The oracle's blend uses the conic (low-pass) weight, so the chain is assembled here from
    the oracle's per-point backward with hand-made accumulators for the NON-conic weight."""
    import ctypes
    dt = torch.float64
    T_cp = torch.eye(4, dtype=dt); T_cp[2, 3] = 2.0
    K = torch.tensor([[32., 0., 16.], [0., 32., 16.], [0., 0., 1.]], dtype=dt)
    xyz = torch.tensor([-0.4325, -0.7224, -0.4733], dtype=dt, requires_grad=True)
    feat0 = torch.tensor(FEATURES, dtype=dt)
    feat0[:4] = feat0[:4] / feat0[:4].norm()   # the operator normalises q in place first
    feat = feat0.clone().requires_grad_(True)
    alpha, uv, cov = _torch_single_point_alpha(xyz, feat, T_cp, K, torch.tensor([3, 3]))
    alpha.backward()
    # hand-made upstream accumulators: dalpha/duv and dalpha/dcov for the non-conic weight
    covd = cov.detach().numpy(); inv = np.linalg.inv(covd)
    d = np.array([3.5, 3.5]) - uv.detach().numpy()
    p = np.exp(-0.5 * d @ inv @ d); a_pt = 1 / (1 + np.exp(-FEATURES[7]))
    m = inv @ d
    g_uv = a_pt * p * m
    g_cov = a_pt * 0.5 * p * np.outer(m, m)
    for precision, tol_x, tol_f in (("f64", 1e-8, 1e-7), ("f32", 1e-4, 1e-2)):
        lib, rt = O._lib(precision), O._real(precision)
        acc = np.zeros((1, 10), rt)
        acc[0, 0:2] = g_uv; acc[0, 2] = g_cov[0, 0]; acc[0, 3] = g_cov[0, 1]; acc[0, 4] = g_cov[1, 1]
        acc[0, 8] = p * a_pt * (1 - a_pt)
        xyz_np = np.ascontiguousarray(xyz.detach().numpy()[None], rt)
        feat_np = np.ascontiguousarray(feat0.numpy()[None], rt)
        q_cp, t_cp = O.inverse_se3_qt(np.array([[0, 0, 0, 1.]]), np.array([[0, 0, -2.]]), precision)
        c = (T_cp.numpy() @ np.append(xyz_np[0].astype(np.float64), 1.0))[:3]
        xyz_cam = np.ascontiguousarray(c[None], rt)
        gx = np.zeros((1, 3), rt); gf = np.zeros((1, 56), rt)
        lib.gs_oracle_point_backward(O._p(xyz_np), O._p(feat_np), O._p(np.zeros(1, np.int32)),
                                     O._p(np.ascontiguousarray(K.numpy().reshape(9), rt)), O._p(q_cp), O._p(t_cp),
                                     O._p(np.ascontiguousarray([[0, 0, -2.]], rt)), O._p(np.zeros(1, np.int32)),
                                     ctypes.c_int(1), O._p(xyz_cam), O._p(acc), O._p(gx), O._p(gf))
        assert np.allclose(gx[0], xyz.grad.numpy(), atol=tol_x)
        assert np.allclose(gf[0, :8], feat.grad.numpy()[:8], atol=tol_f)


def test_feature_column_layout():
    # T_RAS:83-104: q[0:4], s[4:7], alpha[7], r[8:24], g[24:40], b[40:56].
    # Checked through behaviour: only column 7 changes alpha, only 8..23 change red, etc.
    rng = np.random.default_rng(3)
    xyz = np.array([[0.1, -0.2, 0.3]]); feat = rng.normal(size=(1, 56)) * 0.3
    args = (np.zeros(1, np.int8), np.zeros(1, np.int32), np.array([[100., 0, 32], [0, 100., 32], [0, 0, 1]]),
            np.array([[0, 0, 0, 1.]]), np.array([[0, 0, -3.]]), 64, 64)
    base = O.forward(xyz, feat, *args, precision="f64")
    for col, key, ch in ((7, "alpha", None), (8, "rgb", 0), (23, "rgb", 0), (24, "rgb", 1), (39, "rgb", 1),
                         (40, "rgb", 2), (55, "rgb", 2)):
        f2 = feat.copy(); f2[0, col] += 0.5
        out = O.forward(xyz, f2, *args, precision="f64")
        for k in ("alpha", "rgb", "conic", "uv"):
            changed = np.abs(out[k] - base[k]) > 1e-12
            if k == key:
                assert changed.any()
                if ch is not None:
                    assert changed[0, ch] and changed.sum() == 1
            else:
                assert not changed.any(), (col, k)
    for col in range(4, 7):
        f2 = feat.copy(); f2[0, col] += 0.5
        out = O.forward(xyz, f2, *args, precision="f64")
        assert (np.abs(out["conic"] - base["conic"]) > 1e-12).any()
        assert not (np.abs(out["rgb"] - base["rgb"]) > 1e-12).any()
