"""ctypes/numpy front-end of the CPU oracle (oracle/gs_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of gs_oracle.c.  Only tests/,
bench.py's ``cpu_baseline`` leg and ``__graft_entry__.smoke()`` may import this
module, and only as the checker.  The product package never does.

The glue between the kernels restates the torch ops of the reference operator
(RAS = taichi_3d_gaussian_splatting/GaussianPointCloudRasterisation.py):
mask -> index compaction RAS:861-870, exclusive scan RAS:913-922, stable sort
RAS:947-950, zero-initialised tile ranges RAS:954-964, gradient
post-processing RAS:1102-1125,1167-1182 and the hook gathers RAS:1128-1140.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS: Dict[str, ctypes.CDLL] = {}

# frozen class attributes of the reference config, RAS:782-786
GRAD_COLOR_FACTOR = 5.0
GRAD_HIGH_ORDER_COLOR_FACTOR = 1.0
GRAD_S_FACTOR = 0.5
GRAD_Q_FACTOR = 1.0
GRAD_ALPHA_FACTOR = 20.0


def build(force: bool = False) -> None:
    """Compile both oracle builds with the committed Makefile."""
    targets = [os.path.join(_HERE, f"libgs_oracle_{p}.so") for p in ("f32", "f64")]
    src = os.path.join(_HERE, "gs_oracle.c")
    stale = force or any(
        (not os.path.exists(t)) or os.path.getmtime(t) < os.path.getmtime(src) for t in targets)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])


def _lib(precision: str) -> ctypes.CDLL:
    if precision not in ("f32", "f64"):
        raise ValueError(precision)
    if precision not in _LIBS:
        path = os.path.join(_HERE, f"libgs_oracle_{precision}.so")
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        assert lib.gs_oracle_sizeof_real() == (4 if precision == "f32" else 8)
        _LIBS[precision] = lib
    return _LIBS[precision]


def num_threads() -> int:
    return int(_lib("f32").gs_oracle_num_threads())


def _p(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle wants contiguous arrays"
    return a.ctypes.data_as(ctypes.c_void_p)


def _real(precision):
    return np.float32 if precision == "f32" else np.float64


def _creal(precision):
    return ctypes.c_float if precision == "f32" else ctypes.c_double


def inverse_se3_qt(q, t, precision="f32"):
    """UTL:426-432 inverse_SE3_qt_torch."""
    rt = _real(precision)
    q = np.ascontiguousarray(q, dtype=rt).reshape(-1, 4)
    t = np.ascontiguousarray(t, dtype=rt).reshape(-1, 3)
    qi, ti = np.empty_like(q), np.empty_like(t)
    _lib(precision).gs_oracle_inverse_se3_qt(_p(q), _p(t), _p(qi), _p(ti), ctypes.c_int(q.shape[0]))
    return qi, ti


def rotation_matrix_from_quaternion(q, precision="f32"):
    rt = _real(precision)
    q = np.ascontiguousarray(q, dtype=rt)
    R = np.empty(9, dtype=rt)
    _lib(precision).gs_oracle_rotation_matrix_from_quaternion(_p(q), _p(R))
    return R.reshape(3, 3)


def project_covariance(q, s, W, K, c, precision="f32"):
    rt = _real(precision)
    args = [np.ascontiguousarray(a, dtype=rt) for a in (q, s, W, K, c)]
    cov = np.empty(4, dtype=rt)
    _lib(precision).gs_oracle_project_covariance(*[_p(a) for a in args], _p(cov))
    return cov.reshape(2, 2)


def sh_basis(d, precision="f32"):
    rt = _real(precision)
    d = np.ascontiguousarray(d, dtype=rt)
    Y = np.empty(16, dtype=rt)
    _lib(precision).gs_oracle_sh_basis(_p(d), _p(Y))
    return Y


def tile_ranges(keys_sorted: np.ndarray, num_tiles: int):
    """RAS:175-193 on pre-zeroed arrays (RAS:954-957)."""
    keys_sorted = np.ascontiguousarray(keys_sorted, dtype=np.int64)
    start = np.zeros(num_tiles, dtype=np.int32)
    end = np.zeros(num_tiles, dtype=np.int32)
    _lib("f32").gs_oracle_tile_ranges(_p(keys_sorted), ctypes.c_int64(keys_sorted.shape[0]), _p(start), _p(end))
    return start, end


def forward(xyz, feat, invalid_mask, object_id, K, q_pc_cam, t_pc_cam, height, width,
            near_plane=0.8, far_plane=1000.0, depth_to_sort_key_scale=100.0,
            rgb_only=False, precision="f32", want_margin=False) -> dict:
    """Whole forward of the reference operator, RAS:831-1023.

    Returns every intermediate the reference computes (dict of numpy arrays);
    ``feat`` in the result is the feature matrix AFTER the in-place quaternion
    normalisation of visible rows (RAS:196-205,264).  Inputs are not modified.
    """
    assert width % 16 == 0 and height % 16 == 0  # RAS:1193-1194
    lib, rt, cr = _lib(precision), _real(precision), _creal(precision)
    xyz = np.ascontiguousarray(xyz, dtype=rt)
    feat = np.array(feat, dtype=rt, order="C", copy=True)
    invalid_mask = np.ascontiguousarray(invalid_mask, dtype=np.int8)
    object_id = np.ascontiguousarray(object_id, dtype=np.int32)
    K = np.ascontiguousarray(K, dtype=rt).reshape(9)
    n = xyz.shape[0]
    q_cp, t_cp = inverse_se3_qt(q_pc_cam, t_pc_cam, precision)
    t_pc = np.ascontiguousarray(t_pc_cam, dtype=rt).reshape(-1, 3)

    mask = np.zeros(n, dtype=np.int8)
    lib.gs_oracle_filter(_p(xyz), _p(invalid_mask), _p(object_id), _p(K), _p(q_cp), _p(t_cp),
                         ctypes.c_int(n), cr(near_plane), cr(far_plane), ctypes.c_int(width),
                         ctypes.c_int(height), _p(mask))
    ids = np.ascontiguousarray(np.nonzero(mask)[0].astype(np.int32))  # ascending ids, RAS:864
    m = ids.shape[0]
    uv = np.empty((m, 2), rt); xyz_cam = np.empty((m, 3), rt); conic = np.empty((m, 4), rt)
    alpha = np.empty(m, rt); rgb = np.zeros((m, 3), rt); radii = np.empty(m, rt)
    lib.gs_oracle_preprocess(_p(xyz), _p(feat), _p(object_id), _p(K), _p(q_cp), _p(t_cp), _p(ids),
                             ctypes.c_int(m), _p(uv), _p(xyz_cam), _p(conic), _p(alpha), _p(rgb), _p(radii))
    ntiles = np.empty(m, np.int32)
    lib.gs_oracle_num_overlap_tiles(_p(uv), _p(radii), ctypes.c_int(m), ctypes.c_int(width),
                                    ctypes.c_int(height), _p(ntiles))
    incl = np.cumsum(ntiles, dtype=np.int64)
    total = int(incl[-1]) if m > 0 else 0
    offsets = np.ascontiguousarray(np.concatenate([np.zeros(1, np.int64), incl[:-1]])) if m > 0 \
        else np.zeros(0, np.int64)
    keys = np.empty(total, np.int64); payload = np.empty(total, np.int32)
    if total > 0:
        lib.gs_oracle_make_keys(_p(uv), _p(xyz_cam), _p(radii), _p(offsets), ctypes.c_int(m),
                                ctypes.c_int(width), ctypes.c_int(height), cr(depth_to_sort_key_scale),
                                _p(keys), _p(payload))
        lib.gs_oracle_sort_pairs(_p(keys), _p(payload), ctypes.c_int64(total))
    num_tiles = (width // 16) * (height // 16)
    tile_start = np.zeros(num_tiles, np.int32); tile_end = np.zeros(num_tiles, np.int32)
    if total > 0:
        lib.gs_oracle_tile_ranges(_p(keys), ctypes.c_int64(total), _p(tile_start), _p(tile_end))
    # the reference leaves these uninitialised when K == 0 (RAS:967-980); we define zeros
    image = np.zeros((height, width, 3), rt); depth = np.zeros((height, width), rt)
    acc_alpha = np.zeros((height, width), rt); last_eff = np.zeros((height, width), np.int32)
    count = np.zeros((height, width), np.int32)
    margin = np.full((height, width), 1e30, rt) if want_margin else None
    if total > 0:
        lib.gs_oracle_blend_forward(ctypes.c_int(height), ctypes.c_int(width), _p(tile_start), _p(tile_end),
                                    _p(payload), _p(uv), _p(xyz_cam), _p(conic), _p(alpha), _p(rgb),
                                    _p(image), _p(depth), _p(acc_alpha), _p(last_eff), _p(count),
                                    ctypes.c_int(1 if rgb_only else 0), _p(margin))
    return dict(precision=precision, height=height, width=width, xyz=xyz, feat=feat, object_id=object_id,
                K=K, q_cp=q_cp, t_cp=t_cp, t_pc=t_pc, mask=mask, ids=ids, uv=uv, xyz_cam=xyz_cam,
                conic=conic, alpha=alpha, rgb=rgb, radii=radii, num_overlap_tiles=ntiles,
                offsets=offsets, keys=keys, payload=payload, tile_start=tile_start, tile_end=tile_end,
                image=image, depth=depth, acc_alpha=acc_alpha, last_eff=last_eff, count=count,
                margin=margin)


def clear_grad_by_color_max_sh_band(grad_feat: np.ndarray, band: int) -> None:
    """RAS:1167-1182."""
    keep = {0: 1, 1: 4, 2: 9}.get(band, 16) if band < 3 else 16
    for base in (8, 24, 40):
        grad_feat[:, base + keep: base + 16] = 0.0


def backward(fwd: dict, grad_image, color_max_sh_band: int = 2) -> dict:
    """Whole backward of the reference operator, RAS:1025-1163 (hook payload included)."""
    precision = fwd["precision"]
    lib, rt = _lib(precision), _real(precision)
    h, w = fwd["height"], fwd["width"]
    m = fwd["ids"].shape[0]
    n = fwd["xyz"].shape[0]
    grad_image = np.ascontiguousarray(grad_image, dtype=rt)
    acc = np.zeros((m, 10), rt); npix = np.zeros(m, np.int32)
    mag_image = np.zeros((h, w, 2), rt)
    if fwd["keys"].shape[0] > 0:
        lib.gs_oracle_blend_backward(ctypes.c_int(h), ctypes.c_int(w), _p(fwd["tile_start"]), _p(fwd["tile_end"]),
                                     _p(fwd["payload"]), _p(fwd["uv"]), _p(fwd["conic"]), _p(fwd["alpha"]),
                                     _p(fwd["rgb"]), _p(grad_image), _p(fwd["acc_alpha"]), _p(fwd["last_eff"]),
                                     ctypes.c_int(m), _p(acc), _p(npix), _p(mag_image))
    grad_xyz = np.zeros((n, 3), rt); grad_feat = np.zeros((n, 56), rt)
    lib.gs_oracle_point_backward(_p(fwd["xyz"]), _p(fwd["feat"]), _p(fwd["object_id"]), _p(fwd["K"]),
                                 _p(fwd["q_cp"]), _p(fwd["t_cp"]), _p(fwd["t_pc"]), _p(fwd["ids"]),
                                 ctypes.c_int(m), _p(fwd["xyz_cam"]), _p(acc), _p(grad_xyz), _p(grad_feat))
    # RAS:1102-1125
    clear_grad_by_color_max_sh_band(grad_feat, color_max_sh_band)
    grad_feat[:, :4] *= GRAD_Q_FACTOR
    grad_feat[:, 4:7] *= GRAD_S_FACTOR
    grad_feat[:, 7] *= GRAD_ALPHA_FACTOR
    for base in (8, 24, 40):
        grad_feat[:, base] *= GRAD_COLOR_FACTOR
        grad_feat[:, base + 1: base + 16] *= GRAD_HIGH_ORDER_COLOR_FACTOR
    ids = fwd["ids"]
    hook = dict(point_id_in_camera_list=ids, grad_point_in_camera=grad_xyz[ids],
                grad_pointfeatures_in_camera=grad_feat[ids], grad_viewspace=acc[:, 0:2].copy(),
                magnitude_grad_viewspace=acc[:, 9].copy(), magnitude_grad_viewspace_on_image=mag_image,
                num_overlap_tiles=fwd["num_overlap_tiles"], num_affected_pixels=npix,
                point_depth=fwd["xyz_cam"][:, 2].copy(), point_uv_in_camera=fwd["uv"])
    return dict(grad_xyz=grad_xyz, grad_feat=grad_feat, acc=acc, hook=hook)


def ellipsoid_offsets(feat, precision="f32"):
    """GP3:375-388 / ADC:10-25."""
    rt = _real(precision)
    feat = np.ascontiguousarray(feat, dtype=rt)
    out = np.empty((feat.shape[0], 3), rt)
    _lib(precision).gs_oracle_ellipsoid_offsets(_p(feat), ctypes.c_int(feat.shape[0]), _p(out))
    return out


def sample_from_points(xyz, feat, uniforms, precision="f32"):
    """GP3:390-406 / ADC:27-42 with caller-supplied uniforms [n,4]."""
    rt = _real(precision)
    xyz, feat, uniforms = (np.ascontiguousarray(a, dtype=rt) for a in (xyz, feat, uniforms))
    out = np.empty((xyz.shape[0], 3), rt)
    _lib(precision).gs_oracle_sample_from_points(_p(xyz), _p(feat), _p(uniforms), ctypes.c_int(xyz.shape[0]), _p(out))
    return out


def l1_ssim(pred, gt, hwc=True, clamp=True, lambda_value=0.2, g_total=1.0, g_l1=0.0, g_dssim=0.0,
            want_grad=True, precision="f64"):
    """Trainer loss LOS:20-35 (+ clamp TRN:168) and its hand-derived gradient; returns ((L, L1, 1-SSIM), grad)."""
    rt = _real(precision)
    pred = np.ascontiguousarray(pred, dtype=rt)
    gt = np.ascontiguousarray(gt, dtype=rt)
    H, W = gt.shape[1], gt.shape[2]
    out = np.zeros(3, np.float64)
    grad = np.empty_like(pred) if want_grad else None
    _lib(precision).gs_oracle_l1_ssim(_p(pred), ctypes.c_int(int(hwc)), ctypes.c_int(int(clamp)), _p(gt),
                                      ctypes.c_int(H), ctypes.c_int(W), ctypes.c_double(lambda_value),
                                      ctypes.c_double(g_total), ctypes.c_double(g_l1), ctypes.c_double(g_dssim),
                                      _p(out), _p(grad))
    return out, grad
