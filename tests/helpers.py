"""Shared helpers of the parity tests (oracle <-> HIP path)."""
from __future__ import annotations

import numpy as np
import torch

from oracle import gs_oracle as O
from taichi_3d_gaussian_splatting_amd.synthetic import SyntheticScene, make_scene

# pixels whose blend decisions sit this close to a discontinuity (alpha = 1/255 skip, RAS:451;
# T' = 1e-4 stop, RAS:458) may legitimately flip between two correct fp32 implementations
FRAGILE_MARGIN = 5e-8  # ~50x the observed fp32 alpha disagreement between two implementations
PIXEL_TOL = 1e-4  # BASELINE.json north_star: pixel L-inf <= 1e-4


def scene_numpy(s: SyntheticScene):
    return (s.point_cloud.numpy(), s.point_cloud_features.numpy(), s.point_invalid_mask.numpy(),
            s.point_object_id.numpy(), s.camera_intrinsics.numpy(), s.q_pointcloud_camera.numpy(),
            s.t_pointcloud_camera.numpy(), s.height, s.width)


def oracle_forward(s: SyntheticScene, precision="f32", want_margin=True):
    return O.forward(*scene_numpy(s), near_plane=s.near_plane, far_plane=s.far_plane,
                     depth_to_sort_key_scale=s.depth_to_sort_key_scale, precision=precision,
                     want_margin=want_margin)


def pack_attrs(f: dict, exact_cull: bool = False) -> np.ndarray:
    """Oracle SoA intermediates -> the packed [M,16] record of include/gsplat_hip.h."""
    m = f["ids"].shape[0]
    a = np.empty((m, 16), np.float32)
    amp = f["alpha"] * f["conic"][:, 3]                                   # opacity * rescale
    a[:, 0:2] = f["uv"]; a[:, 2] = f["xyz_cam"][:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        a[:, 3] = (np.float32(2.0) * np.log(np.float32(255.0) * amp) + np.float32(1e-2)) if exact_cull else np.inf
    a[:, 4:7] = f["conic"][:, 0:3]; a[:, 7] = f["radii"]
    a[:, 8:11] = f["rgb"]; a[:, 11] = f["alpha"]
    u24, stop_t = np.float32(2.0 ** -24), np.float32(0.0001)
    a[:, 12] = amp
    # gs_stop_weight (csrc/gs_common.h): the Gaussian's share of a pixel's bracket around T' = 1e-4, per unit of alpha
    H = np.float32(1.0) / (np.float32(1.0) - np.minimum(amp, np.float32(0.99))) * np.float32(1.01)
    a[:, 13] = stop_t * np.float32(1.1) * np.float32(4.0 / 3.0) * u24 * (np.float32(12.0) + np.float32(9.0) * H)
    # gs_hit_exponent_lo (csrc/gs_common.h): the exponent below which the reference skips the pair for certain
    with np.errstate(divide="ignore", invalid="ignore"):
        L = -np.log(np.float32(255.0) * amp).astype(np.float32)
        a[:, 14] = L - np.float32(4.0 / 3.0) * u24 * (np.float32(6.0) + np.float32(4.0) * np.abs(L))
    a[:, 15] = f["conic"][:, 3]                                           # rescale
    return a


def pack_acc(acc10: np.ndarray, npix: np.ndarray) -> np.ndarray:
    a = np.zeros((acc10.shape[0], 12), np.float32)
    a[:, :10] = acc10
    a[:, 10] = npix.astype(np.int32).view(np.float32)
    return a


def dev(x, device="cuda"):
    return torch.as_tensor(np.ascontiguousarray(x)).to(device)


def small_scene(n=10_000, size=256, seed=0, sh_degree=3, **kw) -> SyntheticScene:
    return make_scene(n=n, height=size, width=size, s_min=0.01, s_max=0.08, sh_degree=sh_degree, seed=seed, **kw)


def close_fraction(a: np.ndarray, b: np.ndarray, rtol: float, atol: float) -> float:
    """Fraction of elements with |a-b| <= atol + rtol*|b|."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.mean(np.abs(a - b) <= atol + rtol * np.abs(b)))


def rel_l2(a: np.ndarray, b: np.ndarray) -> float:
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def report(name: str, **kv) -> None:
    print(f"[parity] {name}: " + ", ".join(f"{k}={v}" for k, v in kv.items()))


class OracleRasterisation(torch.nn.Module):
    """The operator surface (same Input dataclass, outputs, gradients and hook payload) computed by the CPU oracle.
    TEST ONLY: lets a test run the trainer with the oracle as its rasteriser back end and compare the training
    outcome with the HIP back end."""

    def __init__(self, config, backward_valid_point_hook=None):
        super().__init__()
        from oracle import gs_oracle as O
        from taichi_3d_gaussian_splatting_amd import GaussianPointCloudRasterisation as RAS
        self.config, self.hook = config, backward_valid_point_hook
        outer = self

        class _Fn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, xyz, feat, invalid, obj, K, q, t, height, width, band):
                fwd = O.forward(xyz.detach().cpu().numpy(), feat.detach().cpu().numpy(), invalid.cpu().numpy(),
                                obj.cpu().numpy(), K.cpu().numpy(), q.cpu().numpy(), t.cpu().numpy(), height, width,
                                near_plane=config.near_plane, far_plane=config.far_plane,
                                depth_to_sort_key_scale=config.depth_to_sort_key_scale)
                with torch.no_grad():   # in-place quaternion normalisation of visible rows (RAS:196-205)
                    feat.copy_(torch.from_numpy(fwd["feat"]).to(feat.device))
                ctx.fwd, ctx.band, ctx.device = fwd, band, xyz.device
                dev = xyz.device
                return (torch.from_numpy(fwd["image"]).to(dev), torch.from_numpy(fwd["depth"]).to(dev),
                        torch.from_numpy(fwd["count"]).to(dev))

            @staticmethod
            def backward(ctx, grad_image, grad_depth, grad_count):
                bwd = O.backward(ctx.fwd, grad_image.contiguous().cpu().numpy(), ctx.band)
                dev = ctx.device
                if outer.hook is not None:
                    h = bwd["hook"]
                    outer.hook(RAS.BackwardValidPointHookInput(**{
                        k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in h.items()}))
                return (torch.from_numpy(bwd["grad_xyz"]).to(dev), torch.from_numpy(bwd["grad_feat"]).to(dev),
                        None, None, None, None, None, None, None, None)

        self._fn = _Fn

    def forward(self, inp):
        cam = inp.camera_info
        return self._fn.apply(inp.point_cloud, inp.point_cloud_features, inp.point_invalid_mask, inp.point_object_id,
                              cam.camera_intrinsics, inp.q_pointcloud_camera, inp.t_pointcloud_camera,
                              cam.camera_height, cam.camera_width, inp.color_max_sh_band)
