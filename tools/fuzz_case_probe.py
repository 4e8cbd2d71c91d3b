#!/usr/bin/env python
"""Looks into ONE draw of tests/test_fuzz_gpu.py (development tool; run through gpurun): which rows of the dense gradients
carry the distance to the fp32 oracle, how far the fp32 oracle itself is from the float64 build on those rows, and whether
the distance moves with the operator's options.  usage: python tools/fuzz_case_probe.py <case> [frames]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gs_oracle as O  # noqa: E402
from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Op  # noqa: E402
from tests.helpers import FRAGILE_MARGIN, oracle_forward, rel_l2  # noqa: E402
from tests.test_fuzz_gpu import random_scene  # noqa: E402

case = int(sys.argv[1])
scene, band, needles, opt = random_scene(case)
f = oracle_forward(scene)
spec = oracle_forward(scene, precision="f64")
rng = np.random.default_rng(5_000 + case)
keep = f["margin"] >= FRAGILE_MARGIN
g = (rng.random((scene.height, scene.width, 3)) * 2 - 1).astype(np.float32) * keep[:, :, None]
ob = O.backward(f, g, band)
ob64 = O.backward(spec, g.astype(np.float64), band)
print(f"case {case}: {scene.width}x{scene.height} n={scene.point_cloud.shape[0]} M={len(f['ids'])} band {band} needles={needles} "
      f"near={scene.near_plane} opt={opt}; fp32 oracle vs f64: grad_xyz {rel_l2(ob['grad_xyz'], ob64['grad_xyz']):.3e} "
      f"grad_feat {rel_l2(ob['grad_feat'], ob64['grad_feat']):.3e}; decisions differ on "
      f"{int((f['count'] != spec['count']).sum())} pixels, kept {int(keep.sum())} of {keep.size}")
s = scene.to("cuda")


def run(options):
    op = Op(Op.GaussianPointCloudRasterisationConfig(near_plane=s.near_plane, far_plane=s.far_plane,
                                                     depth_to_sort_key_scale=s.depth_to_sort_key_scale))
    for k, v in options.items():
        if k != "hook":
            setattr(op, k, v)
    xyz = s.point_cloud.clone().requires_grad_(True)
    feat = s.point_cloud_features.clone().requires_grad_(True)
    image, depth, count = op(Op.GaussianPointCloudRasterisationInput(
        point_cloud=xyz, point_cloud_features=feat, point_object_id=s.point_object_id, point_invalid_mask=s.point_invalid_mask,
        camera_info=CameraInfo(camera_intrinsics=s.camera_intrinsics, camera_height=s.height, camera_width=s.width, camera_id=0),
        q_pointcloud_camera=s.q_pointcloud_camera, t_pointcloud_camera=s.t_pointcloud_camera, color_max_sh_band=band))
    (image * torch.from_numpy(g).cuda()).sum().backward()
    return image.detach().cpu().numpy(), count.cpu().numpy(), xyz.grad.cpu().numpy(), feat.grad.cpu().numpy()


variants = [("as drawn", opt), ("fused_slot_reduction off", dict(opt, fused_slot_reduction=False)),
            ("bin_shift 0", dict(opt, bin_shift=0)), ("defaults", {})]
for name, o in variants:
    img, cnt, gx, gf = run(o)
    print(f"  {name:28s} grad_xyz vs fp32 {rel_l2(gx, ob['grad_xyz']):.3e} vs f64 {rel_l2(gx, ob64['grad_xyz']):.3e} | grad_feat vs fp32 "
          f"{rel_l2(gf, ob['grad_feat']):.3e} vs f64 {rel_l2(gf, ob64['grad_feat']):.3e} | counts differ on {int((cnt != f['count'])[keep].sum())} kept pixels")
img, cnt, gx, gf = run(opt)
d = np.abs(gx - ob["grad_xyz"]).max(axis=1)
rows = np.argsort(-d)[:5]
cam_z = f["xyz_cam"][:, 2] if "xyz_cam" in f else None
for r in rows:
    i = int(np.nonzero(f["ids"] == r)[0][0]) if r in f["ids"] else -1
    print(f"  row {r}: |hip - fp32| {d[r]:.3e}, |fp32 - f64| {np.abs(ob['grad_xyz'][r] - ob64['grad_xyz'][r]).max():.3e}, |hip - f64| "
          f"{np.abs(gx[r] - ob64['grad_xyz'][r]).max():.3e}, |grad| {np.abs(ob['grad_xyz'][r]).max():.3e} (largest of all rows "
          f"{np.abs(ob['grad_xyz']).max():.3e}); scales {np.exp(scene.point_cloud_features[r, 4:7].numpy())}, "
          f"depth {float(cam_z[i]) if cam_z is not None and i >= 0 else float('nan'):.3f}, radius {float(f['radii'][i]) if i >= 0 else float('nan'):.1f}")
