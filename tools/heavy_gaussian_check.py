#!/usr/bin/env python
"""What do a few screen-filling Gaussians (floaters close to the camera: a common transient in training) cost?  The
headline scene with 0 / 1 / 16 / 256 of its Gaussians blown up to cover the whole image; forward + backward per frame.
usage: python tools/heavy_gaussian_check.py   (through gpurun)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Op  # noqa: E402
from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene, make_grad_image  # noqa: E402

s = make_config_scene(sys.argv[1] if len(sys.argv) > 1 else "headline_1m_1080p").to("cuda")
g = make_grad_image(s.height, s.width).to("cuda")
for n_heavy in tuple(int(x) for x in os.environ.get("GS_HEAVY", "0,1,16,256").split(",")):
    feat0 = s.point_cloud_features.clone()
    xyz0 = s.point_cloud.clone()
    if n_heavy:
        idx = torch.arange(n_heavy, device="cuda") * 1000 + 7
        xyz0[idx] = torch.tensor([0.0, 0.0, 0.0], device="cuda") + 0.05 * torch.randn(n_heavy, 3, device="cuda")
        feat0[idx, 4:7] = 1.5          # log-scale: sigma = e^1.5 -> far larger than the view
        feat0[idx, 7] = -3.0           # faint (opacity 0.05): it is blended everywhere without saturating anything
    op = Op(Op.GaussianPointCloudRasterisationConfig())
    xyz = xyz0.requires_grad_(True)
    feat = feat0.requires_grad_(True)
    inp = Op.GaussianPointCloudRasterisationInput(
        point_cloud=xyz, point_cloud_features=feat, point_object_id=s.point_object_id,
        point_invalid_mask=s.point_invalid_mask, camera_info=CameraInfo(s.camera_intrinsics, s.height, s.width, 0),
        q_pointcloud_camera=s.q_pointcloud_camera, t_pointcloud_camera=s.t_pointcloud_camera, color_max_sh_band=3)

    def step():
        xyz.grad = None
        feat.grad = None
        image, _, _ = op(inp)
        image.backward(g)
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    print(f"{n_heavy:4d} screen-filling Gaussians: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per frame "
          f"(bin_shift {op.list_layout(s.height).bin_shift}, {op.speculation_stats})")
