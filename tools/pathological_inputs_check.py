#!/usr/bin/env python
"""Frame time of the headline scene under perturbations that real training runs produce: needle-shaped Gaussians
(long tile boxes that the exact cull empties), a dense faint cluster in a few tiles (one very long list), and most rows
invalid (the fixed-capacity point cloud of the trainer).  usage: python tools/pathological_inputs_check.py  (gpurun)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Op  # noqa: E402
from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene, make_grad_image  # noqa: E402

s = make_config_scene("headline_1m_1080p").to("cuda")
g = make_grad_image(s.height, s.width).to("cuda")
gen = torch.Generator(device="cuda").manual_seed(3)


def variant(name):
    xyz, feat, invalid = s.point_cloud.clone(), s.point_cloud_features.clone(), s.point_invalid_mask.clone()
    if name == "needles_10k":          # 10,000 Gaussians 300x longer than wide, random orientation (already random q)
        idx = torch.randperm(xyz.shape[0], device="cuda", generator=gen)[:10_000]
        feat[idx, 4] = 0.0             # sigma_x = 1 (the scene is ~2 units wide)
        feat[idx, 5:7] = -5.5
    elif name == "cluster_200k_faint":  # 200,000 faint Gaussians inside a 0.02-wide ball: ~2 x 2 tiles
        idx = torch.randperm(xyz.shape[0], device="cuda", generator=gen)[:200_000]
        xyz[idx] = 0.01 * torch.randn(len(idx), 3, device="cuda", generator=gen)
        feat[idx, 7] = -5.0           # opacity 0.0067: just above the 1/255 threshold, so that every one is blended
    elif name == "cluster_200k_tiny":   # 200,000 pixel-sized Gaussians in a 0.02-wide ball: a few tiles with 1e5-entry
        idx = torch.randperm(xyz.shape[0], device="cuda", generator=gen)[:200_000]   # lists that never saturate
        xyz[idx] = 0.01 * torch.randn(len(idx), 3, device="cuda", generator=gen)
        feat[idx, 4:7] = -7.5
    elif name == "invalid_87pct":      # trainer capacity: 7 of 8 rows free
        invalid[torch.rand(xyz.shape[0], device="cuda", generator=gen) < 0.875] = 1
    return xyz, feat, invalid


# (the first configuration of a process is run twice: the first ~30 frames after start-up are slow whatever they are)
for name in os.environ.get("GS_CASES", "baseline,baseline,needles_10k,cluster_200k_faint,invalid_87pct").split(","):
    xyz, feat, invalid = variant(name)
    op = Op(Op.GaussianPointCloudRasterisationConfig())
    xyz.requires_grad_(True); feat.requires_grad_(True)
    inp = Op.GaussianPointCloudRasterisationInput(
        point_cloud=xyz, point_cloud_features=feat, point_object_id=s.point_object_id, point_invalid_mask=invalid,
        camera_info=CameraInfo(s.camera_intrinsics, s.height, s.width, 0), q_pointcloud_camera=s.q_pointcloud_camera,
        t_pointcloud_camera=s.t_pointcloud_camera, color_max_sh_band=3)

    def step():
        xyz.grad = None; feat.grad = None
        global count
        image, _, count = op(inp)
        image.backward(g)
        return image
    for _ in range(6):
        image = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    ok = bool(torch.isfinite(image).all() and torch.isfinite(feat.grad).all())
    print(f"{name:22s} {(time.perf_counter() - t0) / 20 * 1e3:7.3f} ms per frame  (bin_shift "
          f"{op.list_layout(s.height, s.width).bin_shift}, finite {ok}, {op.speculation_stats}, "
          f"key capacity {max(v[0] for v in op._size_guesses.values())}, max blended per pixel {int(count.max())})")
