#!/usr/bin/env python
"""How many frames does the operator need to reach its steady state?  Wall-clock time of each of the first frames of a fresh
operator on one workload, every frame fenced (development tool; run through gpurun).
usage: python tools/warmup_probe.py [workload] [frames]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Op  # noqa: E402
from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene, make_grad_image  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "headline_1m_1080p"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 16
s = make_config_scene(workload).to("cuda")
g = make_grad_image(s.height, s.width).to("cuda")
op = Op(Op.GaussianPointCloudRasterisationConfig(near_plane=s.near_plane, far_plane=s.far_plane,
                                                 depth_to_sort_key_scale=s.depth_to_sort_key_scale),
        backward_valid_point_hook=lambda h: None)
op.always_store_normalised_rotation = True
xyz = s.point_cloud.clone().requires_grad_(True)
feat = s.point_cloud_features.clone().requires_grad_(True)
inp = Op.GaussianPointCloudRasterisationInput(
    point_cloud=xyz, point_cloud_features=feat, point_object_id=s.point_object_id, point_invalid_mask=s.point_invalid_mask,
    camera_info=CameraInfo(camera_intrinsics=s.camera_intrinsics, camera_height=s.height, camera_width=s.width, camera_id=0),
    q_pointcloud_camera=s.q_pointcloud_camera, t_pointcloud_camera=s.t_pointcloud_camera, color_max_sh_band=3)
if os.environ.get("GS_PREBURN_MS"):   # unrelated GPU work first: is the slow start the operator's or the GPU's (clocks)?
    a = torch.randn(8192, 8192, device="cuda")
    t0 = time.perf_counter()
    while 1e3 * (time.perf_counter() - t0) < float(os.environ["GS_PREBURN_MS"]):
        (a @ a).sum().item()
torch.cuda.synchronize()
out, gpu = [], []
for i in range(frames):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    xyz.grad = None
    feat.grad = None
    image, _, _ = op(inp)
    image.backward(g)
    b.record()
    torch.cuda.synchronize()
    out.append(round(1e3 * (time.perf_counter() - t0), 3))
    gpu.append(round(a.elapsed_time(b), 3))
print(f"[warmup_probe] {workload}: ms per fenced frame, wall {out}; between two events on the stream {gpu}; "
      f"layout bin_shift {op.list_layout(s.height).bin_shift}; speculation {dict(op.speculation_stats)}")
