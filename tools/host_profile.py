#!/usr/bin/env python
"""Host-side cost of one operator step when the GPU has next to nothing to do (development tool; run through gpurun).
usage: python tools/host_profile.py [workload] [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Op  # noqa: E402
from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene, make_grad_image  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "cfg1_10k_256"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
s = make_config_scene(workload).to("cuda")
g = make_grad_image(s.height, s.width).to("cuda")
op = Op(Op.GaussianPointCloudRasterisationConfig(near_plane=s.near_plane, far_plane=s.far_plane,
                                                 depth_to_sort_key_scale=s.depth_to_sort_key_scale),
        backward_valid_point_hook=lambda h: None)
if os.environ.get("GS_FRAME_ENTRY_POINTS") == "0":   # stage-by-stage foreign calls (the path of rounds 1-3)
    op.frame_entry_points = False
if os.environ.get("GS_SPLIT") == "0":   # no list splitting on small grids
    op.split_small_grid_backward = False
if os.environ.get("GS_NO_PIN") != "1":
    from taichi_3d_gaussian_splatting_amd import host_affinity  # noqa: E402
    host_affinity.pin_host_threads(0)
xyz = s.point_cloud.clone().requires_grad_(True)
feat = s.point_cloud_features.clone().requires_grad_(True)
inp = Op.GaussianPointCloudRasterisationInput(
    point_cloud=xyz, point_cloud_features=feat, point_object_id=s.point_object_id,
    point_invalid_mask=s.point_invalid_mask, camera_info=CameraInfo(s.camera_intrinsics, s.height, s.width, 0),
    q_pointcloud_camera=s.q_pointcloud_camera, t_pointcloud_camera=s.t_pointcloud_camera, color_max_sh_band=3)


def step():
    xyz.grad = None
    feat.grad = None
    image, _, _ = op(inp)
    image.backward(g)


for _ in range(30):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print(f"[host_profile] {workload} frame_entry_points={op.frame_entry_points} split={op.split_small_grid_backward}: "
      f"{1e3 * (time.perf_counter() - t0) / steps:.4f} ms per step (wall, un-profiled)", flush=True)
if os.environ.get("GS_NO_CPROFILE") == "1":
    sys.exit(0)
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
