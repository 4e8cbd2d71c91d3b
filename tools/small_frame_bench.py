#!/usr/bin/env python
"""Operator time (forward + backward, HIP events, median of `reps`) on small square frames -- where the launch of a blend kernel
lasts as long as its longest tile's chain and the list splitting of both passes (include/gsplat_hip.h "List splitting")
decides the frame time.  Development tool: GS_FWD_SPLIT / GS_BWD_SPLIT force the number of workgroups per tile (1 = un-split).
usage: python tools/small_frame_bench.py [size:n ...]      default: 256:10000 384:25000 512:60000 640:90000"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Op  # noqa: E402
from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image, make_scene  # noqa: E402

cases = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or [(256, 10_000), (384, 25_000), (512, 60_000), (640, 90_000)]
reps = int(os.environ.get("GS_REPS", "200"))
for size, n in cases:
    s = make_scene(n=n, height=size, width=size, s_min=0.01, s_max=0.08, seed=size).to("cuda")
    g = make_grad_image(size, size).cuda()
    op = Op(Op.GaussianPointCloudRasterisationConfig(near_plane=s.near_plane, far_plane=s.far_plane,
                                                     depth_to_sort_key_scale=s.depth_to_sort_key_scale),
            backward_valid_point_hook=lambda h: None)
    xyz = s.point_cloud.clone().requires_grad_(True)
    feat = s.point_cloud_features.clone().requires_grad_(True)
    inp = Op.GaussianPointCloudRasterisationInput(
        point_cloud=xyz, point_cloud_features=feat, point_object_id=s.point_object_id, point_invalid_mask=s.point_invalid_mask,
        camera_info=CameraInfo(s.camera_intrinsics, size, size, 0), q_pointcloud_camera=s.q_pointcloud_camera,
        t_pointcloud_camera=s.t_pointcloud_camera, color_max_sh_band=3)

    def step():
        image, _, _ = op(inp)
        image.backward(g)
        xyz.grad = None
        feat.grad = None

    for _ in range(20):
        step()
    times = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        step()
        b.record()
        b.synchronize()
        times.append(a.elapsed_time(b))
    times.sort()
    # ... and back to back (nothing between the steps: what a training loop sees when the host keeps ahead)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        step()
    b.record()
    b.synchronize()
    loop = a.elapsed_time(b) / reps
    print(f"[small_frame] {size}x{size} n={n} tiles={(size // 16) ** 2} fwd_split={os.environ.get('GS_FWD_SPLIT', 'auto')} "
          f"bwd_split={os.environ.get('GS_BWD_SPLIT', 'auto')}: back to back {loop:.4f} ms per step; one at a time: median {times[len(times) // 2]:.4f} ms, min {times[0]:.4f}", flush=True)
