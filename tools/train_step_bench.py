#!/usr/bin/env python
"""Wall-clock split of one TRAINING iteration (rasteriser + loss + optimisers) on a synthetic workload.
Development tool; run through gpurun.  usage: python tools/train_step_bench.py [workload] [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as RAS  # noqa: E402
from taichi_3d_gaussian_splatting_amd.LossFunction import LossFunction  # noqa: E402
from taichi_3d_gaussian_splatting_amd.optim import Adam  # noqa: E402
from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "headline_1m_1080p"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
s = make_config_scene(workload).to("cuda")
xyz = torch.nn.Parameter(s.point_cloud.clone())
feat = torch.nn.Parameter(s.point_cloud_features.clone())
gt = torch.rand(3, s.height, s.width, device="cuda")
ras = RAS(RAS.GaussianPointCloudRasterisationConfig(near_plane=s.near_plane, far_plane=s.far_plane,
                                                    depth_to_sort_key_scale=s.depth_to_sort_key_scale))
loss_fn = LossFunction(LossFunction.LossFunctionConfig())
opt_f = Adam([feat], lr=1e-3) if os.environ.get('GS_TORCH_ADAM') != '1' else torch.optim.Adam([feat], lr=1e-3, fused=True)
opt_p = Adam([xyz], lr=1e-5) if os.environ.get('GS_TORCH_ADAM') != '1' else torch.optim.Adam([xyz], lr=1e-5, fused=True)
if os.environ.get('GS_TORCH_ADAM') != '1':
    opt_f.set_scale_regulariser(feat, loss_fn.config.regularization_weight, s.point_invalid_mask)
cam = CameraInfo(camera_intrinsics=s.camera_intrinsics, camera_height=s.height, camera_width=s.width, camera_id=0)
marks = {}


def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e


for it in range(iters + 3):
    t0 = ev()
    opt_f.zero_grad(set_to_none=True); opt_p.zero_grad(set_to_none=True)
    image, depth, count = ras(RAS.GaussianPointCloudRasterisationInput(
        point_cloud=xyz, point_cloud_features=feat, point_object_id=s.point_object_id,
        point_invalid_mask=s.point_invalid_mask, camera_info=cam, q_pointcloud_camera=s.q_pointcloud_camera,
        t_pointcloud_camera=s.t_pointcloud_camera, color_max_sh_band=3))
    t1 = ev()
    loss, l1, ds = loss_fn(image.permute(2, 0, 1), gt, clamp_prediction=True)
    t2 = ev()
    loss.backward()   # the regulariser's gradient is applied inside the feature Adam kernel (as in the trainer)
    t3 = ev()
    opt_f.step(); opt_p.step()
    t4 = ev()
    if it >= 3:
        marks.setdefault("iters", []).append((t0, t1, t2, t3, t4))
torch.cuda.synchronize()
names = ["raster_fwd", "loss_fwd", "backward(loss+raster)", "adam"]
tot = 0.0
for i, nme in enumerate(names):
    ms = sum(m[i].elapsed_time(m[i + 1]) for m in marks["iters"]) / len(marks["iters"])
    tot += ms
    print(f"{nme:24s} {ms:8.3f} ms")
print(f"{'iteration':24s} {tot:8.3f} ms  -> {1000.0 / tot:.1f} it/s  ({workload})")
