#!/usr/bin/env python
"""End-to-end throughput of GaussianPointCloudTrainer.train() (data path, rasteriser, fused loss, controller hook,
Adam, logging) on a synthetic multi-view data set rendered from a seeded scene.  Development tool; run through gpurun.
usage: python tools/train_throughput.py [n_points] [height] [width] [iterations]"""
import json
import math
import os
import sys
import tempfile
import time

import numpy as np
import pandas as pd
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as RAS  # noqa: E402
from taichi_3d_gaussian_splatting_amd.GaussianPointTrainer import GaussianPointCloudTrainer as TRN  # noqa: E402
from taichi_3d_gaussian_splatting_amd.synthetic import make_scene  # noqa: E402
from taichi_3d_gaussian_splatting_amd.utils import SE3_to_quaternion_and_translation_torch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1072
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 300
dev = torch.device("cuda:0")
root = tempfile.mkdtemp(prefix="gs_train_")
from PIL import Image  # noqa: E402

gt = make_scene(n=n, height=H, width=W, s_min=0.002, s_max=0.02, sh_degree=0, seed=21)
K = torch.tensor([[0.75 * W, 0, W / 2], [0, 0.75 * W, H / 2], [0, 0, 1]])
ras = RAS(RAS.GaussianPointCloudRasterisationConfig())
records = []
for i in range(8):
    ang = 0.15 * (i - 3.5)
    c, s_ = math.cos(ang), math.sin(ang)
    R = torch.tensor([[c, 0, -s_], [0, 1, 0], [s_, 0, c]], dtype=torch.float32)
    T = torch.eye(4); T[:3, :3] = R; T[:3, 3] = R @ torch.tensor([0.0, 0.0, -3.0])
    q, t = SE3_to_quaternion_and_translation_torch(T.unsqueeze(0))
    with torch.no_grad():
        image, _, _ = ras(RAS.GaussianPointCloudRasterisationInput(
            point_cloud=gt.point_cloud.to(dev), point_cloud_features=gt.point_cloud_features.clone().to(dev),
            point_object_id=gt.point_object_id.to(dev), point_invalid_mask=gt.point_invalid_mask.to(dev),
            camera_info=CameraInfo(camera_intrinsics=K.to(dev), camera_height=H, camera_width=W, camera_id=0),
            q_pointcloud_camera=q.to(dev), t_pointcloud_camera=t.to(dev), color_max_sh_band=0))
    path = os.path.join(root, f"view_{i}.png")
    Image.fromarray((image.clamp(0, 1).cpu().numpy() * 255 + 0.5).astype(np.uint8)).save(path)
    records.append(dict(image_path=path, T_pointcloud_camera=T.tolist(), camera_intrinsics=K.tolist(),
                        camera_height=H, camera_width=W, camera_id=0))
json.dump(records[:7], open(os.path.join(root, "train.json"), "w"))
json.dump(records[7:], open(os.path.join(root, "val.json"), "w"))
noisy = gt.point_cloud + 0.005 * torch.randn(n, 3)
pd.DataFrame(np.concatenate([noisy.numpy(), np.full((n, 3), 128.0)], 1),
             columns=["x", "y", "z", "r", "g", "b"]).to_parquet(os.path.join(root, "points.parquet"))

cfg = TRN.TrainConfig(
    train_dataset_json_path=os.path.join(root, "train.json"), val_dataset_json_path=os.path.join(root, "val.json"),
    pointcloud_parquet_path=os.path.join(root, "points.parquet"), num_iterations=iters, val_interval=10 ** 9,
    initial_downsample_factor=1, log_loss_interval=10, log_metrics_interval=100, log_image_interval=10 ** 9,
    summary_writer_log_dir=os.path.join(root, "logs"), num_data_loader_workers=0)
cfg.adaptive_controller_config.num_iterations_warm_up = 100
cfg.gaussian_point_cloud_scene_config.max_num_points_ratio = 1.2
cfg.gaussian_point_cloud_scene_config.initial_alpha = 0.5        # the default (-2.0) is below the default prune threshold
t0 = time.perf_counter()
trainer = TRN(cfg)
t1 = time.perf_counter()
stamps = []
_refine = trainer.adaptive_controller.refinement


def _stamped_refinement():
    _refine()
    if len(stamps) % 50 == 49:
        torch.cuda.synchronize()
    stamps.append(time.perf_counter())


trainer.adaptive_controller.refinement = _stamped_refinement
if os.environ.get("GS_PROFILE") == "1":
    import cProfile
    import pstats
    prof = cProfile.Profile()
    prof.enable()
    trainer.train()
    torch.cuda.synchronize()
    prof.disable()
    pstats.Stats(prof).sort_stats("tottime").print_stats(28)
else:
    trainer.train()
    torch.cuda.synchronize()
t2 = time.perf_counter()
live = int((trainer.scene.point_invalid_mask == 0).sum())
for a in range(49, len(stamps) - 50, 50):
    print(f"iterations {a + 1:4d}..{a + 50:4d}: {50 / (stamps[a + 50] - stamps[a]):7.1f} it/s")
print(f"setup {t1 - t0:.1f} s (parquet + KD-tree init of {n} points); train {iters} iterations in {t2 - t1:.2f} s "
      f"= {iters / (t2 - t1):.1f} it/s incl. first-iteration warm-up, image caching and 2 densifications; live points {live}")
