#!/usr/bin/env python
"""BASELINE config 5, fallback form: the full GaussianPointCloudTrainer loop to 7,000 iterations with the
reference's schedule (progressive 4 -> 2 -> 1 down-sampling, SH band every 1000 iterations, densification every 100
after a 500-iteration warm-up, opacity resets), on a synthetic multi-view data set -- the Truck scene is not in the
container: 24 views (20 train / 4 validation) of a seeded 30,000-Gaussian scene at 800 x 800 rendered by the operator,
training starts from a noisy third of the true positions with grey colours.

  * HIP back end: all 7,000 iterations; validation PSNR / SSIM at 1000, 2000, ..., 5000, 7000 (TRN:266).
  * CPU-oracle back end (tests/helpers.OracleRasterisation; the checker, not the product): the first ORACLE_ITERS
    iterations of the SAME run (same seeds, data, loss kernels, optimiser): before the first densification the run is
    deterministic, so the two loss curves must coincide.

Run through gpurun; writes gpurun_out/train7k/{summary.json, hip_metrics.jsonl, oracle_metrics.jsonl}.
usage: python tools/train_7k.py [iterations=7000] [oracle_iterations=300] [size=800]"""
import json
import math
import os
import sys
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

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 7000
ORACLE_ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
SIZE = int(sys.argv[3]) if len(sys.argv) > 3 else 800
N_TRUE, N_VIEWS = 30_000, 24
dev = torch.device("cuda:0")
from taichi_3d_gaussian_splatting_amd import host_affinity  # noqa: E402
host_affinity.pin_host_threads(0)   # as gaussian_point_train.py: launching threads on one L3 complex next to the GPU
import tempfile  # noqa: E402
out_dir = os.path.join(ROOT, "gpurun_out", "train7k")
os.makedirs(out_dir, exist_ok=True)
data = tempfile.mkdtemp(prefix="gs_train7k_")          # images / json / parquet: scratch, not merged back
from PIL import Image  # noqa: E402

# ---------------------------------------------------------------- ground truth: renders of a known scene
gt = make_scene(n=N_TRUE, height=SIZE, width=SIZE, s_min=0.01, s_max=0.06, sh_degree=3, seed=21)
gt.point_cloud_features[:, 7] = torch.rand(N_TRUE, generator=torch.Generator().manual_seed(5)) * 3.0   # opacity 0.5 .. 0.95
K = torch.tensor([[0.9 * SIZE, 0, SIZE / 2], [0, 0.9 * SIZE, SIZE / 2], [0, 0, 1]])
ras = RAS(RAS.GaussianPointCloudRasterisationConfig())
records = {"train": [], "val": []}
for i in range(N_VIEWS):
    ang = 2 * math.pi * i / N_VIEWS
    elev = 0.25 * math.sin(3 * ang)
    c, s_ = math.cos(ang), math.sin(ang)
    Ry = torch.tensor([[c, 0, -s_], [0, 1, 0], [s_, 0, c]], dtype=torch.float32)
    ce, se = math.cos(elev), math.sin(elev)
    Rx = torch.tensor([[1, 0, 0], [0, ce, -se], [0, se, ce]], dtype=torch.float32)
    Rwc = Ry @ Rx                                              # camera looks at the origin from a wobbling ring
    T = torch.eye(4); T[:3, :3] = Rwc; T[:3, 3] = Rwc @ torch.tensor([0.0, 0.0, -3.6])
    q, t = SE3_to_quaternion_and_translation_torch(T.unsqueeze(0))
    with torch.no_grad():
        image, _, _ = ras(RAS.GaussianPointCloudRasterisationInput(
            point_cloud=gt.point_cloud.to(dev), point_cloud_features=gt.point_cloud_features.clone().to(dev),
            point_object_id=gt.point_object_id.to(dev), point_invalid_mask=gt.point_invalid_mask.to(dev),
            camera_info=CameraInfo(camera_intrinsics=K.to(dev), camera_height=SIZE, camera_width=SIZE, camera_id=0),
            q_pointcloud_camera=q.to(dev), t_pointcloud_camera=t.to(dev), color_max_sh_band=3))
    path = os.path.join(data, f"view_{i:02d}.png")
    Image.fromarray((image.clamp(0, 1).cpu().numpy() * 255 + 0.5).astype(np.uint8)).save(path)
    records["val" if i % 6 == 5 else "train"].append(dict(
        image_path=path, T_pointcloud_camera=T.tolist(), camera_intrinsics=K.tolist(), camera_height=SIZE,
        camera_width=SIZE, camera_id=0))
for split, recs in records.items():
    json.dump(recs, open(os.path.join(data, f"{split}.json"), "w"))
g = torch.Generator().manual_seed(1)
keep = torch.randperm(N_TRUE, generator=g)[: N_TRUE // 3]
init = gt.point_cloud[keep] + 0.02 * torch.randn(len(keep), 3, generator=g)
pd.DataFrame(np.concatenate([init.numpy(), np.full((len(keep), 3), 128.0)], 1),
             columns=["x", "y", "z", "r", "g", "b"]).to_parquet(os.path.join(data, "points.parquet"))


def make_config(tag, iterations):
    cfg = TRN.TrainConfig(
        train_dataset_json_path=os.path.join(data, "train.json"), val_dataset_json_path=os.path.join(data, "val.json"),
        pointcloud_parquet_path=os.path.join(data, "points.parquet"), num_iterations=iterations + 1,
        val_interval=1000, log_loss_interval=10, log_metrics_interval=100, log_image_interval=10 ** 9,
        log_validation_image=False, summary_writer_log_dir=os.path.join(out_dir, tag), num_data_loader_workers=0,
        output_model_dir=os.path.join(data, f"checkpoints_{tag}"))     # parquet checkpoints (19 MB each): scratch
    # everything else is the reference's default schedule (TRN:31-58, ADC:44-83): 4 -> 2 -> 1 down-sampling every 250
    # iterations, SH band + 1 every 1000, position lr decay 0.97 / 100, densify every 100 after 500, alpha reset 3000
    cfg.gaussian_point_cloud_scene_config.max_num_points_ratio = 8.0
    cfg.gaussian_point_cloud_scene_config.initial_alpha = 0.5      # the default (-2.0) is below the prune threshold
    return cfg


def curve(tag, name):
    path = os.path.join(out_dir, tag, "metrics.jsonl")
    return [(r["step"], r["value"]) for r in map(json.loads, open(path)) if r["tag"] == name]


summary = dict(iterations=ITERS, oracle_iterations=ORACLE_ITERS, image=f"{SIZE}x{SIZE}", views=N_VIEWS,
               true_gaussians=N_TRUE, initial_points=len(keep))
# ---------------------------------------------------------------- HIP back end, full length
t0 = time.perf_counter()
trainer = TRN(make_config("hip", ITERS))
if os.environ.get("GS_CPROFILE"):      # host-side profile of the loop (cProfile inflates the wall time)
    import cProfile
    import pstats
    prof = cProfile.Profile()
    prof.runcall(trainer.train)
    pstats.Stats(prof).sort_stats("tottime").print_stats(45)
else:
    trainer.train()
torch.cuda.synchronize()
summary["hip_seconds"] = round(time.perf_counter() - t0, 1)
summary["hip_it_per_s"] = round(ITERS / summary["hip_seconds"], 1)
summary["live_points_end"] = int((trainer.scene.point_invalid_mask == 0).sum())
summary["val_psnr"] = {int(s): round(v, 3) for s, v in curve("hip", "val/psnr")}
summary["val_ssim"] = {int(s): round(v, 4) for s, v in curve("hip", "val/ssim")}
train_psnr = curve("hip", "train/psnr")
summary["train_psnr"] = {int(s): round(v, 3) for s, v in train_psnr if s % 500 == 0}
summary["speculation"] = dict(trainer.rasterisation.speculation_stats)
print(json.dumps(summary), flush=True)

# ---------------------------------------------------------------- oracle back end, first ORACLE_ITERS iterations
if ORACLE_ITERS > 0:
    sys.path.insert(0, ROOT)
    from tests.helpers import OracleRasterisation
    t0 = time.perf_counter()
    cfg = make_config("oracle", ORACLE_ITERS)
    cfg.val_interval = 10 ** 9
    o_trainer = TRN(cfg)
    o_trainer.rasterisation = OracleRasterisation(cfg.rasterisation_config,
                                                  backward_valid_point_hook=o_trainer.adaptive_controller.update)
    o_trainer.train()
    summary["oracle_seconds"] = round(time.perf_counter() - t0, 1)
    hip_loss = dict(curve("hip", "train/loss"))
    ora_loss = dict(curve("oracle", "train/loss"))
    steps = sorted(s for s in ora_loss if s in hip_loss and s <= ORACLE_ITERS)
    rel = [abs(hip_loss[s] - ora_loss[s]) / abs(ora_loss[s]) for s in steps]
    summary["loss_curve_compared_steps"] = len(steps)
    summary["loss_curve_max_rel_diff"] = float(max(rel))
    summary["loss_curve_max_rel_diff_first_100"] = float(max(r for s, r in zip(steps, rel) if s <= 100))
    summary["loss_first_last"] = {"hip": [hip_loss[steps[0]], hip_loss[steps[-1]]],
                                  "oracle": [ora_loss[steps[0]], ora_loss[steps[-1]]]}
    d = (trainer.scene.point_cloud.shape, o_trainer.scene.point_cloud.shape)
    summary["capacity_rows"] = [int(d[0][0]), int(d[1][0])]
with open(os.path.join(out_dir, "summary.json"), "w") as fh:
    json.dump(summary, fh, indent=1)
for tag in ("hip", "oracle"):
    src = os.path.join(out_dir, tag, "metrics.jsonl")
    if os.path.exists(src):
        os.replace(src, os.path.join(out_dir, f"{tag}_metrics.jsonl"))
print(json.dumps(summary, indent=1))
