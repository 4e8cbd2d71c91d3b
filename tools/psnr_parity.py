#!/usr/bin/env python
"""BASELINE config 5, PSNR parity between rasteriser back ends: the full GaussianPointCloudTrainer
loop -- the reference's schedule with densification and opacity resets switched on -- run to the END with the HIP
operator and with the CPU oracle (tests/helpers.OracleRasterisation: the checker, not the product) as its rasteriser,
on the same seeded synthetic multi-view sets, several seeds each, at a size the CPU can finish.

Per run: validation PSNR / SSIM every VAL_EVERY iterations and at the end, and the mean PSNR over ALL training views at
the end (the per-iteration train/psnr the trainer logs is the PSNR of ONE randomly drawn view and swings by several dB
from view to view -- the train / validation gap has to be read from means over the same number of views).

After the first densification the two back ends are two different (chaotic) trajectories of the same algorithm -- a
last-bit difference in one gradient changes which points split -- so parity is statistical: the back-end means must
agree within the spread over seeds.

Run through gpurun; writes gpurun_out/psnr_parity/summary.json.
usage: python tools/psnr_parity.py [iterations=2000] [size=256] [seeds=3] [backends=hip,oracle]"""
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
from tests.helpers import OracleRasterisation  # noqa: E402  (test infrastructure: the oracle as a back end)
from PIL import Image  # noqa: E402

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
SIZE = int(sys.argv[2]) if len(sys.argv) > 2 else 256
SEEDS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
BACKENDS = (sys.argv[4] if len(sys.argv) > 4 else "hip,oracle").split(",")
N_TRUE, N_VIEWS, VAL_EVERY = 6_000, 24, 500
dev = torch.device("cuda:0")
out_dir = os.path.join(ROOT, "gpurun_out", "psnr_parity")
os.makedirs(out_dir, exist_ok=True)


def make_dataset(seed: int) -> str:
    """24 views (20 train / 4 validation) of a seeded N_TRUE-Gaussian scene rendered by the HIP operator; training
    starts from a noisy third of the true positions with grey colours.  -> scratch directory"""
    data = tempfile.mkdtemp(prefix=f"gs_parity_{seed}_")
    gt = make_scene(n=N_TRUE, height=SIZE, width=SIZE, s_min=0.015, s_max=0.08, sh_degree=3, seed=100 + seed)
    gt.point_cloud_features[:, 7] = torch.rand(N_TRUE, generator=torch.Generator().manual_seed(5 + seed)) * 3.0
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
        Rwc = Ry @ Rx
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
    g = torch.Generator().manual_seed(1 + seed)
    keep = torch.randperm(N_TRUE, generator=g)[: N_TRUE // 3]
    init = gt.point_cloud[keep] + 0.02 * torch.randn(len(keep), 3, generator=g)
    pd.DataFrame(np.concatenate([init.numpy(), np.full((len(keep), 3), 128.0)], 1),
                 columns=["x", "y", "z", "r", "g", "b"]).to_parquet(os.path.join(data, "points.parquet"))
    return data


def make_config(data: str, tag: str, seed: int):
    cfg = TRN.TrainConfig(
        train_dataset_json_path=os.path.join(data, "train.json"), val_dataset_json_path=os.path.join(data, "val.json"),
        pointcloud_parquet_path=os.path.join(data, "points.parquet"), num_iterations=ITERS + 1,
        val_interval=VAL_EVERY, log_loss_interval=50, log_metrics_interval=10 ** 9, log_image_interval=10 ** 9,
        log_validation_image=False, summary_writer_log_dir=os.path.join(out_dir, tag), num_data_loader_workers=0,
        output_model_dir=os.path.join(data, f"checkpoints_{tag}"))
    # the reference's schedule (TRN:31-58, ADC:44-83) compressed to the run length: densification every 100 iterations
    # after a 300-iteration warm-up, opacity reset once in the run, SH band + 1 every ITERS / 4 iterations
    cfg.seed = seed
    cfg.increase_color_max_sh_band_interval = max(ITERS // 4, 1)
    cfg.adaptive_controller_config.num_iterations_warm_up = 300
    cfg.adaptive_controller_config.iteration_start_remove_floater = 300
    cfg.adaptive_controller_config.num_iterations_reset_alpha = max(ITERS // 2 - 50, 1)
    cfg.gaussian_point_cloud_scene_config.max_num_points_ratio = 8.0
    cfg.gaussian_point_cloud_scene_config.initial_alpha = 0.5      # the default (-2.0) is below the prune threshold
    return cfg


def curve(tag, name):
    path = os.path.join(out_dir, tag, "metrics.jsonl")
    return [(int(r["step"]), float(r["value"])) for r in map(json.loads, open(path)) if r["tag"] == name]


@torch.no_grad()
def mean_psnr(trainer, dataset) -> float:
    loader = torch.utils.data.DataLoader(dataset, batch_size=None, shuffle=False)
    vals = []
    for sample in loader:
        image_gt, q, t, info = trainer._to_device(sample)
        image_pred, _, _ = trainer._rasterise(q, t, info, band=3)
        psnr, _ = trainer._compute_pnsr_and_ssim(image_pred.clamp(0.0, 1.0).permute(2, 0, 1), image_gt)
        vals.append(float(psnr))
    return float(np.mean(vals))


runs = []
for seed in range(SEEDS):
    data = make_dataset(seed)
    for backend in BACKENDS:
        tag = f"{backend}_seed{seed}"
        torch.manual_seed(1000 + seed)
        np.random.seed(1000 + seed)
        t0 = time.perf_counter()
        trainer = TRN(make_config(data, tag, seed))
        if backend == "oracle":
            trainer.rasterisation = OracleRasterisation(trainer.config.rasterisation_config,
                                                        backward_valid_point_hook=trainer.adaptive_controller.update)
        trainer.train()
        torch.cuda.synchronize()
        val_psnr, val_ssim = dict(curve(tag, "val/psnr")), dict(curve(tag, "val/ssim"))
        last = max(val_psnr)
        rec = dict(backend=backend, seed=seed, seconds=round(time.perf_counter() - t0, 1),
                   live_points_end=int((trainer.scene.point_invalid_mask == 0).sum()),
                   val_psnr={k: round(v, 3) for k, v in sorted(val_psnr.items())},
                   val_ssim={k: round(v, 4) for k, v in sorted(val_ssim.items())},
                   val_psnr_end=round(val_psnr[last], 3), val_ssim_end=round(val_ssim[last], 4),
                   train_views_mean_psnr_end=round(mean_psnr(trainer, trainer.train_dataset), 3),
                   val_views_mean_psnr_end=round(mean_psnr(trainer, trainer.val_dataset), 3))
        runs.append(rec)
        print(json.dumps(rec), flush=True)

summary = dict(iterations=ITERS, image=f"{SIZE}x{SIZE}", views=N_VIEWS, true_gaussians=N_TRUE, seeds=SEEDS, runs=runs)
for backend in BACKENDS:
    mine = [r for r in runs if r["backend"] == backend]
    for key in ("val_psnr_end", "val_ssim_end", "train_views_mean_psnr_end", "live_points_end"):
        v = np.array([r[key] for r in mine], dtype=np.float64)
        summary[f"{backend}.{key}"] = dict(mean=round(float(v.mean()), 4), min=round(float(v.min()), 4),
                                           max=round(float(v.max()), 4), std=round(float(v.std(ddof=1)) if len(v) > 1 else 0.0, 4))
if set(BACKENDS) >= {"hip", "oracle"}:
    for key in ("val_psnr_end", "val_ssim_end", "train_views_mean_psnr_end"):
        h, o = summary[f"hip.{key}"], summary[f"oracle.{key}"]
        spread = max(h["max"] - h["min"], o["max"] - o["min"])
        summary[f"parity.{key}"] = dict(mean_difference=round(h["mean"] - o["mean"], 4), seed_spread=round(spread, 4),
                                        within_spread=bool(abs(h["mean"] - o["mean"]) <= spread))
with open(os.path.join(out_dir, "summary.json"), "w") as fh:
    json.dump(summary, fh, indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "runs"}, indent=1))
