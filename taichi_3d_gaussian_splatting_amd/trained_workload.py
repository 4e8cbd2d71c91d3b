"""A TRAINED scene at full size as a bench / parity workload (``--workload trained_1080p``).

BASELINE configs 3 and 5 name the Tanks-and-Temples Truck scene (config/tat_truck_every_8_test.yaml); its data is not in
the container and there is no network.  The synthetic stand-ins of ``synthetic.py`` are uniformly random clouds of
isotropic-ish Gaussians; what the rasteriser meets in production is different: a cloud GROWN by the adaptive controller
(clones and splits where the view-space gradient is large, ADC:170-283), anisotropic after thousands of Adam steps,
with an opacity-reset history (ADC:285-288), depth-complex along the camera rays.  This module makes such a scene with
the repository's own trainer: a seeded ground-truth cloud is rendered from a ring of cameras at 1920 x 1072, a noisy,
grey subset of it is trained with the reference's schedule (4x -> 2x -> 1x down-sampling, SH band + 1 per 1000
iterations, densification every 100 iterations after 500, opacity reset at 3000) until the controller has grown it
past ``min_points`` live Gaussians, and the result is handed back as a ``SyntheticScene`` seen from one training camera.

Nothing is committed as data: the scene is a function of this script and the kernels (seeded throughout; ~1 minute on
an MI355X); ``load_or_make`` caches it under ``GS_TRAINED_SCENE_CACHE`` (default: the system temp directory).
"""
from __future__ import annotations

import json
import math
import os
import tempfile
import time
from typing import Optional

import numpy as np
import torch

from .synthetic import SyntheticScene, make_scene

WIDTH, HEIGHT = 1920, 1072      # the reference's /16 crop of 1080 (RAS:1193-1194, ImagePoseDataset.py:86-88)
N_TRUE, N_VIEWS = 300_000, 30
INIT_FRACTION = 0.1            # of the true positions (with noise) the training starts from: the cloud grows tenfold
DENSIFY_THRESHOLD = 1e-6       # ADC default 6e-6, tuned for ~1-Mpixel images: at 2 Mpixels the per-pixel loss gradient
                               # is half as large and the 30 k starting points would grow by a few hundred per refinement


def _cache_path(tag: str) -> str:
    root = os.environ.get("GS_TRAINED_SCENE_CACHE") or os.path.join(tempfile.gettempdir(), "gs_trained_scene")
    os.makedirs(root, exist_ok=True)
    return os.path.join(root, f"{tag}.pt")


def make_trained_scene(min_points: int = 300_000, width: int = WIDTH, height: int = HEIGHT, max_iterations: int = 6001,
                       n_true: int = N_TRUE, n_views: int = N_VIEWS, device: Optional[torch.device] = None,
                       verbose: bool = False, init_fraction: float = INIT_FRACTION,
                       densify_threshold: float = DENSIFY_THRESHOLD) -> dict:
    """-> {"scene": SyntheticScene (on the CPU), "stats": {...}}.  Needs a HIP device (it trains)."""
    import pandas as pd
    from PIL import Image
    from . import CameraInfo, GaussianPointCloudRasterisation as RAS
    from .GaussianPointTrainer import GaussianPointCloudTrainer as TRN
    from .utils import SE3_to_quaternion_and_translation_torch

    dev = device or torch.device("cuda", torch.cuda.current_device())
    data = tempfile.mkdtemp(prefix="gs_trained_scene_")
    t_start = time.perf_counter()
    # ---- ground truth: a shell-like cloud (dense surface, sparse interior) of anisotropic Gaussians, seen from a ring
    g = torch.Generator().manual_seed(11)
    gt = make_scene(n=n_true, height=height, width=width, s_min=0.004, s_max=0.03, sh_degree=3, seed=23)
    r = gt.point_cloud.norm(dim=1, keepdim=True).clamp_min(1e-6)
    shell = torch.rand(n_true, 1, generator=g) < 0.7               # 70 % of the points on a unit-ish shell: occlusion
    gt.point_cloud = torch.where(shell, gt.point_cloud / r * (0.85 + 0.1 * torch.rand(n_true, 1, generator=g)), gt.point_cloud)
    gt.point_cloud_features[:, 4:7] += torch.tensor([0.0, -0.8, 0.5]) * torch.rand(n_true, 1, generator=g)   # anisotropy
    gt.point_cloud_features[:, 7] = torch.rand(n_true, generator=g) * 4.0 - 0.5       # opacity 0.38 .. 0.97
    K = torch.tensor([[0.85 * width, 0, width / 2], [0, 0.85 * width, height / 2], [0, 0, 1]], dtype=torch.float32)
    ras = RAS(RAS.GaussianPointCloudRasterisationConfig())
    records = {"train": [], "val": []}
    poses = []
    for i in range(n_views):
        ang = 2 * math.pi * i / n_views
        elev = 0.3 * math.sin(2 * ang)
        c, s_ = math.cos(ang), math.sin(ang)
        Ry = torch.tensor([[c, 0, -s_], [0, 1, 0], [s_, 0, c]], dtype=torch.float32)
        ce, se = math.cos(elev), math.sin(elev)
        Rx = torch.tensor([[1, 0, 0], [0, ce, -se], [0, se, ce]], dtype=torch.float32)
        Rwc = Ry @ Rx
        T = torch.eye(4)
        T[:3, :3] = Rwc
        T[:3, 3] = Rwc @ torch.tensor([0.0, 0.0, -2.9])
        q, t = SE3_to_quaternion_and_translation_torch(T.unsqueeze(0))
        poses.append((q, t))
        with torch.no_grad():
            image, _, _ = ras(RAS.GaussianPointCloudRasterisationInput(
                point_cloud=gt.point_cloud.to(dev), point_cloud_features=gt.point_cloud_features.clone().to(dev),
                point_object_id=gt.point_object_id.to(dev), point_invalid_mask=gt.point_invalid_mask.to(dev),
                camera_info=CameraInfo(camera_intrinsics=K.to(dev), camera_height=height, camera_width=width, camera_id=0),
                q_pointcloud_camera=q.to(dev), t_pointcloud_camera=t.to(dev), color_max_sh_band=3))
        path = os.path.join(data, f"view_{i:02d}.png")
        Image.fromarray((image.clamp(0, 1).cpu().numpy() * 255 + 0.5).astype(np.uint8)).save(path, compress_level=1)
        records["val" if i % 10 == 9 else "train"].append(dict(
            image_path=path, T_pointcloud_camera=T.tolist(), camera_intrinsics=K.tolist(), camera_height=height,
            camera_width=width, camera_id=0))
    for split, recs in records.items():
        with open(os.path.join(data, f"{split}.json"), "w") as fh:
            json.dump(recs, fh)
    keep = torch.randperm(n_true, generator=g)[: max(int(n_true * init_fraction), 1000)]
    init = gt.point_cloud[keep] + 0.01 * torch.randn(len(keep), 3, generator=g)
    pd.DataFrame(np.concatenate([init.numpy(), np.full((len(keep), 3), 128.0)], 1),
                 columns=["x", "y", "z", "r", "g", "b"]).to_parquet(os.path.join(data, "points.parquet"))
    t_data = time.perf_counter()

    # ---- training with the reference's schedule, in chunks, until the controller has grown the cloud far enough
    cfg = TRN.TrainConfig(
        train_dataset_json_path=os.path.join(data, "train.json"), val_dataset_json_path=os.path.join(data, "val.json"),
        pointcloud_parquet_path=os.path.join(data, "points.parquet"), num_iterations=max_iterations,
        val_interval=10 ** 9, log_loss_interval=10 ** 9, log_metrics_interval=10 ** 9, log_image_interval=10 ** 9,
        log_validation_image=False, summary_writer_log_dir=os.path.join(data, "logs"), num_data_loader_workers=0,
        output_model_dir=os.path.join(data, "checkpoints"))
    cfg.gaussian_point_cloud_scene_config.max_num_points_ratio = max(8.0, 2.5 * min_points / max(len(keep), 1))
    cfg.gaussian_point_cloud_scene_config.initial_alpha = 0.5
    cfg.adaptive_controller_config.densification_view_space_position_gradients_threshold = densify_threshold
    trainer = TRN(cfg, device=dev)
    history = []
    original_refinement = trainer.adaptive_controller.refinement

    def refinement_and_watch(*a, **kw):   # after every densification: has the cloud grown far enough?
        out = original_refinement(*a, **kw)
        ctl = trainer.adaptive_controller
        if ctl.iteration_counter >= ctl.config.num_iterations_warm_up and \
                ctl.iteration_counter % ctl.config.num_iterations_densify == 0:
            live = int((trainer.scene.point_invalid_mask == 0).sum())
            history.append((ctl.iteration_counter, live))
            if verbose:
                print(f"[trained_scene] iteration {history[-1][0]}: {live} live Gaussians", flush=True)
        return out

    trainer.adaptive_controller.refinement = refinement_and_watch
    trainer.stop_when = lambda it: (len(history) > 0 and history[-1][1] >= min_points and it >= 3100 and it % 100 == 50)
    trainer.train()
    torch.cuda.synchronize()
    t_train = time.perf_counter()
    scene = trainer.scene
    live = (scene.point_invalid_mask == 0)
    xyz = scene.point_cloud.detach()[live].cpu().contiguous()
    feat = scene.point_cloud_features.detach()[live].cpu().contiguous()
    n = xyz.shape[0]
    q, t = poses[3]   # a training view
    out = SyntheticScene(
        point_cloud=xyz, point_cloud_features=feat, point_invalid_mask=torch.zeros(n, dtype=torch.int8),
        point_object_id=torch.zeros(n, dtype=torch.int32), camera_intrinsics=K, q_pointcloud_camera=q.cpu(),
        t_pointcloud_camera=t.cpu(), height=height, width=width)
    scales = feat[:, 4:7].exp()
    stats = dict(
        live_gaussians=n, iterations=int(trainer.iteration_reached), densifications=len(history),
        growth=[h[1] for h in history][:80], seconds_data=round(t_data - t_start, 1), seconds_training=round(t_train - t_data, 1),
        anisotropy_median=float((scales.max(dim=1).values / scales.min(dim=1).values).median()),
        anisotropy_p99=float((scales.max(dim=1).values / scales.min(dim=1).values).quantile(0.99)),
        opacity_median=float(torch.sigmoid(feat[:, 7]).median()), true_gaussians=n_true, views=n_views,
        speculation=dict(trainer.rasterisation.speculation_stats))
    return {"scene": out, "stats": stats, "poses": [(q.cpu(), t.cpu()) for q, t in poses]}


def load_or_make(tag: str = "trained_1080p", **kwargs) -> dict:
    path = _cache_path(tag)
    if os.path.exists(path):
        return torch.load(path, weights_only=False)
    made = make_trained_scene(**kwargs)
    torch.save(made, path)
    return made
