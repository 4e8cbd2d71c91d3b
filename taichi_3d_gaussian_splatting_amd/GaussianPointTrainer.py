"""Training loop around the rasteriser.

Mirror of the reference's ``taichi_3d_gaussian_splatting/GaussianPointTrainer.py`` (TRN): same ``TrainConfig``
fields and defaults (TRN:32-58), same schedule --

* two Adam optimisers, features / positions (TRN:126-129), exponential decay of the position learning rate every
  ``position_learning_rate_decay_interval`` iterations (TRN:131-132,181-182);
* coarse-to-fine images: start at 1/``initial_downsample_factor`` and halve the factor every
  ``half_downsample_factor_interval`` iterations, antialiased resize + crop to the 16-px tile grid (TRN:96-118,139-148);
* spherical-harmonics band ``iteration // increase_color_max_sh_band_interval`` (TRN:163);
* clamp to [0,1], HWC -> CHW, loss of ``LossFunction`` with the scale regulariser (TRN:167-175) -- clamp, L1, SSIM
  and their backward run as one fused HIP kernel pair (csrc/gs_loss.hip);
* the adaptive controller is the rasteriser's backward hook and refines after the optimiser steps (TRN:85-88,192);
* validation every ``val_interval`` iterations and at 5000 / 7000 (TRN:271-272): PSNR, SSIM, inference time,
  ``scene_{iteration}.parquet`` and ``best_scene.parquet`` (TRN:334-415).

MI355X-side choices: the scene AND the decoded training images live in HBM (``DeviceResidentSamples``: no image
decoding, host resize or PCIe copy inside the loop), a hand-written streaming Adam kernel (optim.py), no per-iteration host
synchronisation (the loss is only read back at the logging interval, so the "problematic iteration" check of
TRN:232-237 runs at that interval too), and -- when ``torch.distributed`` is initialised -- the rasteriser is
sharded over tile rows (distributed.py) while loss, optimiser and controller run replicated and bit-identical
on every rank.  TensorBoard and torchvision are optional: without ``tensorboard`` the scalars go to
``metrics.jsonl`` and images to PNG files under the log directory; resizing uses ``F.interpolate(antialias=True)``
(what torchvision's tensor resize calls).
"""
from __future__ import annotations

import json
import logging
import os
from collections import deque
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F

from .Camera import CameraInfo
from .GaussianPointAdaptiveController import GaussianPointAdaptiveController
from .GaussianPointCloudRasterisation import GaussianPointCloudRasterisation
from .GaussianPointCloudScene import GaussianPointCloudScene
from .ImagePoseDataset import ImagePoseDataset
from .LossFunction import LossFunction, ssim
from .optim import Adam
from .yaml_config import YAMLConfig

_log = logging.getLogger(__name__)
TILE = 16


def cycle(loader):
    while True:
        for item in loader:
            yield item


class _FileWriter:
    """Stand-in for ``torch.utils.tensorboard.SummaryWriter`` when tensorboard is not installed: scalars and
    histogram summaries as JSON lines, images as PNG files."""

    def __init__(self, log_dir: str):
        self.log_dir = log_dir
        self._fh = open(os.path.join(log_dir, "metrics.jsonl"), "a")

    def _emit(self, record: dict) -> None:
        self._fh.write(json.dumps(record) + "\n")
        self._fh.flush()

    def add_scalar(self, tag: str, value, step: int) -> None:
        self._emit({"tag": tag, "step": int(step), "value": float(value)})

    def add_histogram(self, tag: str, values: torch.Tensor, step: int) -> None:
        v = values.detach().float().flatten()
        if v.numel() == 0:
            return
        q = torch.quantile(v[:: max(1, v.numel() // 100000)], torch.tensor([0.0, 0.5, 1.0], device=v.device))
        self._emit({"tag": tag, "step": int(step), "mean": float(v.mean()), "min": float(q[0]),
                    "median": float(q[1]), "max": float(q[2])})

    def add_image(self, tag: str, chw: torch.Tensor, step: int) -> None:
        import PIL.Image
        arr = (chw.detach().clamp(0, 1).permute(1, 2, 0).cpu().numpy() * 255.0 + 0.5).astype(np.uint8)
        name = tag.replace("/", "_").replace(" ", "_")
        PIL.Image.fromarray(arr).save(os.path.join(self.log_dir, f"{name}_{int(step):07d}.png"))

    def close(self) -> None:
        self._fh.close()


def _make_writer(log_dir: str):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(log_dir=log_dir)
    except Exception:   # tensorboard missing (or broken): fall back to plain files
        return _FileWriter(log_dir)


class DeviceResidentSamples:
    """The training set decoded once and kept in HBM (MI355X: 288 GB per GPU; 250 full-HD float images are 6 GB).

    The reference streams every iteration's image through DataLoader workers and a host->device copy
    (TRN:122-150): PNG/JPEG decoding (tens of ms per full-HD image) and a 25 MB PCIe transfer per iteration are
    invisible next to its rasteriser, but would cap this trainer -- whose whole iteration takes ~2 ms -- far below
    what the GPU can do.  Iterating yields ``(image, q, t, CameraInfo)`` tuples of device tensors in a fresh seeded
    permutation per epoch (the same order on every rank)."""

    def __init__(self, loader, device: torch.device, seed: int):
        self.samples = []
        for image, q, t, info in loader:
            info = CameraInfo(camera_intrinsics=info.camera_intrinsics.to(device), camera_height=int(info.camera_height),
                              camera_width=int(info.camera_width), camera_id=info.camera_id)
            self.samples.append((image.to(device), q.to(device), t.to(device), info))
        self._generator = torch.Generator().manual_seed(seed)

    @staticmethod
    def estimated_bytes(dataset) -> int:
        return sum(3 * 4 * int(r["camera_height"]) * int(r["camera_width"]) for r in dataset.records)

    def __len__(self) -> int:
        return len(self.samples)

    def __iter__(self):
        for i in torch.randperm(len(self.samples), generator=self._generator).tolist():
            image, q, t, info = self.samples[i]
            # a fresh CameraInfo per use: callers may rescale the intrinsics
            yield image, q, t, CameraInfo(camera_intrinsics=info.camera_intrinsics, camera_height=info.camera_height,
                                          camera_width=info.camera_width, camera_id=info.camera_id)


def _image_grid(images, nrow: int = 2, pad: int = 2) -> torch.Tensor:
    """Minimal ``torchvision.utils.make_grid`` for equally sized [3,H,W] tensors."""
    h, w = images[0].shape[1:]
    ncol = (len(images) + nrow - 1) // nrow
    grid = torch.zeros(3, ncol * (h + pad) + pad, nrow * (w + pad) + pad, device=images[0].device)
    for i, im in enumerate(images):
        r, c = divmod(i, nrow)
        grid[:, pad + r * (h + pad): pad + r * (h + pad) + h, pad + c * (w + pad): pad + c * (w + pad) + w] = im
    return grid


class _OwnerRasterisationFacade:
    """The few attributes the training loop touches on its rasteriser, for the owner-sharded module (whose options live on
    the rank's ``core``)."""

    def __init__(self, module):
        self.module, self.core = module, module.core

    def __call__(self, input_data):
        return self.module(input_data)

    @property
    def hook_feature_gradients(self):
        return self.core.hook_feature_gradients

    @hook_feature_gradients.setter
    def hook_feature_gradients(self, value):
        self.core.hook_feature_gradients = bool(value)

    @property
    def speculation_stats(self):
        return self.core.speculation_stats


class GaussianPointCloudTrainer:
    @dataclass
    class TrainConfig(YAMLConfig):
        train_dataset_json_path: str = ""
        val_dataset_json_path: str = ""
        pointcloud_parquet_path: str = ""
        num_iterations: int = 300000
        val_interval: int = 1000
        feature_learning_rate: float = 1e-3
        position_learning_rate: float = 1e-5
        position_learning_rate_decay_rate: float = 0.97
        position_learning_rate_decay_interval: int = 100
        increase_color_max_sh_band_interval: int = 1000
        log_loss_interval: int = 10
        log_metrics_interval: int = 100
        print_metrics_to_console: bool = False
        log_image_interval: int = 1000
        enable_taichi_kernel_profiler: bool = False      # accepted for YAML compatibility, unused (no Taichi)
        log_taichi_kernel_profile_interval: int = 1000   # idem
        log_validation_image: bool = True
        initial_downsample_factor: int = 4
        half_downsample_factor_interval: int = 250
        summary_writer_log_dir: str = "logs"
        output_model_dir: Optional[str] = None
        rasterisation_config: GaussianPointCloudRasterisation.GaussianPointCloudRasterisationConfig = field(
            default_factory=GaussianPointCloudRasterisation.GaussianPointCloudRasterisationConfig)
        adaptive_controller_config: GaussianPointAdaptiveController.GaussianPointAdaptiveControllerConfig = field(
            default_factory=GaussianPointAdaptiveController.GaussianPointAdaptiveControllerConfig)
        gaussian_point_cloud_scene_config: GaussianPointCloudScene.PointCloudSceneConfig = field(
            default_factory=GaussianPointCloudScene.PointCloudSceneConfig)
        loss_function_config: LossFunction.LossFunctionConfig = field(
            default_factory=LossFunction.LossFunctionConfig)
        # additions (not in the reference)
        num_data_loader_workers: int = 4
        seed: int = 0
        # multi-GPU (torch.distributed initialised): "replicated" = every rank holds the whole point cloud and the rasteriser
        # is sharded over tile rows (distributed.py; loss, optimiser and controller run replicated and bit-identical);
        # "owner" = every rank OWNS a contiguous block of the point cloud -- its parameters, its Adam state, its share of
        # the controller's work -- and the rasteriser routes projected records / accumulator rows between the ranks
        # (owner_sharding.py): nothing per-Gaussian is replicated
        distributed_mode: str = "replicated"
        # "owner" mode: every this-many iterations the ranks sum their walk lengths per tile row and move the band
        # boundaries if the heaviest band carries more than 1.15x the mean (owner_sharding.balanced_row_weights); 0 = equal
        # bands.  One view's weights place the boundaries for the following views too: scenes crowd the same rows from
        # most cameras of a capture
        owner_rebalance_every: int = 100
        cache_dataset_on_device: bool = True      # keep the decoded training images in HBM (DeviceResidentSamples)
        device_cache_max_gb: float = 128.0        # ... unless they would need more than this; then stream them

    def __init__(self, config: "GaussianPointCloudTrainer.TrainConfig", device: Optional[torch.device] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("GaussianPointCloudTrainer needs a HIP device: the rasteriser has no CPU path")
        self.config = config
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
        os.makedirs(config.summary_writer_log_dir, exist_ok=True)
        if config.output_model_dir is None:
            config.output_model_dir = config.summary_writer_log_dir
        os.makedirs(config.output_model_dir, exist_ok=True)
        self.writer = _make_writer(config.summary_writer_log_dir) if self.rank == 0 else None
        torch.manual_seed(config.seed)   # the same stream on every rank keeps replicated refinement identical

        self.train_dataset = ImagePoseDataset(dataset_json_path=config.train_dataset_json_path)
        self.val_dataset = ImagePoseDataset(dataset_json_path=config.val_dataset_json_path)
        self.scene = GaussianPointCloudScene.from_parquet(
            config.pointcloud_parquet_path, config=config.gaussian_point_cloud_scene_config).to(self.device)
        if torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            # one authoritative replica: whatever randomness went into the construction (KD-tree initialisation, the
            # sky sphere), every rank starts from rank 0's tensors
            with torch.no_grad():
                for t in (self.scene.point_cloud, self.scene.point_cloud_features, self.scene.point_invalid_mask,
                          self.scene.point_object_id):
                    torch.distributed.broadcast(t.data if isinstance(t, torch.nn.Parameter) else t, src=0)
        self.owner_sharded = bool(torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1 and
                                  config.distributed_mode == "owner")
        if config.distributed_mode not in ("replicated", "owner"):
            raise ValueError(f"distributed_mode {config.distributed_mode!r}: 'replicated' or 'owner'")
        if self.owner_sharded:
            # this rank keeps ITS block of the (fixed-capacity) point cloud: parameters, mask, ids -- and with them the
            # optimiser state and the controller's statistics, which are created from these tensors below
            # (the live points sit at the front of the fixed-capacity tensors: the rank takes its contiguous share of the
            #  LIVE rows and of the free rows, not a block of the capacity -- else the last ranks would own empty rows only)
            from .owner_sharding import owned_point_rows
            world = torch.distributed.get_world_size()
            live_rows = torch.nonzero(self.scene.point_invalid_mask == 0).flatten()
            free_rows = torch.nonzero(self.scene.point_invalid_mask != 0).flatten()
            mine_live = owned_point_rows(live_rows.shape[0], self.rank, world)
            mine_free = owned_point_rows(free_rows.shape[0], self.rank, world)
            rows = torch.cat([live_rows[mine_live.start:mine_live.stop], free_rows[mine_free.start:mine_free.stop]])
            with torch.no_grad():
                self.scene.point_cloud = torch.nn.Parameter(self.scene.point_cloud.detach()[rows].clone())
                self.scene.point_cloud_features = torch.nn.Parameter(self.scene.point_cloud_features.detach()[rows].clone())
                self.scene.point_invalid_mask = self.scene.point_invalid_mask[rows].clone()
                self.scene.point_object_id = self.scene.point_object_id[rows].clone()
        self.adaptive_controller = GaussianPointAdaptiveController(
            config=config.adaptive_controller_config,
            maintained_parameters=GaussianPointAdaptiveController.GaussianPointAdaptiveControllerMaintainedParameters(
                pointcloud=self.scene.point_cloud,
                pointcloud_features=self.scene.point_cloud_features,
                point_invalid_mask=self.scene.point_invalid_mask,
                point_object_id=self.scene.point_object_id))
        if self.owner_sharded:
            from .owner_sharding import OwnerShardedRasterisation
            module = OwnerShardedRasterisation(config.rasterisation_config,
                                               backward_valid_point_hook=self.adaptive_controller.update)
            module.rebalance_every = int(config.owner_rebalance_every)
            self.rasterisation = _OwnerRasterisationFacade(module)
        else:
            self.rasterisation = GaussianPointCloudRasterisation(
                config=config.rasterisation_config, backward_valid_point_hook=self.adaptive_controller.update)
            if torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
                from .distributed import shard_rasteriser_across_tile_rows
                shard_rasteriser_across_tile_rows(self.rasterisation)
        self.loss_function = LossFunction(config=config.loss_function_config)
        self.best_psnr_score = 0.0

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _downsample_image_and_camera_info(image: torch.Tensor, camera_info: CameraInfo, downsample_factor: int):
        """Antialiased resize by 1/factor, crop to whole tiles, intrinsics divided by the factor (TRN:96-118)."""
        h = camera_info.camera_height // downsample_factor
        w = camera_info.camera_width // downsample_factor
        image = F.interpolate(image[None], size=(h, w), mode="bilinear", antialias=True, align_corners=False)[0]
        h, w = h - h % TILE, w - w % TILE
        image = image[:3, :h, :w].contiguous()
        K = camera_info.camera_intrinsics.clone()
        K[0, 0] /= downsample_factor; K[1, 1] /= downsample_factor
        K[0, 2] /= downsample_factor; K[1, 2] /= downsample_factor
        return image, CameraInfo(camera_intrinsics=K, camera_height=h, camera_width=w,
                                 camera_id=camera_info.camera_id)

    @staticmethod
    def _easy_cmap(depth: torch.Tensor) -> torch.Tensor:
        """Three-range depth colouring (0-10 / 10-60 / 60-260), inverted (TRN:274-280)."""
        rgb = torch.stack([depth.clamp(0, 10) / 10.0, (depth - 10).clamp(0, 50) / 50.0,
                           (depth - 60).clamp(0, 200) / 200.0])
        return 1.0 - rgb

    @staticmethod
    @torch.no_grad()
    def _compute_pnsr_and_ssim(image_pred: torch.Tensor, image_gt: torch.Tensor):
        psnr = 10.0 * torch.log10(1.0 / torch.mean((image_pred - image_gt) ** 2))
        return psnr, ssim(image_pred.unsqueeze(0), image_gt.unsqueeze(0), data_range=1.0, size_average=True)

    def _to_device(self, sample):
        image, q, t, info = sample
        info = CameraInfo(camera_intrinsics=info.camera_intrinsics.to(self.device, non_blocking=True),
                          camera_height=int(info.camera_height), camera_width=int(info.camera_width),
                          camera_id=info.camera_id)
        return (image.to(self.device, non_blocking=True), q.to(self.device, non_blocking=True),
                t.to(self.device, non_blocking=True), info)

    def _rasterise(self, q, t, info, band: int):
        scene = self.scene
        return self.rasterisation(GaussianPointCloudRasterisation.GaussianPointCloudRasterisationInput(
            point_cloud=scene.point_cloud, point_cloud_features=scene.point_cloud_features,
            point_object_id=scene.point_object_id, point_invalid_mask=scene.point_invalid_mask,
            camera_info=info, q_pointcloud_camera=q, t_pointcloud_camera=t, color_max_sh_band=band))

    def _whole_scene(self):
        """The scene to checkpoint: this rank's own when nothing is sharded; under owner sharding rank 0 receives every
        rank's block (fixed capacity, so equal sizes up to the last block: padded with invalid rows) and assembles them."""
        if not self.owner_sharded:
            return self.scene
        import torch.distributed as dist
        world = dist.get_world_size()
        n = torch.tensor([self.scene.point_cloud.shape[0]], dtype=torch.int64, device=self.device)
        sizes = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(sizes, n)
        cap = int(max(int(x) for x in sizes))

        def padded(t, fill):
            out = torch.full((cap,) + tuple(t.shape[1:]), fill, dtype=t.dtype, device=self.device)
            out[:t.shape[0]] = t.detach()
            return out

        parts = [padded(self.scene.point_cloud, 0.0), padded(self.scene.point_cloud_features, 0.0),
                 padded(self.scene.point_invalid_mask, 1), padded(self.scene.point_object_id, 0)]
        gathered = []
        for t in parts:
            buf = [torch.empty_like(t) for _ in range(world)] if self.rank == 0 else None
            dist.gather(t, buf, dst=0)
            gathered.append(buf)
        if self.rank != 0:
            return None
        import types
        whole = types.SimpleNamespace(point_cloud=torch.cat(gathered[0]), point_cloud_features=torch.cat(gathered[1]),
                                      point_invalid_mask=torch.cat(gathered[2]), point_object_id=torch.cat(gathered[3]))
        whole.to_parquet = lambda path: GaussianPointCloudScene.to_parquet(whole, path)   # (reads exactly these tensors)
        return whole

    def _scalar(self, tag: str, value: float, step: int, console_key: Optional[str] = None) -> None:
        if self.writer is not None:
            self.writer.add_scalar(tag, value, step)
        if console_key and self.config.print_metrics_to_console and self.rank == 0:
            print(f"{console_key}={value};")   # the reference's console format (scraped by its CI, TRN:213-231,403-409)
            if console_key.endswith(("_psnr", "_ssim")):   # plus the per-iteration keys, e.g. train_psnr_7000= (TRN:229)
                print(f"{console_key}_{step}={value};")

    def _loaders(self):
        import functools
        from .host_affinity import original_mask, reset_worker_affinity   # the main thread may be pinned to one L3 complex
        reset_worker_affinity = functools.partial(reset_worker_affinity, mask=original_mask())   # (also for spawned workers)
        kw = dict(batch_size=None, pin_memory=True, num_workers=self.config.num_data_loader_workers,
                  worker_init_fn=reset_worker_affinity if self.config.num_data_loader_workers > 0 else None)
        generator = torch.Generator().manual_seed(self.config.seed)   # same shuffling on every rank
        return (torch.utils.data.DataLoader(self.train_dataset, shuffle=True, generator=generator, **kw),
                torch.utils.data.DataLoader(self.val_dataset, shuffle=False, **kw))

    def _local_live_share(self) -> float:
        """Owner-sharded Gaussians: n_live(this rank's block) / n_live(all blocks), as the scale regulariser needs it (a mean
        over ALL live Gaussians, LOS:42-54).  The live set only changes when the controller densifies or prunes, so the two
        counts are re-read (one all-reduce, one host read-back) only when its densification counter has moved."""
        version = self.adaptive_controller.mask_version
        cached = getattr(self, "_live_share_cache", None)
        if cached is None or cached[0] != version:
            counts = (self.scene.point_invalid_mask == 0).sum().to(torch.float64).reshape(1)
            total = counts.clone()
            torch.distributed.all_reduce(total)
            cached = (version, float(counts.item()) / max(float(total.item()), 1.0))
            self._live_share_cache = cached
        return cached[1]

    # ------------------------------------------------------------------ training (TRN:120-272)
    def train(self):
        cfg = self.config
        train_loader, val_loader = self._loaders()
        if cfg.cache_dataset_on_device and \
                DeviceResidentSamples.estimated_bytes(self.train_dataset) <= cfg.device_cache_max_gb * 2 ** 30:
            train_loader = DeviceResidentSamples(train_loader, self.device, cfg.seed)
            _log.info("training set resident on %s: %d images", self.device, len(train_loader))
        batches = cycle(train_loader)
        feature_optimizer = Adam([self.scene.point_cloud_features], lr=cfg.feature_learning_rate, betas=(0.9, 0.999))
        position_optimizer = Adam([self.scene.point_cloud], lr=cfg.position_learning_rate, betas=(0.9, 0.999))
        # fixed-capacity tensors (max_num_points_ratio): the rows without a point are not stepped
        feature_optimizer.set_row_mask(self.scene.point_cloud_features, self.scene.point_invalid_mask)
        position_optimizer.set_row_mask(self.scene.point_cloud, self.scene.point_invalid_mask)
        regularised = self.loss_function.config.enable_regularization
        if regularised:
            feature_optimizer.set_scale_regulariser(self.scene.point_cloud_features,
                                                    self.loss_function.config.regularization_weight,
                                                    self.scene.point_invalid_mask,
                                                    local_share=self._local_live_share if self.owner_sharded else None)
        scheduler = torch.optim.lr_scheduler.ExponentialLR(position_optimizer,
                                                           gamma=cfg.position_learning_rate_decay_rate)
        downsample_factor = cfg.initial_downsample_factor
        recent_losses = deque(maxlen=100)
        last_problematic = -1000

        for iteration in range(cfg.num_iterations):
            self.iteration_reached = iteration
            # (addition: tools that train until a condition holds -- trained_workload.py -- set `stop_when`)
            if getattr(self, "stop_when", None) is not None and self.stop_when(iteration):
                break
            if iteration > 0 and iteration % cfg.half_downsample_factor_interval == 0 and downsample_factor > 1:
                downsample_factor //= 2
            feature_optimizer.zero_grad(set_to_none=True)
            position_optimizer.zero_grad(set_to_none=True)
            image_gt, q, t, info = next(batches)
            image_gt, q, t, info = self._to_device((image_gt, q, t, info))   # no-op for resident samples
            if downsample_factor > 1:   # on the device (the reference resizes on the host before the copy)
                image_gt, info = self._downsample_image_and_camera_info(image_gt, info, downsample_factor)

            band = int(iteration // cfg.increase_color_max_sh_band_interval)
            # the compact per-feature gradients of the hook are only looked at (histograms) on densification iterations
            self.rasterisation.hook_feature_gradients = self.adaptive_controller.next_update_selects()
            image_pred, image_depth, pixel_valid_point_count = self._rasterise(q, t, info, band)
            # clamp (TRN:168) is folded into the fused loss kernel; the permute (TRN:170) is a view
            loss, l1_loss, ssim_loss = self.loss_function(image_pred.permute(2, 0, 1), image_gt, clamp_prediction=True)
            loss.backward()
            # scale regulariser (TRN:172-174, LOS:36-38): its gradient is added inside the feature Adam kernel (same
            # gradient, from the same pre-step parameters); its VALUE is only needed where the loss is logged
            loss = loss.detach()
            if regularised and iteration % cfg.log_loss_interval == 0:
                reg_value = self.loss_function.regularization_value(self.scene.point_invalid_mask,
                                                                    self.scene.point_cloud_features)
                if self.owner_sharded:   # mean over ALL live Gaussians: the blocks' means weighted by their share
                    reg_value = reg_value * self._local_live_share()
                    torch.distributed.all_reduce(reg_value)
                loss = loss + reg_value
            raw_pred = image_pred.detach()
            image_pred = None   # the clamped CHW copy is only materialised on logging iterations (below)
            feature_optimizer.step()
            position_optimizer.step()
            if iteration % cfg.position_learning_rate_decay_interval == 0:
                scheduler.step()

            hook_input = self.adaptive_controller.input_data     # set on densification iterations only
            grad_image = None
            if hook_input is not None:
                grad_image = hook_input.magnitude_grad_viewspace_on_image
                if self.writer is not None:
                    self._plot_grad_histogram(hook_input, self.writer, iteration)
                    self._plot_value_histogram(self.scene, self.writer, iteration)
                    self.writer.add_histogram("train/pixel_valid_point_count", pixel_valid_point_count, iteration)
            self.adaptive_controller.refinement()

            is_problematic = False
            if iteration % cfg.log_loss_interval == 0:
                # the only regular host read-back of the loop: one copy for the three scalars
                loss_value, l1_value, ssim_value = torch.stack([loss, l1_loss.detach(), ssim_loss.detach()]).tolist()
                if cfg.print_metrics_to_console and self.rank == 0:
                    print(f"train_iteration={iteration};")   # TRN:213
                self._scalar("train/loss", loss_value, iteration, "train_loss")
                self._scalar("train/l1 loss", l1_value, iteration, "train_l1_loss")
                self._scalar("train/ssim loss", ssim_value, iteration, "train_ssim_loss")
                if len(recent_losses) == recent_losses.maxlen and \
                        iteration - last_problematic > recent_losses.maxlen * cfg.log_loss_interval and \
                        loss_value > 1.5 * sum(recent_losses) / len(recent_losses):
                    is_problematic, last_problematic = True, iteration
                recent_losses.append(loss_value)
            log_image = (iteration % cfg.log_image_interval == 0 or is_problematic) and self.writer is not None
            if iteration % cfg.log_metrics_interval == 0 or log_image:
                image_pred = raw_pred.clamp(0.0, 1.0).permute(2, 0, 1)
            if iteration % cfg.log_metrics_interval == 0:
                psnr, ssim_score = self._compute_pnsr_and_ssim(image_pred, image_gt)
                self._scalar("train/psnr", psnr.item(), iteration, "train_psnr")
                self._scalar("train/ssim", ssim_score.item(), iteration, "train_ssim")
            if log_image:
                panels = [image_pred, image_gt, self._easy_cmap(image_depth),
                          self._count_panel(pixel_valid_point_count)]
                if grad_image is not None:
                    g = grad_image.permute(2, 0, 1)
                    panels += [(g[0] / g[0].max().clamp_min(1e-30)).expand(3, -1, -1),
                               (g[1] / g[1].max().clamp_min(1e-30)).expand(3, -1, -1),
                               (image_pred - image_gt).abs()]
                self.writer.add_image("train/image_problematic" if is_problematic else "train/image",
                                      _image_grid(panels), iteration)
            del image_gt, image_pred, raw_pred, image_depth, pixel_valid_point_count, loss, l1_loss, ssim_loss
            if (iteration % cfg.val_interval == 0 and iteration != 0) or iteration in (5000, 7000):
                self.validation(val_loader, iteration)
        if self.writer is not None and hasattr(self.writer, "flush"):
            self.writer.flush()

    @staticmethod
    def _count_panel(count: torch.Tensor) -> torch.Tensor:
        c = count.float()
        return (c / c.max().clamp_min(1.0)).unsqueeze(0).expand(3, -1, -1)

    @staticmethod
    @torch.no_grad()
    def _plot_grad_histogram(grad_input, writer, iteration: int) -> None:
        f = grad_input.grad_pointfeatures_in_camera
        for tag, values in (("grad/xyz_grad", grad_input.grad_point_in_camera), ("grad/uv_grad", grad_input.grad_viewspace),
                            ("grad/q_grad", f[:, :4]), ("grad/s_grad", f[:, 4:7]), ("grad/alpha_grad", f[:, 7]),
                            ("grad/r_grad", f[:, 8:24]), ("grad/g_grad", f[:, 24:40]), ("grad/b_grad", f[:, 40:56]),
                            ("value/num_overlap_tiles", grad_input.num_overlap_tiles),
                            ("value/num_affected_pixels", grad_input.num_affected_pixels)):
            writer.add_histogram(tag, values, iteration)

    @staticmethod
    @torch.no_grad()
    def _plot_value_histogram(scene: GaussianPointCloudScene, writer, iteration: int) -> None:
        live = scene.point_invalid_mask == 0
        f = scene.point_cloud_features[live]
        writer.add_scalar("value/num_valid_points", int(live.sum()), iteration)
        for tag, values in (("value/q", f[:, :4]), ("value/s", f[:, 4:7]), ("value/alpha", f[:, 7]),
                            ("value/sigmoid_alpha", torch.sigmoid(f[:, 7])), ("value/r", f[:, 8:24]),
                            ("value/g", f[:, 24:40]), ("value/b", f[:, 40:56])):
            writer.add_histogram(tag, values, iteration)

    # ------------------------------------------------------------------ validation (TRN:334-415)
    @torch.no_grad()
    def validation(self, val_loader, iteration: int):
        totals = {"loss": 0.0, "psnr": 0.0, "ssim": 0.0, "ms": 0.0}
        n = 0
        for idx, sample in enumerate(val_loader):
            image_gt, q, t, info = self._to_device(sample)
            start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            image_pred, image_depth, count = self._rasterise(q, t, info, band=3)
            stop.record()
            torch.cuda.synchronize()
            totals["ms"] += start.elapsed_time(stop)
            image_pred = image_pred.clamp(0.0, 1.0).permute(2, 0, 1)
            loss, _, _ = self.loss_function(image_pred, image_gt)
            psnr, ssim_score = self._compute_pnsr_and_ssim(image_pred, image_gt)
            totals["loss"] += loss.item(); totals["psnr"] += psnr.item(); totals["ssim"] += ssim_score.item()
            n += 1
            if self.config.log_validation_image and self.writer is not None:
                self.writer.add_image(f"val/image {idx}", _image_grid(
                    [image_pred, image_gt, self._easy_cmap(image_depth), self._count_panel(count),
                     (image_pred - image_gt).abs()]), iteration)
        n = max(n, 1)
        mean = {k: v / n for k, v in totals.items()}
        for tag, key, console in (("val/loss", "loss", "val_loss"), ("val/psnr", "psnr", "val_psnr"),
                                  ("val/ssim", "ssim", "val_ssim"), ("val/inference_time", "ms", "val_inference_time")):
            self._scalar(tag, mean[key], iteration, console)
        scene = self._whole_scene()     # (owner-sharded: the ranks' blocks, gathered on rank 0; a collective)
        if self.rank == 0:
            scene.to_parquet(os.path.join(self.config.output_model_dir, f"scene_{iteration}.parquet"))
            if mean["psnr"] > self.best_psnr_score:
                scene.to_parquet(os.path.join(self.config.output_model_dir, "best_scene.parquet"))
        self.best_psnr_score = max(self.best_psnr_score, mean["psnr"])
        return mean
