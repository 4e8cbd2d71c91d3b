"""Adaptive density control around the rasteriser.

Mirror of the reference's ``taichi_3d_gaussian_splatting/GaussianPointAdaptiveController.py`` (ADC).  The point
cloud has a fixed capacity; ``point_invalid_mask`` says which rows are live (ADC:46-51).  The controller is the
consumer of the rasteriser's backward hook (``BackwardValidPointHookInput``, RAS:806-817):

* ``update(hook_input)``  -- called inside ``backward`` (ADC:130-146): per-Gaussian statistics are accumulated
  and, on a densification iteration, the rows to delete / clone / split are chosen *before* the optimiser step
  so that the pre-step positions can be recorded (ADC:169-267);
* ``refinement()``        -- called after the optimiser step (ADC:148-167): deletes transparent Gaussians and
  floaters, writes the clones / splits into free rows (ADC:289-353), clears the statistics and periodically
  clamps the opacity logits (ADC:355-358).

Same config fields, dataclasses, method names and decision rules as the reference.  The two device kernels the
reference runs through Taichi (ADC:10-42) are the HIP kernels ``gs_ellipsoid_offsets`` /
``gs_sample_from_points`` (csrc/gs_controller.hip).  Differences: the matplotlib scatter of the chosen points
(ADC:268-287) is replaced by ``last_densify_uv`` (a dict of uv tensors a caller may plot); decisions are logged
through ``logging`` instead of ``print``.
"""
from __future__ import annotations

import logging
import math
from dataclasses import dataclass
from typing import Callable, Dict, Optional

import torch

from .GaussianPointCloudRasterisation import GaussianPointCloudRasterisation
from .yaml_config import YAMLConfig

_log = logging.getLogger(__name__)
HookInput = GaussianPointCloudRasterisation.BackwardValidPointHookInput


def _ratio(numerator: torch.Tensor, denominator: torch.Tensor) -> torch.Tensor:
    """numerator / denominator with 0/0 -> 0 (the reference divides, then overwrites the NaNs)."""
    q = numerator / denominator
    return torch.where(torch.isnan(q), torch.zeros_like(q), q)


def _hip_sample_from_point(xyz: torch.Tensor, features: torch.Tensor) -> torch.Tensor:
    from . import hip_ops
    return hip_ops.sample_from_points(xyz.contiguous(), features.contiguous())


def _hip_ellipsoid_offset(features: torch.Tensor) -> torch.Tensor:
    from . import hip_ops
    return hip_ops.ellipsoid_offsets(features.contiguous())


class GaussianPointAdaptiveController:
    @dataclass
    class GaussianPointAdaptiveControllerConfig(YAMLConfig):
        num_iterations_warm_up: int = 500
        num_iterations_densify: int = 100
        transparent_alpha_threshold: float = -0.5            # on the opacity logit
        densification_view_space_position_gradients_threshold: float = 6e-6
        densification_view_avg_space_position_gradients_threshold: float = 1e3
        densification_multi_frame_view_space_position_gradients_threshold: float = 1e3
        densification_multi_frame_view_pixel_avg_space_position_gradients_threshold: float = 1e3
        densification_multi_frame_position_gradients_threshold: float = 1e3
        gaussian_split_factor_phi: float = 1.6
        num_iterations_reset_alpha: int = 3000
        reset_alpha_value: float = 0.1
        floater_num_pixels_threshold: int = 10000
        floater_near_camrea_num_pixels_threshold: int = 10000   # (sic) the reference's spelling, kept for YAML
        floater_depth_threshold: float = 100
        iteration_start_remove_floater: int = 2000
        plot_densify_interval: int = 200
        under_reconstructed_num_pixels_threshold: int = 512
        under_reconstructed_move_factor: float = 100.0
        enable_ellipsoid_offset: bool = False
        enable_sample_from_point: bool = True

    @dataclass
    class GaussianPointAdaptiveControllerMaintainedParameters:
        pointcloud: torch.Tensor            # [N,3]
        pointcloud_features: torch.Tensor   # [N,56]
        point_invalid_mask: torch.Tensor    # int8[N]
        point_object_id: torch.Tensor       # int32[N]

    @dataclass
    class GaussianPointAdaptiveControllerDensifyPointInfo:
        floater_point_id: torch.Tensor
        transparent_point_id: torch.Tensor
        densify_point_id: torch.Tensor
        densify_point_position_before_optimization: torch.Tensor   # [D,3]
        densify_size_reduction_factor: torch.Tensor                # [D,1]: log(phi) for a split, 0 for a clone
        densify_point_grad_position: torch.Tensor                  # [D,3]

    def __init__(self, config: "GaussianPointAdaptiveController.GaussianPointAdaptiveControllerConfig",
                 maintained_parameters: "GaussianPointAdaptiveController.GaussianPointAdaptiveControllerMaintainedParameters",
                 sample_from_point: Optional[Callable[[torch.Tensor, torch.Tensor], torch.Tensor]] = None,
                 ellipsoid_offset: Optional[Callable[[torch.Tensor], torch.Tensor]] = None):
        self.config = config
        self.maintained_parameters = maintained_parameters
        self.iteration_counter = -1
        self.mask_version = 0   # bumped whenever the controller changes which rows are live (_add_densify_points)
        self.input_data: Optional[HookInput] = None
        self.densify_point_info = None
        self.has_plot = False
        self.last_densify_uv: Optional[Dict[str, torch.Tensor]] = None
        # the two kernels are injectable so the decision logic can be exercised without a GPU
        self._sample_fn = sample_from_point or _hip_sample_from_point
        self._offset_fn = ellipsoid_offset or _hip_ellipsoid_offset
        self._clear_statistics()

    # ------------------------------------------------------------------ statistics (ADC:111-124,153-164)
    def _clear_statistics(self) -> None:
        xyz = self.maintained_parameters.pointcloud
        n, dev = xyz.shape[0], xyz.device
        self.accumulated_num_pixels = torch.zeros(n, dtype=torch.int32, device=dev)
        self.accumulated_num_in_camera = torch.zeros(n, dtype=torch.int32, device=dev)
        self.accumulated_view_space_position_gradients = torch.zeros(n, dtype=torch.float32, device=dev)
        self.accumulated_view_space_position_gradients_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.accumulated_position_gradients = torch.zeros(n, 3, dtype=torch.float32, device=dev)
        self.accumulated_position_gradients_norm = torch.zeros(n, dtype=torch.float32, device=dev)

    def next_update_selects(self) -> bool:
        """True if the next ``update`` call is a densification iteration, i.e. keeps its hook input (``input_data``) for
        the trainer's histograms -- lets the trainer ask the rasteriser for the costly compact feature gradients
        only then."""
        nxt = self.iteration_counter + 1
        return nxt >= self.config.num_iterations_warm_up and nxt % self.config.num_iterations_densify == 0

    def _is_densify_iteration(self) -> bool:
        return (self.iteration_counter >= self.config.num_iterations_warm_up and
                self.iteration_counter % self.config.num_iterations_densify == 0)

    @torch.no_grad()
    def update(self, input_data: HookInput) -> None:
        """Backward hook (ADC:130-146).  The visible ids are unique, so indexed += is a plain scatter."""
        self.iteration_counter += 1
        pixels = input_data.num_affected_pixels
        magnitude = input_data.magnitude_grad_viewspace
        grad_xyz = input_data.grad_point_in_camera
        if grad_xyz.is_cuda:   # one HIP pass instead of ~20 eager gather/scatter kernels per training iteration
            from . import hip_ops
            hip_ops.controller_accumulate(
                input_data.point_id_in_camera_list.to(torch.int32).contiguous(), pixels.to(torch.int32).contiguous(),
                magnitude.contiguous(), grad_xyz.contiguous(), self.accumulated_num_in_camera,
                self.accumulated_num_pixels, self.accumulated_view_space_position_gradients,
                self.accumulated_view_space_position_gradients_avg, self.accumulated_position_gradients,
                self.accumulated_position_gradients_norm)
        else:
            ids = input_data.point_id_in_camera_list.long()
            self.accumulated_num_in_camera[ids] += 1
            self.accumulated_num_pixels[ids] += pixels.to(torch.int32)
            self.accumulated_view_space_position_gradients[ids] += magnitude
            self.accumulated_view_space_position_gradients_avg[ids] += _ratio(magnitude, pixels)
            self.accumulated_position_gradients[ids] += grad_xyz
            self.accumulated_position_gradients_norm[ids] += grad_xyz.norm(dim=1)
        if self._is_densify_iteration():
            self._find_densify_points(input_data)
            self.input_data = input_data

    @torch.no_grad()
    def refinement(self) -> None:
        """After the optimiser step (ADC:148-167)."""
        if self.iteration_counter < self.config.num_iterations_warm_up:
            return
        if self.iteration_counter % self.config.num_iterations_densify == 0:
            self._add_densify_points()
            self._clear_statistics()
        if self.iteration_counter % self.config.num_iterations_reset_alpha == 0:
            self.reset_alpha()
        self.input_data = None

    # ------------------------------------------------------------------ selection (ADC:169-267)
    def _find_densify_points(self, input_data: HookInput) -> None:
        cfg, mp = self.config, self.maintained_parameters
        xyz, feat = mp.pointcloud, mp.pointcloud_features
        n, dev = xyz.shape[0], xyz.device
        live = mp.point_invalid_mask == 0
        ids_in_cam = input_data.point_id_in_camera_list.long()
        pixels = input_data.num_affected_pixels
        magnitude = input_data.magnitude_grad_viewspace
        frames = self.accumulated_num_in_camera
        mean_pixels = _ratio(self.accumulated_num_pixels, frames)

        # floaters: huge footprints close to the camera, judged on the current frame only (ADC:190-199)
        floater_in_cam = torch.zeros(ids_in_cam.shape[0], dtype=torch.bool, device=dev)
        floater = torch.zeros(n, dtype=torch.bool, device=dev)
        floater_id = torch.empty(0, dtype=torch.int32, device=dev)
        if self.iteration_counter > cfg.iteration_start_remove_floater:
            floater_in_cam = ((pixels > cfg.floater_near_camrea_num_pixels_threshold) &
                              (input_data.point_depth < cfg.floater_depth_threshold))
            floater_id = input_data.point_id_in_camera_list[floater_in_cam]
            floater[floater_id.long()] = True
            floater &= live

        # transparent (or NaN) Gaussians, over every live row (ADC:201-206)
        broken = torch.isnan(feat).any(dim=1)
        transparent = ((feat[:, 7] < cfg.transparent_alpha_threshold) | broken) & live & ~floater
        removed = floater | transparent
        removed_in_cam = floater_in_cam | transparent[ids_in_cam]

        # single-frame criteria on the visible Gaussians (ADC:209-226)
        chosen_in_cam = (magnitude > cfg.densification_view_space_position_gradients_threshold) & ~removed_in_cam
        n_by_sum = chosen_in_cam.sum()
        chosen_in_cam = (chosen_in_cam | (magnitude / pixels >
                                          cfg.densification_view_avg_space_position_gradients_threshold)) & ~removed_in_cam
        single_frame = torch.zeros(n, dtype=torch.bool, device=dev)
        single_frame[ids_in_cam[chosen_in_cam]] = True

        # multi-frame criteria on the accumulated statistics (ADC:228-238); thresholds default to "off" (1e3)
        multi_frame = _ratio(self.accumulated_view_space_position_gradients, frames) > \
            cfg.densification_multi_frame_view_space_position_gradients_threshold
        per_pixel = _ratio(self.accumulated_view_space_position_gradients_avg, frames) / mean_pixels
        multi_frame |= per_pixel > cfg.densification_multi_frame_view_pixel_avg_space_position_gradients_threshold
        multi_frame |= (self.accumulated_position_gradients_norm / frames) > \
            cfg.densification_multi_frame_position_gradients_threshold

        to_densify = (single_frame | multi_frame) & ~removed
        densify_id = torch.nonzero(to_densify).squeeze(1)
        _log.info("densify candidates: %d single-frame (%d by summed gradient), %d after merging multi-frame",
                  int(chosen_in_cam.sum()), int(n_by_sum), densify_id.shape[0])

        # a Gaussian that covered many pixels is split (shrunk by phi), a small one is cloned (ADC:250-255)
        shrink = torch.zeros(densify_id.shape[0], dtype=torch.float32, device=dev)
        shrink[self.accumulated_num_pixels[densify_id] > cfg.under_reconstructed_num_pixels_threshold] = \
            math.log(cfg.gaussian_split_factor_phi)
        mean_grad = _ratio(self.accumulated_position_gradients[densify_id], frames[densify_id].unsqueeze(-1))
        self.densify_point_info = self.GaussianPointAdaptiveControllerDensifyPointInfo(
            floater_point_id=floater_id,
            transparent_point_id=torch.nonzero(transparent).squeeze(1),
            densify_point_id=densify_id,
            densify_point_position_before_optimization=xyz[densify_id].detach().clone(),
            densify_size_reduction_factor=shrink.unsqueeze(-1),
            densify_point_grad_position=mean_grad)

        if self.iteration_counter % cfg.plot_densify_interval == 0:   # ADC:268-287, data only
            uv = input_data.point_uv_in_camera
            split_in_cam = self.accumulated_num_pixels[ids_in_cam[chosen_in_cam]] > \
                cfg.under_reconstructed_num_pixels_threshold
            self.last_densify_uv = {"floater": uv[floater_in_cam],
                                    "over_reconstructed": uv[chosen_in_cam][split_in_cam],
                                    "under_reconstructed": uv[chosen_in_cam][~split_in_cam]}

    # ------------------------------------------------------------------ apply (ADC:289-353)
    def _add_densify_points(self) -> None:
        info, cfg, mp = self.densify_point_info, self.config, self.maintained_parameters
        assert info is not None, "refinement() on a densification iteration needs the hook to have run"
        xyz, feat, invalid = mp.pointcloud, mp.pointcloud_features, mp.point_invalid_mask
        n_live_before = int((invalid == 0).sum())
        invalid[info.transparent_point_id.long()] = 1
        invalid[info.floater_point_id.long()] = 1
        n_removed = info.transparent_point_id.shape[0] + info.floater_point_id.shape[0]

        wanted = info.densify_point_id.shape[0]
        free_rows = torch.nonzero(invalid == 1).squeeze(1)[:wanted]   # lowest free rows first
        filled = free_rows.shape[0]
        if filled > 0:
            src = info.densify_point_id[:filled]
            shrink = info.densify_size_reduction_factor[:filled]
            # the new row starts from the pre-step position, so parent and child differ (ADC:303-306)
            xyz[free_rows] = info.densify_point_position_before_optimization[:filled]
            feat[free_rows] = feat[src]
            mp.point_object_id[free_rows] = mp.point_object_id[src]
            feat[free_rows, 4:7] -= shrink
            feat[src, 4:7] -= shrink
            is_split = (shrink > 1e-6).reshape(-1)
            if cfg.enable_ellipsoid_offset:   # parent and child to the two foci (ADC:323-328)
                offset = self._offset_fn(feat[src].detach())
                xyz[free_rows] += offset
                xyz[src] -= offset
            if cfg.enable_sample_from_point:  # ADC:329-345
                parents, children = src[is_split], free_rows[is_split]
                parent_xyz, parent_feat = xyz[parents].detach(), feat[parents].detach()
                child_pos = self._sample_fn(parent_xyz, parent_feat)
                parent_pos = self._sample_fn(parent_xyz, parent_feat)   # independent second draw
                xyz[children] = child_pos
                xyz[parents] = parent_pos
                clones = free_rows[~is_split]
                xyz[clones] += info.densify_point_grad_position[:filled][~is_split] * \
                    cfg.under_reconstructed_move_factor
            invalid[free_rows] = 0
            _log.info("densified %d of %d candidates (%d splits, %d clones)", filled, wanted,
                      int(is_split.sum()), filled - int(is_split.sum()))
        self.mask_version += 1   # the live set has changed (what depends on its size re-reads it)
        n_live_after = int((invalid == 0).sum())
        assert n_live_after == n_live_before - n_removed + filled
        _log.info("valid points %d -> %d (removed %d transparent, %d floaters)", n_live_before, n_live_after,
                  info.transparent_point_id.shape[0], info.floater_point_id.shape[0])
        self.densify_point_info = None

    @torch.no_grad()
    def reset_alpha(self) -> None:
        """Clamp every opacity logit from above (ADC:355-358)."""
        self.maintained_parameters.pointcloud_features[:, 7].clamp_(max=self.config.reset_alpha_value)

    # reference-named helpers (ADC:360-389)
    def _generate_point_offset(self, point_to_split: torch.Tensor, point_feature_to_split: torch.Tensor):
        return self._offset_fn(point_feature_to_split.detach())

    def _sample_from_point(self, point_to_split: torch.Tensor, point_feature_to_split: torch.Tensor):
        return self._sample_fn(point_to_split.detach(), point_feature_to_split.detach())
