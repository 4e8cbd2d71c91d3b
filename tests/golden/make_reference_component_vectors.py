#!/usr/bin/env python
"""Generates tests/golden/reference_components.npz by EXECUTING the reference's adaptive controller and scene
initialisation (unmodified sources under /root/reference, Taichi emulated by tests/golden/taichi_emulation.py,
``plyfile`` / ``dataclass_wizard`` stubbed: they are import-time dependencies only) on seeded inputs.

Run in the build container only:  python tests/golden/make_reference_component_vectors.py

* controller (ADC:44-358): three hook calls + refinements on a 60-row cloud with 40 live rows; statistics, the
  selection (floaters / transparent / clone / split), the refill of free rows, the ellipsoid-foci offsets (the Taichi
  kernel ADC:10-25 runs through the emulation) and the opacity reset.  Random sampling (ADC:27-42, ti.random) is off.
* scene (SCN:74-130, 183-211): ``from_parquet`` of a raw x,y,z,r,g,b cloud -> KD-tree scales, random unit quaternions
  (torch RNG, seeded), opacity, SH DC from the colours; with and without spare capacity.
"""
import os
import sys
import types

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import matplotlib  # noqa: E402
matplotlib.use("Agg")
import taichi_emulation as E  # noqa: E402


def hook_inputs(g, n_rows, live, step):
    """A synthetic BackwardValidPointHookInput (numpy dict): a random subset of the live rows is 'in camera'."""
    ids = np.sort(g.choice(np.nonzero(live)[0], size=max(4, int(0.7 * live.sum())), replace=False)).astype(np.int32)
    m = len(ids)
    pixels = g.integers(0, 900, m).astype(np.int32)
    pixels[g.random(m) < 0.1] = 0                                    # never touched: 0/0 in the averages
    return dict(point_id_in_camera_list=ids,
                grad_point_in_camera=(g.normal(size=(m, 3)) * 1e-3).astype(np.float32),
                grad_pointfeatures_in_camera=(g.normal(size=(m, 56)) * 1e-3).astype(np.float32),
                grad_viewspace=(g.normal(size=(m, 2)) * 1e-5).astype(np.float32),
                magnitude_grad_viewspace=np.abs(g.normal(size=m) * 8e-6).astype(np.float32),
                magnitude_grad_viewspace_on_image=np.zeros((16, 16, 2), np.float32),
                num_overlap_tiles=g.integers(1, 9, m).astype(np.int32), num_affected_pixels=pixels,
                point_depth=(g.random(m) * 6).astype(np.float32), point_uv_in_camera=(g.random((m, 2)) * 16).astype(np.float32))


def main():
    plyfile = types.ModuleType("plyfile")
    plyfile.PlyData = plyfile.PlyElement = object
    sys.modules["plyfile"] = plyfile
    mods = E.load_reference("/root/reference", modules=("Camera", "utils", "SphericalHarmonics", "GaussianPoint3D",
                                                         "GaussianPointCloudRasterisation",
                                                         "GaussianPointAdaptiveController", "GaussianPointCloudScene"))
    ADC = mods["GaussianPointAdaptiveController"].GaussianPointAdaptiveController
    RAS = mods["GaussianPointCloudRasterisation"].GaussianPointCloudRasterisation
    Scene = mods["GaussianPointCloudScene"].GaussianPointCloudScene
    out = {}

    # ------------------------------------------------------------------ controller
    g = np.random.default_rng(77)
    n, n_live = 60, 40
    xyz = torch.tensor(g.normal(size=(n, 3)), dtype=torch.float32)
    feat = torch.tensor(g.normal(size=(n, 56)) * 0.5, dtype=torch.float32)
    feat[:, 0:4] = feat[:, 0:4] / feat[:, 0:4].norm(dim=1, keepdim=True)
    feat[:, 7] = torch.tensor(g.uniform(-1.5, 2.0, n), dtype=torch.float32)
    feat[3, 20] = float("nan")                                                     # a broken row is removed
    invalid = torch.zeros(n, dtype=torch.int8); invalid[n_live:] = 1
    obj = torch.tensor(g.integers(0, 3, n), dtype=torch.int32)
    cfg = dict(num_iterations_warm_up=0, num_iterations_densify=1, num_iterations_reset_alpha=2, reset_alpha_value=0.3,
               iteration_start_remove_floater=-1, floater_near_camrea_num_pixels_threshold=700, floater_depth_threshold=2.0,
               under_reconstructed_num_pixels_threshold=300, enable_ellipsoid_offset=True, enable_sample_from_point=False,
               plot_densify_interval=10 ** 9, densification_view_space_position_gradients_threshold=6e-6,
               transparent_alpha_threshold=-0.5, under_reconstructed_move_factor=100.0)
    out["adc_config"] = np.array(repr(cfg))
    out.update(adc_xyz0=xyz.numpy().copy(), adc_feat0=feat.numpy().copy(), adc_invalid0=invalid.numpy().copy(),
               adc_obj0=obj.numpy().copy())
    ctrl = ADC(ADC.GaussianPointAdaptiveControllerConfig(**cfg),
               ADC.GaussianPointAdaptiveControllerMaintainedParameters(pointcloud=xyz, pointcloud_features=feat,
                                                                       point_invalid_mask=invalid, point_object_id=obj))
    for step in range(3):
        h = hook_inputs(g, n, invalid.numpy() == 0, step)
        for k, v in h.items():
            out[f"adc_hook{step}_{k}"] = v
        ctrl.update(RAS.BackwardValidPointHookInput(**{k: torch.from_numpy(v) for k, v in h.items()}))
        info = ctrl.densify_point_info
        out[f"adc_step{step}_floater_id"] = info.floater_point_id.numpy().copy()
        out[f"adc_step{step}_transparent_id"] = info.transparent_point_id.numpy().copy()
        out[f"adc_step{step}_densify_id"] = info.densify_point_id.numpy().copy()
        out[f"adc_step{step}_shrink"] = info.densify_size_reduction_factor.numpy().copy()
        out[f"adc_step{step}_grad_position"] = info.densify_point_grad_position.numpy().copy()
        with torch.no_grad():
            xyz += 0.01 * (step + 1)                                   # the optimiser step between hook and refinement
        ctrl.refinement()
        out[f"adc_step{step}_xyz"] = xyz.numpy().copy()
        out[f"adc_step{step}_feat"] = feat.numpy().copy()
        out[f"adc_step{step}_invalid"] = invalid.numpy().copy()
        out[f"adc_step{step}_obj"] = obj.numpy().copy()
        print("controller step", step, "live", int((invalid == 0).sum()), "densify", len(info.densify_point_id),
              "transparent", len(info.transparent_point_id), "floaters", len(info.floater_point_id))

    # ------------------------------------------------------------------ scene initialisation
    pts = g.random((150, 3)).astype(np.float32)
    rgb = g.integers(0, 256, (150, 3)).astype(np.float64)
    out.update(scn_points=pts, scn_rgb=rgb)
    path = "/tmp/_ref_raw_cloud.parquet"
    pd.DataFrame(np.concatenate([pts, rgb], 1), columns=["x", "y", "z", "r", "g", "b"]).to_parquet(path)
    for tag, kw in (("plain", dict(initial_alpha=-1.5, initial_covariance_ratio=0.7)),
                    ("capacity", dict(max_num_points_ratio=2.0, initial_alpha=0.05, max_initial_covariance=0.05))):
        torch.manual_seed(123)
        scene = Scene.from_parquet(path, config=Scene.PointCloudSceneConfig(**kw))
        out[f"scn_{tag}_config"] = np.array(repr(kw))
        out[f"scn_{tag}_xyz"] = scene.point_cloud.detach().numpy().copy()
        out[f"scn_{tag}_feat"] = scene.point_cloud_features.detach().numpy().copy()
        out[f"scn_{tag}_invalid"] = scene.point_invalid_mask.numpy().copy()
        print("scene", tag, scene.point_cloud.shape, int(scene.point_invalid_mask.sum()))
    np.savez_compressed(os.path.join(HERE, "reference_components.npz"), **out)


if __name__ == "__main__":
    main()
