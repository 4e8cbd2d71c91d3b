#!/usr/bin/env python
"""Offline renderer on the MI355X rasteriser: scene parquet + camera poses -> PNG frames.

Same command line as the reference's ``gaussian_point_render.py`` (RENDER:123-176):
    python gaussian_point_render.py --parquet_path scene.parquet --poses poses.{pt|json} --output_prefix out/
``--poses`` is either a ``torch.save``d tensor [F,4,4] of camera->pointcloud matrices (rendered at 976x544, or
544x976 with ``--portrait_mode``; RENDER:23-38) or a dataset JSON (docs/RawDataFormat.md), whose image size and
intrinsics are then used (RENDER:148-169).  Several parquet files separated by ',' are merged into one scene with
one object id per file (RENDER:68-98).  No Taichi: the operator is the HIP one.
"""
from __future__ import annotations

import argparse
import os
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np
import torch

from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation
from taichi_3d_gaussian_splatting_amd.GaussianPointCloudScene import GaussianPointCloudScene
from taichi_3d_gaussian_splatting_amd.ImagePoseDataset import ImagePoseDataset
from taichi_3d_gaussian_splatting_amd.utils import (SE3_to_quaternion_and_translation_torch,
                                                    quaternion_to_rotation_matrix_torch)


class GaussianPointRenderer:
    @dataclass
    class GaussianPointRendererConfig:
        parquet_path: str
        cameras: torch.Tensor  # [F,4,4] camera -> pointcloud
        device: str = "cuda"
        image_height: int = 544
        image_width: int = 976
        camera_intrinsics: torch.Tensor = field(default_factory=lambda: torch.tensor(
            [[581.743, 0.0, 488.0], [0.0, 581.743, 272.0], [0.0, 0.0, 1.0]]))

        def set_portrait_mode(self):
            self.image_height, self.image_width = 976, 544
            self.camera_intrinsics = torch.tensor([[1163.486, 0.0, 272.0], [0.0, 1163.486, 488.0], [0.0, 0.0, 1.0]])

    def __init__(self, config: "GaussianPointRenderer.GaussianPointRendererConfig"):
        self.config = config
        config.image_height -= config.image_height % 16
        config.image_width -= config.image_width % 16
        scenes = [GaussianPointCloudScene.from_parquet(p, GaussianPointCloudScene.PointCloudSceneConfig())
                  for p in config.parquet_path.split(",")]
        self.scene = self._merge_scenes(scenes).to(config.device)
        self.cameras = config.cameras.to(config.device)
        self.camera_info = CameraInfo(camera_intrinsics=config.camera_intrinsics.to(config.device),
                                      camera_width=config.image_width, camera_height=config.image_height, camera_id=0)
        self.rasteriser = GaussianPointCloudRasterisation(
            GaussianPointCloudRasterisation.GaussianPointCloudRasterisationConfig(
                near_plane=0.8, far_plane=1000., depth_to_sort_key_scale=100.))

    @staticmethod
    def _merge_scenes(scenes):
        xyz = torch.cat([s.point_cloud.detach() for s in scenes])
        feat = torch.cat([s.point_cloud_features.detach() for s in scenes])
        obj = torch.cat([torch.full((s.point_cloud.shape[0],), i, dtype=torch.int32) for i, s in enumerate(scenes)])
        return GaussianPointCloudScene(xyz, GaussianPointCloudScene.PointCloudSceneConfig(), point_cloud_features=feat,
                                       point_object_id=obj)

    @torch.no_grad()
    def render(self, index: int) -> torch.Tensor:
        pose = self.cameras[index].unsqueeze(0)
        q, t = SE3_to_quaternion_and_translation_torch(pose)
        n_obj = int(self.scene.point_object_id.max().item()) + 1 if self.scene.point_object_id.numel() else 1
        image, _, _ = self.rasteriser(GaussianPointCloudRasterisation.GaussianPointCloudRasterisationInput(
            point_cloud=self.scene.point_cloud, point_cloud_features=self.scene.point_cloud_features,
            point_invalid_mask=self.scene.point_invalid_mask, point_object_id=self.scene.point_object_id,
            camera_info=self.camera_info, q_pointcloud_camera=q.expand(n_obj, 4).contiguous(),
            t_pointcloud_camera=t.expand(n_obj, 3).contiguous(), color_max_sh_band=3))
        return image

    def run(self, output_prefix: Path) -> None:
        from PIL import Image
        for i in range(self.cameras.shape[0]):
            image = self.render(i)
            Image.fromarray(torch.clamp(image * 255, 0, 255).byte().cpu().numpy(), "RGB").save(
                output_prefix / f"frame_{i:03}.png")


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--parquet_path", type=str, required=True)
    ap.add_argument("--poses", type=str, required=True,
                    help="a .pt file saved with torch.save() ([F,4,4]) or a dataset json (docs/RawDataFormat.md)")
    ap.add_argument("--output_prefix", type=str, required=True)
    ap.add_argument("--gt_prefix", type=str, default="")
    ap.add_argument("--portrait_mode", action="store_true", default=False)
    args = ap.parse_args()
    out = Path(args.output_prefix)
    os.makedirs(out, exist_ok=True)
    if torch.cuda.is_available():   # the launching threads on one L3 complex next to the GPU (host_affinity.py)
        from taichi_3d_gaussian_splatting_amd import host_affinity
        host_affinity.pin_host_threads(torch.cuda.current_device())
    if args.poses.endswith(".pt"):
        config = GaussianPointRenderer.GaussianPointRendererConfig(args.parquet_path, torch.load(args.poses))
        if args.portrait_mode:
            config.set_portrait_mode()
    elif args.poses.endswith(".json"):
        ds = ImagePoseDataset(args.poses, load_images=bool(args.gt_prefix))
        cameras = torch.zeros(len(ds), 4, 4)
        info = None
        for i in range(len(ds)):
            image_gt, q, t, info = ds[i]
            cameras[i, :3, :3] = quaternion_to_rotation_matrix_torch(q)[0]
            cameras[i, :3, 3] = t[0]
            cameras[i, 3, 3] = 1.0
            if args.gt_prefix:
                from PIL import Image
                os.makedirs(args.gt_prefix, exist_ok=True)
                arr = (image_gt.permute(1, 2, 0).clamp(0, 1) * 255).byte().numpy()
                Image.fromarray(arr, "RGB").save(Path(args.gt_prefix) / f"frame_{i:03}.png")
        config = GaussianPointRenderer.GaussianPointRendererConfig(args.parquet_path, cameras)
        config.image_width, config.image_height = info.camera_width, info.camera_height
        config.camera_intrinsics = info.camera_intrinsics
    else:
        raise ValueError(f"Unrecognized poses file format: {args.poses}, Must be .pt or .json file")
    GaussianPointRenderer(config).run(out)


if __name__ == "__main__":
    main()
