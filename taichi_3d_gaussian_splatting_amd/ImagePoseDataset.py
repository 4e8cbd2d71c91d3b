"""Image + pose dataset feeding the rasteriser.

Mirror of the reference's ``taichi_3d_gaussian_splatting/ImagePoseDataset.py`` (DST): a JSON list of records
``{image_path, T_pointcloud_camera 4x4 (camera -> pointcloud), camera_intrinsics 3x3, camera_height,
camera_width, camera_id}`` (docs/RawDataFormat.md:17-60) -> ``(image f32[3,H,W] in 0..1, q[1,4] (x,y,z,w),
t[1,3], CameraInfo)`` with the reference's resolution rules, which define the sizes the operator is called with:
  * intrinsics rescaled to the real image size (DST:77-81),
  * width/height cropped to multiples of the 16-px tile (DST:82-86; the operator asserts it, RAS:1193-1194),
  * images larger than 1600 px in either dimension resized like torchvision ``resize(size=1024, max_size=1600)``
    (shorter edge -> 1024 unless that pushes the longer edge over 1600) with intrinsics scaled accordingly and
    cropped to /16 again (DST:41-62).
Image decoding uses PIL + numpy (no torchvision).  ``load_images=False`` skips the pixels (pose-only use, e.g.
rendering), then the sizes come from the JSON.
"""
from __future__ import annotations

import json
from typing import List

import numpy as np
import torch
import torch.utils.data

from .Camera import CameraInfo
from .utils import SE3_to_quaternion_and_translation_torch

TILE_WIDTH = 16
TILE_HEIGHT = 16
MAX_RESOLUTION_TRAIN = 1600
REQUIRED_KEYS = ("image_path", "T_pointcloud_camera", "camera_intrinsics", "camera_height", "camera_width",
                 "camera_id")


def _resized_hw(h: int, w: int, size: int = 1024, max_size: int = 1600):
    """Output size of torchvision.transforms.functional.resize(img, size=size, max_size=max_size)."""
    short, long_ = (h, w) if h <= w else (w, h)
    new_short, new_long = size, int(size * long_ / short)
    if new_long > max_size:
        new_short, new_long = int(max_size * new_short / new_long), max_size
    return (new_short, new_long) if h <= w else (new_long, new_short)


class ImagePoseDataset(torch.utils.data.Dataset):
    def __init__(self, dataset_json_path: str, load_images: bool = True):
        super().__init__()
        with open(dataset_json_path) as fh:
            self.records: List[dict] = json.load(fh)
        for rec in self.records:
            for key in REQUIRED_KEYS:
                if key not in rec:
                    raise KeyError(f"column {key} is not in the dataset")
        self.load_images = load_images

    def __len__(self) -> int:
        return len(self.records)

    @staticmethod
    def _autoscale_image_and_camera_info(image, camera_info: CameraInfo):
        h, w = camera_info.camera_height, camera_info.camera_width
        if h <= MAX_RESOLUTION_TRAIN and w <= MAX_RESOLUTION_TRAIN:
            return image, camera_info
        new_h, new_w = _resized_hw(h, w)
        if image is not None:
            image = torch.nn.functional.interpolate(image[None], size=(new_h, new_w), mode="bilinear",
                                                    antialias=True, align_corners=False)[0]
        scale_x, scale_y = new_w / w, new_h / h
        new_w -= new_w % TILE_WIDTH
        new_h -= new_h % TILE_HEIGHT
        if image is not None:
            image = image[:3, :new_h, :new_w].contiguous()
        K = camera_info.camera_intrinsics.clone()
        K[0, 0] *= scale_x; K[0, 2] *= scale_x
        K[1, 1] *= scale_y; K[1, 2] *= scale_y
        return image, CameraInfo(camera_intrinsics=K, camera_height=new_h, camera_width=new_w,
                                 camera_id=camera_info.camera_id)

    def __getitem__(self, idx: int):
        rec = self.records[idx]
        T = torch.tensor(rec["T_pointcloud_camera"], dtype=torch.float32)
        q, t = SE3_to_quaternion_and_translation_torch(T.unsqueeze(0))
        K = torch.tensor(rec["camera_intrinsics"], dtype=torch.float32)
        base_h, base_w = int(rec["camera_height"]), int(rec["camera_width"])
        image = None
        h, w = base_h, base_w
        if self.load_images:
            import PIL.Image
            arr = np.asarray(PIL.Image.open(rec["image_path"]).convert("RGB"), dtype=np.float32) / 255.0
            image = torch.from_numpy(arr).permute(2, 0, 1).contiguous()
            h, w = image.shape[1], image.shape[2]  # the real image size wins over the JSON (DST:74-81)
        K[0, :] = K[0, :] * w / base_w
        K[1, :] = K[1, :] * h / base_h
        w -= w % TILE_WIDTH
        h -= h % TILE_HEIGHT
        if image is not None:
            image = image[:3, :h, :w].contiguous()
        info = CameraInfo(camera_intrinsics=K, camera_height=h, camera_width=w, camera_id=rec["camera_id"])
        image, info = self._autoscale_image_and_camera_info(image, info)
        return image, q, t, info
