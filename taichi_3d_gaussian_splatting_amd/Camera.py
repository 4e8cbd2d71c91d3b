"""Camera types of the operator boundary.

Same names and fields as the reference's ``taichi_3d_gaussian_splatting/Camera.py:6-22``
(``CameraInfo`` is part of ``GaussianPointCloudRasterisationInput``, RAS:799).
"""
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class CameraInfo:
    camera_intrinsics: torch.Tensor  # 3x3 matrix, on the compute device
    camera_height: int  # height of the image
    camera_width: int  # width of the image
    camera_id: int  # camera id


@dataclass
class CameraView:
    camera_view_id: int
    # 4x4 SE(3) matrix, transforms points from the camera frame to the pointcloud frame
    T_pointcloud_camera: torch.Tensor
    camera_id: int
    image_id: int
    timestamp: Optional[int] = None
