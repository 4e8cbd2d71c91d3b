"""MI355X-native differentiable 3D-Gaussian-splatting rasteriser.

Drop-in for the hot path of wanmeihuali/taichi_3d_gaussian_splatting: the operator
``GaussianPointCloudRasterisation`` (same dataclasses, outputs, gradients and backward hook as the
reference) backed by hand-written HIP kernels for gfx950 behind a C ABI (include/gsplat_hip.h).
"""
from .Camera import CameraInfo, CameraView  # noqa: F401
from .GaussianPointCloudRasterisation import (  # noqa: F401
    BOUNDARY_TILES, TILE_HEIGHT, TILE_WIDTH, GaussianPointCloudRasterisation, find_tile_start_and_end)
