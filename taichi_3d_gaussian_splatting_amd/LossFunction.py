"""Photometric loss of the trainer.

Mirror of the reference's ``taichi_3d_gaussian_splatting/LossFunction.py`` (LOS:8-54):
``L = (1 - lambda) * L1 + lambda * (1 - SSIM) [+ w * mean ||exp(s)||_2 over valid points]`` returning
``(L, L1, 1 - SSIM)``.  The reference takes SSIM from the third-party ``pytorch_msssim`` package (absent here);
``ssim`` below restates that package's published definition for the call the reference makes
(``ssim(X, Y, data_range=1, size_average=True)``): separable 11-tap Gaussian window (sigma 1.5), 'valid'
convolution per channel, K1 = 0.01, K2 = 0.03, mean over channels and images; a spatial dimension shorter than
the window is left unfiltered, as in that package.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F

from .yaml_config import YAMLConfig


def _gaussian_window(size: int, sigma: float, device, dtype) -> torch.Tensor:
    x = torch.arange(size, dtype=dtype, device=device) - size // 2
    g = torch.exp(-(x * x) / (2.0 * sigma * sigma))
    return g / g.sum()


def _blur(x: torch.Tensor, win: torch.Tensor) -> torch.Tensor:
    """Separable 'valid' Gaussian filter of a [B,C,H,W] tensor, one group per channel."""
    c = x.shape[1]
    k = win.numel()
    if x.shape[2] >= k:
        x = F.conv2d(x, win.view(1, 1, k, 1).expand(c, 1, k, 1), groups=c)
    if x.shape[3] >= k:
        x = F.conv2d(x, win.view(1, 1, 1, k).expand(c, 1, 1, k), groups=c)
    return x


def ssim(x: torch.Tensor, y: torch.Tensor, data_range: float = 1.0, size_average: bool = True,
         win_size: int = 11, win_sigma: float = 1.5, k1: float = 0.01, k2: float = 0.03) -> torch.Tensor:
    """Structural similarity of two [B,C,H,W] image batches (Wang et al. 2004, Gaussian-window form)."""
    if x.shape != y.shape or x.dim() != 4:
        raise ValueError("ssim expects two [B,C,H,W] tensors of the same shape")
    win = _gaussian_window(win_size, win_sigma, x.device, x.dtype)
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    mu_x, mu_y = _blur(x, win), _blur(y, win)
    mu_xx, mu_yy, mu_xy = mu_x * mu_x, mu_y * mu_y, mu_x * mu_y
    var_x = _blur(x * x, win) - mu_xx
    var_y = _blur(y * y, win) - mu_yy
    cov = _blur(x * y, win) - mu_xy
    contrast = (2.0 * cov + c2) / (var_x + var_y + c2)
    ssim_map = ((2.0 * mu_xy + c1) / (mu_xx + mu_yy + c1)) * contrast
    per_channel = ssim_map.flatten(2).mean(-1)     # [B,C]
    return per_channel.mean() if size_average else per_channel.mean(1)


class LossFunction(nn.Module):
    @dataclass
    class LossFunctionConfig(YAMLConfig):
        lambda_value: float = 0.2
        enable_regularization: bool = True
        regularization_weight: float = 2

    def __init__(self, config: "LossFunction.LossFunctionConfig"):
        super().__init__()
        self.config = config

    def forward(self, predicted_image, ground_truth_image, point_invalid_mask=None, pointcloud_features=None):
        """Images are [C,H,W] or [B,C,H,W] in 0..1."""
        pred = predicted_image if predicted_image.dim() == 4 else predicted_image.unsqueeze(0)
        target = ground_truth_image if ground_truth_image.dim() == 4 else ground_truth_image.unsqueeze(0)
        l1 = (pred - target).abs().mean()
        d_ssim = 1.0 - ssim(pred, target, data_range=1.0, size_average=True)
        lam = self.config.lambda_value
        total = (1.0 - lam) * l1 + lam * d_ssim
        if pointcloud_features is not None and self.config.enable_regularization:
            total = total + self.config.regularization_weight * self._regularization_loss(
                point_invalid_mask, pointcloud_features)
        return total, l1, d_ssim

    @staticmethod
    def _regularization_loss(point_invalid_mask, pointcloud_features):
        """Mean Euclidean norm of the three axis lengths exp(s) of the valid Gaussians (LOS:42-54)."""
        log_scale = pointcloud_features[point_invalid_mask == 0, 4:7]
        return torch.exp(log_scale).norm(dim=1).mean()
