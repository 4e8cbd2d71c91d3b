"""Photometric loss of the trainer.

Mirror of the reference's ``taichi_3d_gaussian_splatting/LossFunction.py`` (LOS:8-54):
``L = (1 - lambda) * L1 + lambda * (1 - SSIM) [+ w * mean ||exp(s)||_2 over valid points]`` returning
``(L, L1, 1 - SSIM)``.  The reference takes SSIM from the third-party ``pytorch_msssim`` package (absent here);
``ssim`` below restates that package's published definition for the call the reference makes
(``ssim(X, Y, data_range=1, size_average=True)``): separable 11-tap Gaussian window (sigma 1.5), 'valid'
convolution per channel, K1 = 0.01, K2 = 0.03, mean over channels and images; a spatial dimension shorter than
the window is left unfiltered, as in that package.

Two implementations of the same function:
* device tensors, one 3-channel image (the trainer's case): ``fused_l1_ssim`` -- the HIP kernels of
  csrc/gs_loss.hip (forward + hand-derived backward, optionally with the ``clamp(0,1)`` of TRN:168 folded in).
  Eager PyTorch needs ~6.5 ms per 1920x1072 iteration for this chain, 5x the rasteriser; the fused pair is
  HBM-bound at ~0.1 ms.  It raises if the HIP library is missing.
* anything else (CPU tensors, batches, other channel counts): the plain PyTorch formulation below, which is also
  the fp32 reference the kernel tests compare against.
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
    if (_fusable(x, y) and not (x.requires_grad or y.requires_grad) and data_range == 1.0 and win_size == 11 and
            win_sigma == 1.5 and (k1, k2) == (0.01, 0.03) and size_average):
        # metric use on the device (PSNR/SSIM logging, validation): the fused forward kernel -- MIOpen needs ~90 ms per
        # depth-wise 11-tap convolution at 1920x1072, i.e. seconds per SSIM evaluation
        from . import hip_ops
        losses, _ = hip_ops.loss_forward(x[0], y[0].contiguous(), 0.0, False, need_grad=False)
        return 1.0 - losses[2]
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


class _FusedL1SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prediction, target, lambda_value, clamp):
        from . import hip_ops
        need_grad = prediction.requires_grad
        losses, maps = hip_ops.loss_forward(prediction.detach(), target.detach(), lambda_value, clamp, need_grad)
        ctx.lambda_value, ctx.clamp = lambda_value, clamp
        ctx.save_for_backward(prediction.detach(), target.detach(), maps)
        ctx.set_materialize_grads(False)
        return losses[0], losses[1], losses[2]

    @staticmethod
    def backward(ctx, grad_total, grad_l1, grad_dssim):
        from . import hip_ops
        prediction, target, maps = ctx.saved_tensors
        if maps is None or (grad_total is None and grad_l1 is None and grad_dssim is None):
            return None, None, None, None
        grad = hip_ops.loss_backward(prediction, target, maps, ctx.lambda_value, ctx.clamp, grad_total, grad_l1,
                                     grad_dssim)
        return grad, None, None, None


def fused_l1_ssim(prediction: torch.Tensor, target: torch.Tensor, lambda_value: float = 0.2,
                  clamp_prediction: bool = False):
    """(L, L1, 1 - SSIM) of one [3,H,W] image pair on the HIP device, differentiable w.r.t. ``prediction``.
    ``prediction`` may be the ``permute(2,0,1)`` view of the rasteriser's [H,W,3] output (no copy is made);
    ``clamp_prediction`` applies ``clamp(0,1)`` to it inside the kernel."""
    return _FusedL1SSIM.apply(prediction, target, float(lambda_value), bool(clamp_prediction))


def _fusable(pred: torch.Tensor, target: torch.Tensor) -> bool:
    return (pred.is_cuda and target.is_cuda and pred.dtype == torch.float32 and target.dtype == torch.float32 and
            pred.dim() == 4 and pred.shape[0] == 1 and pred.shape[1] == 3 and pred.shape == target.shape and
            pred.shape[2] >= 11 and pred.shape[3] >= 11)


class LossFunction(nn.Module):
    @dataclass
    class LossFunctionConfig(YAMLConfig):
        lambda_value: float = 0.2
        enable_regularization: bool = True
        regularization_weight: float = 2

    def __init__(self, config: "LossFunction.LossFunctionConfig"):
        super().__init__()
        self.config = config

    def forward(self, predicted_image, ground_truth_image, point_invalid_mask=None, pointcloud_features=None,
                clamp_prediction: bool = False):
        """Images are [C,H,W] or [B,C,H,W] in 0..1.  ``clamp_prediction`` (an addition) folds the trainer's
        ``clamp(prediction, 0, 1)`` into the loss."""
        pred = predicted_image if predicted_image.dim() == 4 else predicted_image.unsqueeze(0)
        target = ground_truth_image if ground_truth_image.dim() == 4 else ground_truth_image.unsqueeze(0)
        lam = self.config.lambda_value
        if _fusable(pred, target):
            total, l1, d_ssim = fused_l1_ssim(pred[0], target[0].contiguous(), lam, clamp_prediction)
        else:
            if clamp_prediction:
                pred = pred.clamp(0.0, 1.0)
            l1 = (pred - target).abs().mean()
            d_ssim = 1.0 - ssim(pred, target, data_range=1.0, size_average=True)
            total = (1.0 - lam) * l1 + lam * d_ssim
        if pointcloud_features is not None and self.config.enable_regularization:
            total = total + self.config.regularization_weight * self._regularization_loss(
                point_invalid_mask, pointcloud_features)
        return total, l1, d_ssim

    @torch.no_grad()
    def add_regularization_gradient_(self, point_invalid_mask: torch.Tensor, pointcloud_features: torch.Tensor):
        """Trainer fast path (an addition): returns ``regularization_weight * R`` and adds its gradient to
        ``pointcloud_features.grad[:, 4:7]`` in place, instead of sending R through autograd -- which would
        materialise a dense [N,56] gradient and add it to the rasteriser's.  Call it after ``backward()`` of the
        image loss; the sum of both equals the loss / gradient of ``forward(..., pointcloud_features=...)``.
        Returns None when the regulariser is disabled."""
        if not self.config.enable_regularization:
            return None
        weight = float(self.config.regularization_weight)
        if pointcloud_features.grad is None:
            pointcloud_features.grad = torch.zeros_like(pointcloud_features)
        if pointcloud_features.is_cuda:
            from . import hip_ops
            value_and_count = hip_ops.scale_regulariser(pointcloud_features, point_invalid_mask, weight,
                                                        grad_features=pointcloud_features.grad)
            return weight * value_and_count[0]
        with torch.enable_grad():
            leaf = pointcloud_features.detach().requires_grad_(True)
            value = self._regularization_loss(point_invalid_mask, leaf)
            (grad,) = torch.autograd.grad(value, leaf)
        pointcloud_features.grad.add_(grad, alpha=weight)
        return weight * value.detach()

    @torch.no_grad()
    def regularization_value(self, point_invalid_mask: torch.Tensor, pointcloud_features: torch.Tensor) -> torch.Tensor:
        """``regularization_weight * R`` without any gradient (for logging when the gradient is applied elsewhere, e.g.
        inside ``optim.Adam.set_scale_regulariser``)."""
        weight = float(self.config.regularization_weight)
        if pointcloud_features.is_cuda:
            from . import hip_ops
            return weight * hip_ops.scale_regulariser(pointcloud_features, point_invalid_mask)[0]
        return weight * self._regularization_loss(point_invalid_mask, pointcloud_features.detach())

    @staticmethod
    def _regularization_loss(point_invalid_mask, pointcloud_features):
        """Mean Euclidean norm of the three axis lengths exp(s) of the valid Gaussians (LOS:42-54), written as a
        masked mean: boolean indexing would cost a device->host synchronisation per iteration."""
        live = point_invalid_mask == 0
        # rows the controller invalidated may hold NaN (that is why they were invalidated): select, do not multiply
        scale = torch.where(live[:, None], pointcloud_features[:, 4:7], torch.zeros_like(pointcloud_features[:, 4:7]))
        axis_norm = torch.where(live, torch.exp(scale).norm(dim=1), torch.zeros_like(scale[:, 0]))
        return axis_norm.sum() / live.sum()
