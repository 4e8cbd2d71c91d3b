#!/usr/bin/env python
"""HIP-event timing of the fused loss kernels (forward, backward) at a given image size.  Development tool.
usage: python tools/loss_bench.py [H W] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_3d_gaussian_splatting_amd import hip_ops  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 2 else 1072
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device("cuda:0")
gt = torch.rand(3, H, W, device=dev)
pred = (gt.permute(1, 2, 0) + 0.1 * torch.randn(H, W, 3, device=dev)).contiguous().permute(2, 0, 1)
one = torch.ones((), device=dev)
losses, maps = hip_ops.loss_forward(pred, gt, 0.2, True, True)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for i in range(reps + 5):
    ev[0].record()
    losses, maps = hip_ops.loss_forward(pred, gt, 0.2, True, True)
    ev[1].record()
    g = hip_ops.loss_backward(pred, gt, maps, 0.2, True, one, None, None)
    ev[2].record()
    torch.cuda.synchronize()
    if i >= 5:
        tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
px = 3 * H * W
print(f"{H}x{W}: forward {1e3 * tf / reps:.1f} us ({20 * px / (tf / reps * 1e-3) / 1e9:.0f} GB/s algorithmic), "
      f"backward {1e3 * tb / reps:.1f} us ({24 * px / (tb / reps * 1e-3) / 1e9:.0f} GB/s algorithmic)")
