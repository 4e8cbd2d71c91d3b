#!/usr/bin/env python
"""Adam step over the trainer's fixed-capacity tensors: every row (torch.optim.Adam's behaviour, optim.Adam without a row
mask) against skipping the rows of invalid points (optim.Adam.set_row_mask).  usage: python tools/adam_bench.py [rows] [live fraction]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taichi_3d_gaussian_splatting_amd.optim import Adam  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
live_fraction = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
dev = torch.device("cuda:0")
invalid = (torch.rand(n, device=dev) >= live_fraction).to(torch.int8)
for masked in (False, True):
    feat = torch.nn.Parameter(torch.randn(n, 56, device=dev))
    xyz = torch.nn.Parameter(torch.randn(n, 3, device=dev))
    opt = Adam([feat, xyz], lr=1e-3)
    opt.set_scale_regulariser(feat, 0.01, invalid)
    if masked:
        opt.set_row_mask(feat, invalid)
        opt.set_row_mask(xyz, invalid)
    feat.grad = torch.randn_like(feat) * (invalid == 0)[:, None]
    xyz.grad = torch.randn_like(xyz) * (invalid == 0)[:, None]
    for _ in range(5):
        opt.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        opt.step()
    torch.cuda.synchronize()
    print(f"{n} rows, {live_fraction:.0%} live, {'rows of invalid points skipped' if masked else 'every row stepped':30s}: "
          f"{(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per step (features + positions)")
