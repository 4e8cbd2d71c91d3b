#!/usr/bin/env python
"""Times gs_sort_pairs on random 32-bit (bin | depth) keys at the sizes the frames of BASELINE.json produce
(development tool; run through gpurun).  One process per library configuration (the knobs are read once):
    GS_SORT_IMPL=lsd3       the three-launches-per-pass sort of rounds 1-3
    GS_SWEEP_TICKET=0/1     tiles by blockIdx / by ticket
    GS_SWEEP_ROUNDS=R       keys per tile = 1024 R
usage: python tools/sort_bench.py [reps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_3d_gaussian_splatting_amd import hip_ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tag = " ".join(f"{k}={os.environ[k]}" for k in ("GS_SORT_IMPL", "GS_SWEEP_TICKET", "GS_SWEEP_ROUNDS") if k in os.environ)
ws = hip_ops.Workspaces()
for n, depth_bits, tile_bits in ((48_000, 9, 8), (360_000, 11, 11), (1_130_000, 9, 12), (2_877_171, 11, 11),
                                 (4_400_000, 11, 11), (9_500_000, 9, 13)):
    rng = np.random.default_rng(n)
    keys = (rng.integers(0, 1 << depth_bits, size=n) + (rng.integers(0, 1 << tile_bits, size=n) << depth_bits)).astype(np.uint32)
    k0 = torch.from_numpy(keys.view(np.int32)).cuda()
    p0 = torch.arange(n, dtype=torch.int32, device="cuda")
    times = []
    ok = None
    for r in range(reps + 3):
        k, p = k0.clone(), p0.clone()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        k, p = hip_ops.sort_pairs(k, p, depth_bits, tile_bits, depth_bits, in_place=False, ws=ws)
        b.record()
        torch.cuda.synchronize()
        if r >= 3:
            times.append(a.elapsed_time(b) * 1e3)
        if ok is None:
            order = np.argsort(keys, kind="stable")
            ok = bool(np.array_equal(k.cpu().numpy().view(np.uint32), keys[order]) and
                      np.array_equal(p.cpu().numpy(), order.astype(np.int32)))
    times.sort()
    print(f"[sort_bench] {tag or 'default'} n={n} bits={depth_bits + tile_bits} median={times[len(times) // 2]:.1f}us "
          f"min={times[0]:.1f}us correct={ok}", flush=True)
