#!/usr/bin/env python
"""Times gs_sort_pairs on random 32-bit (bin | depth) keys at the sizes the frames of BASELINE.json produce
(development tool; run through gpurun).  One process per library configuration (the knobs are read once):
    GS_SORT_IMPL=lsd        LSD passes (three launches per eight bits) at every size
    GS_SORT_MSD_BITS=8|9    width of the MSD-first sort's partitioning digit
    GS_SORT_GROUPED=1       the frame's contract (bins in any order: partition by the lowest bits of the bin field)
    GS_SORT_SKEWED=1        bins drawn from a bell around the image centre instead of uniformly
usage: python tools/sort_bench.py [reps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_3d_gaussian_splatting_amd import hip_ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
grouped = os.environ.get("GS_SORT_GROUPED", "0") == "1"   # the frame's contract: bins in any order
skewed = os.environ.get("GS_SORT_SKEWED", "0") == "1"     # bins drawn from a bell around the image centre
tag = " ".join(f"{k}={os.environ[k]}" for k in ("GS_SORT_IMPL", "GS_SORT_MSD_BITS", "GS_SORT_GROUPED", "GS_SORT_SKEWED")
               if k in os.environ)
ws = hip_ops.Workspaces()
for n, depth_bits, tile_bits in ((48_000, 9, 8), (360_000, 11, 11), (1_130_000, 9, 12), (2_877_171, 11, 11),
                                 (4_400_000, 11, 11), (9_500_000, 9, 13)):
    rng = np.random.default_rng(n)
    tile = (np.clip(rng.normal(0.5, 0.15, size=n) * (1 << tile_bits), 0, (1 << tile_bits) - 1).astype(np.int64) if skewed
            else rng.integers(0, 1 << tile_bits, size=n))
    keys = (rng.integers(0, 1 << depth_bits, size=n) + (tile << depth_bits)).astype(np.uint32)
    k0 = torch.from_numpy(keys.view(np.int32)).cuda()
    p0 = torch.arange(n, dtype=torch.int32, device="cuda")
    times = []
    ok = None
    for r in range(reps + 3):
        k, p = k0.clone(), p0.clone()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        k, p = hip_ops.sort_pairs(k, p, depth_bits, tile_bits, depth_bits, in_place=False, ws=ws, bins_in_any_order=grouped)
        b.record()
        torch.cuda.synchronize()
        if r >= 3:
            times.append(a.elapsed_time(b) * 1e3)
        if ok is None:
            order = np.argsort(keys, kind="stable")
            got_k, got_p = k.cpu().numpy().view(np.uint32), p.cpu().numpy()
            if grouped:   # bins in the sort's own order: re-order them stably before comparing
                by_bin = np.argsort(got_k >> depth_bits, kind="stable")
                got_k, got_p = got_k[by_bin], got_p[by_bin]
            ok = bool(np.array_equal(got_k, keys[order]) and np.array_equal(got_p, order.astype(np.int32)))
    times.sort()
    print(f"[sort_bench] {tag or 'default'} n={n} bits={depth_bits + tile_bits} median={times[len(times) // 2]:.1f}us "
          f"min={times[0]:.1f}us correct={ok}", flush=True)
