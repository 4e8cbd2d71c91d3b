#!/usr/bin/env python
"""Rank time of an OWNER-SHARDED frame (owner_sharding.py) on ONE GPU: all G ranks are played in lockstep
(simulate_frame: every rank's device work as under torch.distributed -- projection of its own N/G rows, routing, key count
/ sort / blend of the records it receives, backward of its band, gather of the returned rows, per-point backward of its
own rows -- the two all-to-alls and the all-gather as device copies), and each rank's phases are timed with HIP events.
What is NOT in a rank's time: the wire (computed from the byte counts printed here) and the host's size reads.
usage: python tools/owner_shard_bench.py [workload] ;  GS_SHARD_WORLDS=1,2,4,8  GS_BIN_SHIFT=n  GS_SHARD_REPS=10
GS_SHARD_BALANCE=1: after the warm-up frames the band boundaries are moved to balance the ranks' walk lengths
(owner_sharding.balanced_row_weights), as OwnerShardedRasterisation.rebalance_every does under torch.distributed"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Op  # noqa: E402
from taichi_3d_gaussian_splatting_amd import host_affinity  # noqa: E402
from taichi_3d_gaussian_splatting_amd.owner_sharding import (OwnerShardedRasteriser, balanced_row_weights,  # noqa: E402
                                                               owned_point_rows, simulate_frame)
from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene, make_grad_image  # noqa: E402

if os.environ.get("GS_NO_PIN") != "1":
    host_affinity.pin_host_threads(0)
workload = sys.argv[1] if len(sys.argv) > 1 else "headline_1m_1080p"
s = make_config_scene(workload).to("cuda")
g = make_grad_image(s.height, s.width).to("cuda")
reps = int(os.environ.get("GS_SHARD_REPS", "10"))
cfg = Op.GaussianPointCloudRasterisationConfig(near_plane=s.near_plane, far_plane=s.far_plane,
                                               depth_to_sort_key_scale=s.depth_to_sort_key_scale)
n = s.point_cloud.shape[0]
for G in tuple(int(x) for x in os.environ.get("GS_SHARD_WORLDS", "1,2,4,8").split(",")):
    cores = [OwnerShardedRasteriser(cfg, r, G, backward_valid_point_hook=lambda h: None) for r in range(G)]
    for c in cores:
        c.always_store_normalised_rotation = True     # training-like, as bench.py
        if os.environ.get("GS_BIN_SHIFT"):
            c.bin_shift = int(os.environ["GS_BIN_SHIFT"])
    blocks = [owned_point_rows(n, r, G) for r in range(G)]
    feats = [s.point_cloud_features[b.start:b.stop].clone() for b in blocks]
    inputs = [Op.GaussianPointCloudRasterisationInput(
        point_cloud=s.point_cloud[b.start:b.stop], point_cloud_features=feats[r],
        point_object_id=s.point_object_id[b.start:b.stop], point_invalid_mask=s.point_invalid_mask[b.start:b.stop],
        camera_info=CameraInfo(s.camera_intrinsics, s.height, s.width, 0), q_pointcloud_camera=s.q_pointcloud_camera,
        t_pointcloud_camera=s.t_pointcloud_camera, color_max_sh_band=3) for r, b in enumerate(blocks)]
    for _ in range(3):
        simulate_frame(cores, inputs, g)
    if os.environ.get("GS_SHARD_BALANCE") == "1":
        for _ in range(2):   # (a second look with the new boundaries in force: stays put when they are balanced)
            work = []
            simulate_frame(cores, inputs, g, row_work=work)
            weights = balanced_row_weights(work, G, cores[0].weights_for(s.height)) if work else None
            for c in cores:
                c.row_weights = weights
            for _ in range(3):
                simulate_frame(cores, inputs, g)
    per_rank = {}
    last = None
    for _ in range(reps):
        t = {}
        simulate_frame(cores, inputs, g, timings=t)
        last = t
        for r in range(G):
            for k, v in t[r].items():
                per_rank.setdefault(r, {}).setdefault(k, []).append(v)
    med = {r: {k: sorted(v)[len(v) // 2] for k, v in d.items()} for r, d in per_rank.items()}
    totals = {r: sum(d.values()) for r, d in med.items()}
    worst = max(totals, key=totals.get)
    cap = last["capacity"]
    sent = last["records_sent"]
    print(f"[owner_shard] {workload} G={G}: slowest rank {worst} {totals[worst]:.3f} ms "
          f"({', '.join(f'{k} {v:.3f}' for k, v in med[worst].items())}); rank times "
          f"{' '.join(f'{totals[r]:.3f}' for r in range(G))} ms; chunk capacity {cap}, records sent per rank "
          f"{min(sent)}..{max(sent)} (visible {min(last['visible'])}..{max(last['visible'])}), "
          f"wire per rank forward {G * (cap + 1) * 64 / 1e6:.2f} MB padded / {max(sent) * 64 / 1e6:.2f} MB of records, "
          f"backward {G * (cap + 1) * 48 / 1e6:.2f} / {max(sent) * 48 / 1e6:.2f} MB; "
          f"band keys {[f['keys'] for f in last['frames']]}, bin_shift {[f['bin_shift'] for f in last['frames']]}, "
          f"band rows {cores[0].band_bounds(s.height)}", flush=True)
