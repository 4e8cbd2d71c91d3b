#!/usr/bin/env python
"""Stage times of gs_preprocess and gs_make_keys at the headline size (2x2-tile bins, cull on, training-like write-back) for
the library named by GS_LIB_PATH -- the ablation builds of tools/build_ablations.sh (their outputs are garbage by design:
only the times are looked at).  Development tool; run through gpurun: python tools/ablate_probe.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from taichi_3d_gaussian_splatting_amd import hip_ops as ops
from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene
s = make_config_scene("headline_1m_1080p").to("cuda")
layout = ops.ListLayout(bin_shift=1, exact_cull=True)
q_cp, t_cp = ops.pose_inverse(s.q_pointcloud_camera, s.t_pointcloud_camera)
_, ids, counters = ops.filter_compact(s.point_cloud, s.point_invalid_mask, s.point_object_id, s.camera_intrinsics, q_cp, t_cp,
                                      s.near_plane, s.far_plane, s.width, s.height)
feat = s.point_cloud_features.clone()
tp, tk = [], []
for r in range(25):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    e[0].record()
    a, nfull, nkeys, bsums, bsf = ops.preprocess(s.point_cloud, feat, s.point_object_id, s.camera_intrinsics, q_cp, t_cp, ids,
                                                  s.width, s.height, layout, s.depth_to_sort_key_scale, always_store_rotation=True)
    e[1].record()
    c2 = torch.zeros(ops.NUM_COUNTERS, dtype=torch.int32, device="cuda")
    k, n_slots, mdk, _ = ops.scan_block_sums(bsums, c2, bsf)
    kdb, db, tb = ops.key_layout(s.near_plane, s.far_plane, s.depth_to_sort_key_scale, layout.num_bins(s.width, s.height), mdk)
    e[2].record()
    keys, payload, so = ops.make_keys(a, nkeys, bsums, max(k, 3_000_000), s.width, s.height, s.depth_to_sort_key_scale, layout, kdb, nfull, bsf)
    e[3].record()
    torch.cuda.synchronize()
    if r >= 5:
        tp.append(e[0].elapsed_time(e[1]) * 1e3); tk.append(e[2].elapsed_time(e[3]) * 1e3)
tp.sort(); tk.sort()
print(os.environ.get("GS_LIB_PATH", "default").split("_")[-1], "K", k, "preprocess median %.1f us  make_keys median %.1f us" % (tp[len(tp)//2], tk[len(tk)//2]))
