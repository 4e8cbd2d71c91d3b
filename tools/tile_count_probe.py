#!/usr/bin/env python
"""Looks into the tile boxes of ONE draw of tests/test_fuzz_gpu.py (development tool; run through gpurun): the rows whose
reference box count (`num_overlap_tiles`, RAS:106-128) differs between the HIP projection and the fp32 oracle, with both
sides' radius and the distance of the box edges to the next tile boundary.  usage: python tools/tile_count_probe.py <case>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_3d_gaussian_splatting_amd import hip_ops  # noqa: E402
from tests.helpers import oracle_forward  # noqa: E402
from tests.test_fuzz_gpu import random_scene  # noqa: E402

case = int(sys.argv[1])
scene, band, needles, opt = random_scene(case)
f = oracle_forward(scene)
s = scene.to("cuda")
q_cp, t_cp = hip_ops.pose_inverse(s.q_pointcloud_camera, s.t_pointcloud_camera)
mask, ids, counters = hip_ops.filter_compact(s.point_cloud, s.point_invalid_mask, s.point_object_id, s.camera_intrinsics,
                                             q_cp, t_cp, s.near_plane, s.far_plane, s.width, s.height)
feat = s.point_cloud_features.clone()
attrs, ntiles, nkeys, _, _ = hip_ops.preprocess(s.point_cloud, feat, s.point_object_id, s.camera_intrinsics, q_cp, t_cp, ids,
                                                s.width, s.height, hip_ops.ListLayout(), s.depth_to_sort_key_scale, counters)
torch.cuda.synchronize()
a, nt = attrs.cpu().numpy(), ntiles.cpu().numpy()
m = len(f["ids"])
assert np.array_equal(ids.cpu().numpy()[:m], f["ids"])
print(f"case {case}: {scene.width}x{scene.height} M={m} needles={needles}; uv identical: {np.array_equal(a[:m, 0:2], f['uv'])}; "
      f"radius: {int((a[:m, 7] != f['radii']).sum())} of {m} differ in bits, largest relative difference "
      f"{float(np.abs(a[:m, 7] - f['radii']).max() / 1.0):.3e} px; box counts differ on {int((nt[:m] != f['num_overlap_tiles']).sum())} rows")
scales = np.exp(scene.point_cloud_features[:, 4:7].numpy())
for i in np.nonzero(nt[:m] != f["num_overlap_tiles"])[0]:
    u, v = f["uv"][i]
    r_h, r_o = np.float32(max(a[i, 7], 1.0)), np.float32(max(f["radii"][i], 1.0))
    edges = {}
    for name, c in (("u", u), ("v", v)):
        for sign in (-1, 1):
            e_h, e_o = np.float32(c + sign * r_h) / np.float32(16), np.float32(c + sign * r_o) / np.float32(16)
            edges[f"{name}{'+' if sign > 0 else '-'}r"] = (float(e_h), float(e_o))
    print(f"  row {i} (point {f['ids'][i]}): count hip {nt[i]} oracle {f['num_overlap_tiles'][i]}; uv ({u:.4f}, {v:.4f}); radius hip "
          f"{a[i, 7]!r} oracle {f['radii'][i]!r} (ulps apart: {abs(int(np.float32(a[i, 7]).view(np.int32)) - int(np.float32(f['radii'][i]).view(np.int32)))}); "
          f"scales {scales[f['ids'][i]]}")
    for k, (e_h, e_o) in edges.items():
        print(f"      ({k})/16: hip {e_h!r} oracle {e_o!r}  floor {int(np.floor(e_h))} / {int(np.floor(e_o))}")
