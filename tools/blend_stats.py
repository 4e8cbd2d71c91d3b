#!/usr/bin/env python
"""Path statistics of the two-waves-per-tile blend kernels on one workload: how many (wave, list entry) visits evaluate alpha,
how many of them run the hit path and with how many live lanes, how often a decision lands inside a bracket and is settled
by the reference's expression, how many pixel histories are replayed (development tool; run through gpurun).

Needs a counting build of the library:
    tools/build_variants.sh "stats:-DGS_TUNING_BUILD=1 -DGS_STATS=1"
    GS_LIB_PATH=variants/libgsplat_hip_stats.so python tools/blend_stats.py [workload]
"""
import ctypes
import json
import os
import sys

import torch

os.environ.setdefault("GS_ALLOW_TUNING_LIB", "1")   # this tool measures a counting build (never the product path)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_3d_gaussian_splatting_amd import _lib  # noqa: E402
from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Op  # noqa: E402
from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene, make_grad_image  # noqa: E402

NAMES = ["fwd_entries", "fwd_hit_entries", "fwd_hit_pixels", "fwd_hit_lanes", "fwd_careful_entries", "fwd_exact_alpha",
         "fwd_replays", "fwd_replay_entries", "bwd_entries", "bwd_hit_entries", "bwd_hit_pixels", "bwd_hit_lanes",
         "bwd_bracketed", "bwd_exact_alpha", "fwd_hit_blocks", "bwd_hit_blocks"]


def read(clear=True):
    buf = (ctypes.c_uint64 * 16)()
    counting = _lib.load().gs_blend_read_stats(buf, int(clear), None)
    assert counting == 1, "not a counting build: tools/build_variants.sh 'stats:-DGS_TUNING_BUILD=1 -DGS_STATS=1' and set GS_LIB_PATH"
    return dict(zip(NAMES, list(buf)))


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "headline_1m_1080p"
    s = make_config_scene(workload).to("cuda")
    g = make_grad_image(s.height, s.width).to("cuda")
    op = Op(Op.GaussianPointCloudRasterisationConfig(near_plane=s.near_plane, far_plane=s.far_plane,
                                                     depth_to_sort_key_scale=s.depth_to_sort_key_scale))
    xyz = s.point_cloud.clone().requires_grad_(True)
    feat = s.point_cloud_features.clone().requires_grad_(True)
    inp = Op.GaussianPointCloudRasterisationInput(
        point_cloud=xyz, point_cloud_features=feat, point_object_id=s.point_object_id, point_invalid_mask=s.point_invalid_mask,
        camera_info=CameraInfo(camera_intrinsics=s.camera_intrinsics, camera_height=s.height, camera_width=s.width, camera_id=0),
        q_pointcloud_camera=s.q_pointcloud_camera, t_pointcloud_camera=s.t_pointcloud_camera, color_max_sh_band=3)
    for _ in range(2):   # (the second frame runs with the first one's capacities: the steady state)
        read()
        image, _depth, _count = op(inp)
        (image * g).sum().backward()
        torch.cuda.synchronize()
    c = read()
    out = dict(workload=workload, counters=c)
    if c["fwd_entries"]:
        out["forward"] = dict(
            hit_path_share_of_visits=c["fwd_hit_entries"] / c["fwd_entries"],
            live_pixels_per_hit_visit_of_128=c["fwd_hit_pixels"] / max(c["fwd_hit_entries"], 1),
            live_lanes_per_hit_visit_of_64=c["fwd_hit_lanes"] / max(c["fwd_hit_entries"], 1),
            blocks_of_8x8_with_a_hit_per_hit_visit_of_2=c["fwd_hit_blocks"] / max(c["fwd_hit_entries"], 1),
            careful_share_of_visits=c["fwd_careful_entries"] / c["fwd_entries"],
            replayed_entries_per_replay=c["fwd_replay_entries"] / max(c["fwd_replays"], 1))
    if c["bwd_entries"]:
        out["backward"] = dict(
            hit_path_share_of_visits=c["bwd_hit_entries"] / c["bwd_entries"],
            live_pixels_per_hit_visit_of_128=c["bwd_hit_pixels"] / max(c["bwd_hit_entries"], 1),
            live_lanes_per_hit_visit_of_64=c["bwd_hit_lanes"] / max(c["bwd_hit_entries"], 1),
            blocks_of_8x8_with_a_hit_per_hit_visit_of_2=c["bwd_hit_blocks"] / max(c["bwd_hit_entries"], 1),
            bracketed_share_of_visits=c["bwd_bracketed"] / c["bwd_entries"])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
