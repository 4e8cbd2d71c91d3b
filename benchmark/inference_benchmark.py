"""Forward-only frame time of the rasteriser, with the protocol of the reference's
``benchmark/inference_benchmark.py:13-160``: warm-up frames, then N timed frames under ``torch.no_grad`` between two
device events, cycling through the poses of a data set; prints ms/frame and FPS.

    python benchmark/inference_benchmark.py --ply point_cloud.ply --dataset train.json
    python benchmark/inference_benchmark.py --parquet scene_30000.parquet --dataset train.json
    python benchmark/inference_benchmark.py --synthetic headline_1m_1080p          # no data needed

(The reference hard-codes the paths, 1000 warm-up and 100 timed frames; they are arguments here.)
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation  # noqa: E402
from taichi_3d_gaussian_splatting_amd.GaussianPointCloudScene import GaussianPointCloudScene  # noqa: E402
from taichi_3d_gaussian_splatting_amd.ImagePoseDataset import ImagePoseDataset  # noqa: E402
from taichi_3d_gaussian_splatting_amd.synthetic import CONFIGS, make_config_scene  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    src = ap.add_mutually_exclusive_group(required=True)
    src.add_argument("--ply", help="3DGS-format PLY (official implementation / to_ply)")
    src.add_argument("--parquet", help="scene parquet written by the trainer")
    src.add_argument("--synthetic", choices=sorted(CONFIGS) + ["stress_t_ras", "trained_1080p"],
                     help="seeded synthetic workload, single pose; trained_1080p: the scene grown by the repository's own "
                          "trainer (trained_workload.py), cycled through its thirty training / validation cameras")
    ap.add_argument("--dataset", help="dataset JSON whose poses / intrinsics are cycled (required with --ply/--parquet)")
    ap.add_argument("--iterations", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--near_plane", type=float, default=0.8)
    ap.add_argument("--far_plane", type=float, default=1000.0)
    ap.add_argument("--depth_to_sort_key_scale", type=float, default=100.0)
    args = ap.parse_args()
    from taichi_3d_gaussian_splatting_amd import host_affinity
    host_affinity.pin_host_threads(torch.cuda.current_device())   # launching threads on one L3 complex next to the GPU
    dev = torch.device("cuda:0")

    if args.synthetic:
        s = make_config_scene(args.synthetic).to(dev)
        xyz, feat, invalid, obj = s.point_cloud, s.point_cloud_features, s.point_invalid_mask, s.point_object_id
        cam = CameraInfo(camera_intrinsics=s.camera_intrinsics, camera_height=s.height, camera_width=s.width, camera_id=0)
        views = [(s.q_pointcloud_camera, s.t_pointcloud_camera, cam)]
        if args.synthetic == "trained_1080p":   # the reference's protocol cycles the data set's poses (BENCH:109-160)
            from taichi_3d_gaussian_splatting_amd.trained_workload import load_or_make
            views = [(q.to(dev), t.to(dev), cam) for q, t in load_or_make("trained_1080p")["poses"]]
    else:
        if not args.dataset:
            ap.error("--dataset is required with --ply / --parquet")
        scene = (GaussianPointCloudScene.from_ply(args.ply) if args.ply
                 else GaussianPointCloudScene.from_parquet(args.parquet)).to(dev)
        xyz, feat = scene.point_cloud.detach(), scene.point_cloud_features.detach()
        invalid, obj = scene.point_invalid_mask, scene.point_object_id
        views = []
        for _, q, t, info in ImagePoseDataset(args.dataset, load_images=False):   # poses resident on the device
            views.append((q.to(dev), t.to(dev), CameraInfo(camera_intrinsics=info.camera_intrinsics.to(dev),
                                                            camera_height=info.camera_height,
                                                            camera_width=info.camera_width, camera_id=info.camera_id)))
    rasteriser = GaussianPointCloudRasterisation(GaussianPointCloudRasterisation.GaussianPointCloudRasterisationConfig(
        near_plane=args.near_plane, far_plane=args.far_plane, depth_to_sort_key_scale=args.depth_to_sort_key_scale))

    def frame(i: int):
        q, t, cam = views[i % len(views)]
        return rasteriser(GaussianPointCloudRasterisation.GaussianPointCloudRasterisationInput(
            point_cloud=xyz, point_cloud_features=feat, point_invalid_mask=invalid, point_object_id=obj,
            camera_info=cam, q_pointcloud_camera=q, t_pointcloud_camera=t, color_max_sh_band=3))

    with torch.no_grad():
        for i in range(args.warmup):
            frame(i)
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for i in range(args.iterations):
            frame(i)
        stop.record()
        torch.cuda.synchronize()
    ms = start.elapsed_time(stop) / args.iterations
    cam = views[0][2]
    print(f"Inference time: {ms} ms")
    print(f"FPS: {1000.0 / ms}")
    print(json.dumps({"ms_per_frame": ms, "fps": 1000.0 / ms, "points": int(xyz.shape[0]),
                      "image": f"{cam.camera_width}x{cam.camera_height}", "views": len(views),
                      "warmup": args.warmup, "iterations": args.iterations}))


if __name__ == "__main__":
    main()
