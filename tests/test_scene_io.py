"""Scene / dataset formats either side of the hot path -- CPU tests, plus one GPU test of
the render script."""
import json
import os
import struct
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest
import torch

from taichi_3d_gaussian_splatting_amd.GaussianPointCloudScene import FEATURE_COLUMNS, GaussianPointCloudScene as Scene
from taichi_3d_gaussian_splatting_amd.ImagePoseDataset import ImagePoseDataset, _resized_hw
from taichi_3d_gaussian_splatting_amd.synthetic import make_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parquet_round_trip_and_schema(tmp_path):
    s = make_scene(n=300, height=64, width=64, s_min=0.01, s_max=0.05, seed=2)
    scene = Scene(s.point_cloud, Scene.PointCloudSceneConfig(max_num_points_ratio=1.5),
                  point_cloud_features=s.point_cloud_features)
    assert scene.point_cloud.shape[0] == 450 and int(scene.point_invalid_mask.sum()) == 150
    path = str(tmp_path / "scene.parquet")
    scene.to_parquet(path)
    df = pd.read_parquet(path)
    assert list(df.columns) == ["x", "y", "z"] + FEATURE_COLUMNS and len(df) == 300   # SCN:136-146
    back = Scene.from_parquet(path)
    assert torch.equal(back.point_cloud.data, s.point_cloud) and torch.equal(back.point_cloud_features.data,
                                                                           s.point_cloud_features)
    assert back.point_invalid_mask.dtype == torch.int8 and back.point_object_id.dtype == torch.int32


def test_raw_point_cloud_is_initialised(tmp_path):
    rng = np.random.default_rng(0)
    pts = rng.random((200, 3)).astype(np.float32)
    rgb = rng.integers(0, 256, (200, 3)).astype(np.float64)
    path = str(tmp_path / "raw.parquet")
    pd.DataFrame(np.concatenate([pts, rgb], 1), columns=["x", "y", "z", "r", "g", "b"]).to_parquet(path)
    scene = Scene.from_parquet(path, Scene.PointCloudSceneConfig(initial_alpha=-1.5))
    f = scene.point_cloud_features.data
    assert torch.allclose(f[:, :4].norm(dim=1), torch.ones(200), atol=1e-5)       # unit quaternions
    assert torch.all(f[:, 7] == -1.5) and not f[:, 9:24].any() and not f[:, 41:56].any()
    # isotropic log-scale = log(mean distance to the 3 nearest neighbours)      (SCN:80-92)
    from scipy.spatial import cKDTree
    d, _ = cKDTree(pts).query(pts, k=4)
    assert np.allclose(f[:, 4].numpy(), np.log(d[:, 1:].mean(1)), atol=1e-5) and torch.equal(f[:, 4], f[:, 6])
    # SH DC reproduces the point colour through sigmoid(SH_C0 * dc)
    col = torch.sigmoid(0.28209479177387814 * f[:, [8, 24, 40]])
    assert torch.allclose(col, torch.tensor(rgb / 255.0, dtype=torch.float32).clamp(0, 0.99), atol=1e-5)
    sphere = Scene.from_parquet(path, Scene.PointCloudSceneConfig(add_sphere=True, num_points_sphere=50))
    assert sphere.point_cloud.shape[0] == 250


def test_ply_export_layout(tmp_path):
    s = make_scene(n=10, height=64, width=64, s_min=0.01, s_max=0.05, seed=3)
    scene = Scene(s.point_cloud, Scene.PointCloudSceneConfig(), point_cloud_features=s.point_cloud_features)
    path = str(tmp_path / "scene.ply")
    scene.to_ply(path)
    raw = open(path, "rb").read()
    header, body = raw.split(b"end_header\n", 1)
    names = [l.split()[-1] for l in header.decode().splitlines() if l.startswith("property")]
    assert names[:6] == ["x", "y", "z", "nx", "ny", "nz"] and names[6:9] == ["f_dc_0", "f_dc_1", "f_dc_2"]
    assert names[9] == "f_rest_0" and names[53] == "f_rest_44" and names[54] == "opacity"
    assert names[55:58] == ["scale_0", "scale_1", "scale_2"] and names[58:] == ["rot_0", "rot_1", "rot_2", "rot_3"]
    assert len(body) == 10 * 62 * 4
    row0 = struct.unpack("<62f", body[:62 * 4])
    f = s.point_cloud_features[0]
    assert np.allclose(row0[0:3], s.point_cloud[0].numpy()) and row0[6] == f[8] and row0[7] == f[24]
    assert row0[9] == f[9] and row0[9 + 15] == f[25]            # f_rest is channel-major (SCN:156-157)
    assert row0[58:62] == (f[3], f[0], f[1], f[2])              # rot = (w, x, y, z)      (SCN:160)


def _write_dataset(tmp_path, sizes):
    from PIL import Image
    recs = []
    for i, (h, w) in enumerate(sizes):
        p = str(tmp_path / f"img{i}.png")
        Image.fromarray((np.random.default_rng(i).random((h, w, 3)) * 255).astype(np.uint8)).save(p)
        T = np.eye(4); T[2, 3] = -3.0
        recs.append(dict(image_path=p, T_pointcloud_camera=T.tolist(),
                         camera_intrinsics=[[2 * w, 0, w / 2], [0, 2 * w, h / 2], [0, 0, 1]],
                         camera_height=2 * h, camera_width=2 * w, camera_id=i))   # JSON size != image size
    path = str(tmp_path / "ds.json")
    json.dump(recs, open(path, "w"))
    return path


def test_dataset_resolution_rules(tmp_path):
    path = _write_dataset(tmp_path, [(100, 150), (1700, 900)])
    ds = ImagePoseDataset(path)
    image, q, t, info = ds[0]
    assert image.shape == (3, 96, 144) and (info.camera_height, info.camera_width) == (96, 144)   # crop to /16
    assert q.shape == (1, 4) and t.shape == (1, 3) and torch.allclose(t, torch.tensor([[0., 0., -3.]]))
    # intrinsics follow the real image size (half the JSON size here)            (DST:77-81)
    assert torch.allclose(info.camera_intrinsics, torch.tensor([[150., 0, 37.5], [0, 150., 25.], [0, 0, 1]]))
    image, _, _, info = ds[1]   # 1700 x 900 -> cropped 1696 x 896 -> resized like torchvision(size=1024, max_size=1600)
    assert _resized_hw(1696, 896) == (1600, 845)
    assert (info.camera_height, info.camera_width) == (1600, 832) and image.shape == (3, 1600, 832)
    assert abs(info.camera_intrinsics[0, 0].item() - 900 * (845 / 896)) < 1e-3
    poses_only = ImagePoseDataset(path, load_images=False)[0]
    assert poses_only[0] is None and poses_only[3].camera_height == 192   # sizes from the JSON, /16


@pytest.mark.gpu
def test_render_script_end_to_end(tmp_path):
    from oracle import gs_oracle as O
    s = make_scene(n=1500, height=64, width=96, s_min=0.02, s_max=0.08, seed=5)
    Scene(s.point_cloud, Scene.PointCloudSceneConfig(), point_cloud_features=s.point_cloud_features).to_parquet(
        str(tmp_path / "scene.parquet"))
    T = torch.eye(4); T[2, 3] = -3.0
    recs = [dict(image_path="unused.png", T_pointcloud_camera=T.tolist(), camera_intrinsics=s.camera_intrinsics.tolist(),
                 camera_height=64, camera_width=96, camera_id=0)]
    json.dump(recs, open(tmp_path / "poses.json", "w"))
    out = tmp_path / "frames"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "gaussian_point_render.py"), "--parquet_path",
                           str(tmp_path / "scene.parquet"), "--poses", str(tmp_path / "poses.json"),
                           "--output_prefix", str(out)], cwd=ROOT)
    from PIL import Image
    frame = np.asarray(Image.open(out / "frame_000.png"), dtype=np.float32) / 255.0
    ref = O.forward(s.point_cloud.numpy(), s.point_cloud_features.numpy(), s.point_invalid_mask.numpy(),
                    s.point_object_id.numpy(), s.camera_intrinsics.numpy(), s.q_pointcloud_camera.numpy(),
                    s.t_pointcloud_camera.numpy(), 64, 96)["image"]
    assert np.abs(frame - np.clip(ref, 0, 1)).max() <= 1.0 / 255.0 + 1e-6   # 8-bit quantisation (truncation)


def test_ply_import_round_trip_and_formats(tmp_path):
    """to_ply -> from_ply reproduces the scene (quaternions come back normalised); the reader also takes ascii
    and big-endian files and rejects what it does not support."""
    from taichi_3d_gaussian_splatting_amd.GaussianPointCloudScene import _read_ply_vertices
    s = make_scene(n=37, height=64, width=64, s_min=0.01, s_max=0.05, seed=8)
    scene = Scene(s.point_cloud, Scene.PointCloudSceneConfig(), point_cloud_features=s.point_cloud_features)
    path = str(tmp_path / "a.ply")
    scene.to_ply(path)
    back = Scene.from_ply(path)
    assert torch.equal(back.point_cloud.data, s.point_cloud)
    assert torch.allclose(back.point_cloud_features.data, s.point_cloud_features, atol=1e-6)   # unit quaternions already
    assert torch.equal(back.point_cloud_features.data[:, 4:], s.point_cloud_features[:, 4:])
    v = _read_ply_vertices(path)
    names = v.dtype.names
    # the same content as ascii and as big-endian binary
    ascii_path, be_path = str(tmp_path / "b.ply"), str(tmp_path / "c.ply")
    header = ["ply", "format ascii 1.0", "comment made by a test", f"element vertex {len(v)}"] + \
        [f"property float {n}" for n in names] + ["end_header"]
    with open(ascii_path, "w") as fh:
        fh.write("\n".join(header) + "\n")
        for row in v:
            fh.write(" ".join(repr(float(x)) for x in row) + "\n")
    with open(be_path, "wb") as fh:
        fh.write(("\n".join(["ply", "format binary_big_endian 1.0", f"element vertex {len(v)}"] +
                            [f"property float {n}" for n in names] + ["end_header"]) + "\n").encode())
        fh.write(v.astype([(n, ">f4") for n in names]).tobytes())
    for other in (ascii_path, be_path):
        again = Scene.from_ply(other)
        assert torch.equal(again.point_cloud.data, back.point_cloud.data)
        assert torch.equal(again.point_cloud_features.data, back.point_cloud_features.data)
    bad = str(tmp_path / "d.ply")
    with open(bad, "wb") as fh:
        fh.write(b"ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nend_header\n0\n")
    with pytest.raises((ValueError, KeyError)):
        Scene.from_ply(bad)
    with open(bad, "wb") as fh:
        fh.write(b"obj\n")
    with pytest.raises(ValueError):
        _read_ply_vertices(bad)
