"""Scene state and on-disk formats around the rasteriser.

Mirror of the reference's ``taichi_3d_gaussian_splatting/GaussianPointCloudScene.py`` (SCN): the
``nn.Module`` that owns ``point_cloud[N,3]``, ``point_cloud_features[N,56]`` and the buffers
``point_invalid_mask int8[N]`` / ``point_object_id int32[N]`` (SCN:12-72), the KD-tree based initialisation
(SCN:74-130), the parquet schema ``x,y,z, cov_q0-3, cov_s0-2, alpha0, r_sh0-15, g_sh0-15, b_sh0-15``
(SCN:132-146,183-211) and the 3DGS-compatible PLY export with the quaternion re-ordered to (w,x,y,z)
(SCN:148-180).  Host-side code (pandas / scipy / numpy); nothing here is on the GPU hot path.
The PLY writer is self-contained (binary little-endian, no ``plyfile`` dependency).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Union

import numpy as np
import pandas as pd
import torch
import torch.nn as nn
from scipy.spatial import cKDTree

def _require_parquet_engine() -> None:
    """pandas needs pyarrow for parquet.  On a cold container image the first import of pyarrow's shared
    libraries has been seen to fail transiently; retry a few times before giving up loudly."""
    import importlib
    import time
    last = None
    for _ in range(5):
        try:
            importlib.import_module("pyarrow")
            return
        except ImportError as exc:  # pragma: no cover - only on a cold image
            last = exc
            time.sleep(0.5)
    raise ImportError(f"pyarrow is required for the parquet scene format: {last}")


FEATURE_COLUMNS = ([f"cov_q{i}" for i in range(4)] + [f"cov_s{i}" for i in range(3)] + ["alpha0"] +
                   [f"{c}_sh{i}" for c in "rgb" for i in range(16)])
SH_C0 = 0.28209479177387814


class GaussianPointCloudScene(nn.Module):
    @dataclass
    class PointCloudSceneConfig:
        num_of_features: int = 56
        max_num_points_ratio: Optional[float] = None   # > 1: pre-allocate room for densification
        add_sphere: bool = False
        sphere_radius_factor: float = 4.0
        num_points_sphere: int = 10000
        max_initial_covariance: Optional[float] = None
        initial_alpha: float = -2.0
        initial_covariance_ratio: float = 1.0

    def __init__(self, point_cloud: Union[np.ndarray, torch.Tensor], config: "GaussianPointCloudScene.PointCloudSceneConfig",
                 point_cloud_features: Optional[torch.Tensor] = None, point_object_id: Optional[torch.Tensor] = None):
        super().__init__()
        if point_cloud.ndim != 2 or point_cloud.shape[1] != 3:
            raise ValueError("point_cloud must be [N,3]")
        xyz = torch.as_tensor(point_cloud, dtype=torch.float32)
        n_valid = xyz.shape[0]
        if point_cloud_features is not None:
            point_cloud_features = torch.as_tensor(point_cloud_features, dtype=torch.float32)
        if config.max_num_points_ratio is not None:
            capacity = int(n_valid * config.max_num_points_ratio)
            if capacity <= n_valid:
                raise ValueError("max_num_points_ratio should be greater than 1.0")
            xyz = torch.cat([xyz, torch.zeros(capacity - n_valid, 3)], dim=0)
            if point_cloud_features is not None:
                point_cloud_features = torch.cat(
                    [point_cloud_features, torch.zeros(capacity - n_valid, config.num_of_features)], dim=0)
        self.config = config
        self.point_cloud = nn.Parameter(xyz.contiguous())
        if point_cloud_features is None:
            point_cloud_features = torch.zeros(xyz.shape[0], config.num_of_features)
        self.point_cloud_features = nn.Parameter(point_cloud_features.contiguous())
        invalid = torch.zeros(xyz.shape[0], dtype=torch.int8)
        invalid[n_valid:] = 1  # spare rows are masked out until the controller fills them
        self.register_buffer("point_invalid_mask", invalid)
        if point_object_id is None:
            point_object_id = torch.zeros(xyz.shape[0], dtype=torch.int32)
        self.register_buffer("point_object_id", point_object_id.to(torch.int32))

    def forward(self):
        return self.point_cloud, self.point_cloud_features

    # ------------------------------------------------------------------ initialisation (SCN:74-130)
    @torch.no_grad()
    def initialize(self, point_cloud_rgb: Optional[np.ndarray] = None) -> None:
        """Isotropic Gaussians whose scale is the mean distance to the 3 nearest neighbours, random unit
        quaternions, opacity logit ``initial_alpha``, SH DC from the point colour (or 1.0), higher orders 0."""
        valid = self.point_invalid_mask == 0
        pts = self.point_cloud[valid].detach().cpu().numpy()
        f = self.point_cloud_features
        if pts.shape[0] > 0:
            k = min(4, pts.shape[0])
            dist, _ = cKDTree(pts).query(pts, k=k)
            dist = np.asarray(dist).reshape(pts.shape[0], k)
            scale = dist[:, 1:].mean(axis=1) if k > 1 else np.full(pts.shape[0], 1e-6)
            scale = np.clip(scale * self.config.initial_covariance_ratio, 1e-6, self.config.max_initial_covariance)
            f[valid, 4:7] = torch.tensor(np.log(scale), dtype=torch.float32, device=f.device).unsqueeze(1)
        q = torch.rand_like(f[:, 0:4])
        f[:, 0:4] = q / q.norm(dim=1, keepdim=True)
        f[:, 7] = self.config.initial_alpha
        f[:, 8:] = 0.0
        for base in (8, 24, 40):
            f[:, base] = 1.0
        if point_cloud_rgb is not None:
            rgb = torch.as_tensor(point_cloud_rgb, dtype=torch.float32, device=f.device) / 255.0
            rgb = rgb.clamp(0.0, 0.99)
            logit = torch.log(rgb / (1.0 - rgb)) / SH_C0  # sigmoid(SH_C0 * dc) = rgb
            for ch, base in enumerate((8, 24, 40)):
                f[valid, base] = logit[:, ch]

    # ------------------------------------------------------------------ parquet (SCN:132-146,183-211)
    def to_parquet(self, path: str) -> None:
        valid = (self.point_invalid_mask == 0).cpu()
        xyz = self.point_cloud.detach().cpu()[valid].numpy()
        feat = self.point_cloud_features.detach().cpu()[valid].numpy()
        df = pd.concat([pd.DataFrame(xyz, columns=["x", "y", "z"]), pd.DataFrame(feat, columns=FEATURE_COLUMNS)], axis=1)
        _require_parquet_engine()
        df.to_parquet(path)

    @staticmethod
    def from_parquet(path: str, config: Optional["GaussianPointCloudScene.PointCloudSceneConfig"] = None):
        """A parquet with the 56 feature columns is a checkpoint (features are taken as they are); one with only
        x,y,z (+ optional r,g,b in 0..255) is a raw point cloud and is initialised."""
        config = config or GaussianPointCloudScene.PointCloudSceneConfig()
        _require_parquet_engine()
        df = pd.read_parquet(path)
        if config.add_sphere:
            df = GaussianPointCloudScene._add_sphere(df, config.sphere_radius_factor, config.num_points_sphere)
        xyz = df[["x", "y", "z"]].to_numpy(dtype=np.float32)
        if set(FEATURE_COLUMNS).issubset(df.columns):
            feat = torch.from_numpy(df[FEATURE_COLUMNS].to_numpy(dtype=np.float32))
            return GaussianPointCloudScene(xyz, config, point_cloud_features=feat)
        scene = GaussianPointCloudScene(xyz, config)
        has_color = {"r", "g", "b"}.issubset(df.columns)
        scene.initialize(point_cloud_rgb=df[["r", "g", "b"]].to_numpy() if has_color else None)
        return scene

    @staticmethod
    def _add_sphere(df: pd.DataFrame, radius_factor: float, num_points: int) -> pd.DataFrame:
        """Sky sphere (SCN:213-239): uniformly sampled points on a sphere of radius = half the largest extent of
        the cloud x radius_factor, mid-grey if the cloud has colours."""
        extent = max(df[c].max() - df[c].min() for c in "xyz") / 2.0
        radius = extent * radius_factor
        # own, fixed-seed generator (the reference draws from the global numpy state, SCN:222-223): every rank of a
        # multi-GPU run builds its replica of the scene from the parquet and the replicas must be identical
        rng = np.random.default_rng(0x5EED)
        phi = 2.0 * np.pi * rng.random(num_points)
        theta = np.arccos(2.0 * rng.random(num_points) - 1.0)
        pts = {"x": radius * np.sin(theta) * np.cos(phi), "y": radius * np.sin(theta) * np.sin(phi),
               "z": radius * np.cos(theta)}
        if {"r", "g", "b"}.issubset(df.columns):
            for c in "rgb":
                pts[c] = np.full(num_points, 255 // 2, dtype=np.float64)
        return pd.concat([df, pd.DataFrame(pts)], ignore_index=True)

    # ------------------------------------------------------------------ 3DGS-compatible PLY (SCN:148-180)
    def to_ply(self, path: str) -> None:
        """Binary little-endian PLY with the attribute names/order of the official 3DGS viewer:
        x y z nx ny nz f_dc_0..2 f_rest_0..44 opacity scale_0..2 rot_0..3 (rot = w,x,y,z)."""
        valid = (self.point_invalid_mask == 0).cpu()
        xyz = self.point_cloud.detach().cpu()[valid].numpy().astype(np.float32)
        feat = self.point_cloud_features.detach().cpu()[valid].numpy().astype(np.float32)
        sh = feat[:, 8:56].reshape(-1, 3, 16)
        columns = ([("x", xyz[:, 0]), ("y", xyz[:, 1]), ("z", xyz[:, 2])] +
                   [(n, np.zeros(len(xyz), np.float32)) for n in ("nx", "ny", "nz")] +
                   [(f"f_dc_{c}", sh[:, c, 0]) for c in range(3)] +
                   [(f"f_rest_{i}", sh[:, :, 1:].reshape(len(xyz), 45)[:, i]) for i in range(45)] +
                   [("opacity", feat[:, 7])] + [(f"scale_{i}", feat[:, 4 + i]) for i in range(3)] +
                   [(f"rot_{i}", feat[:, j]) for i, j in enumerate((3, 0, 1, 2))])
        rec = np.empty(len(xyz), dtype=[(name, "<f4") for name, _ in columns])
        for name, values in columns:
            rec[name] = values
        header = ["ply", "format binary_little_endian 1.0", f"element vertex {len(xyz)}"]
        header += [f"property float {name}" for name, _ in columns] + ["end_header"]
        with open(path, "wb") as fh:
            fh.write(("\n".join(header) + "\n").encode("ascii"))
            fh.write(rec.tobytes())

    @staticmethod
    def from_ply(path: str, config: Optional["GaussianPointCloudScene.PointCloudSceneConfig"] = None):
        """Load a 3DGS-format PLY (what ``to_ply`` writes and what the official implementation trains): the
        mapping the reference's benchmark uses (benchmark/inference_benchmark.py:21-82) -- ``f_rest`` is
        channel-major [3][15], ``rot`` is (w,x,y,z) and is normalised, opacity / scales stay logit / log."""
        config = config or GaussianPointCloudScene.PointCloudSceneConfig()
        v = _read_ply_vertices(path)
        n = v.shape[0]
        names = v.dtype.names
        rest = sorted((k for k in names if k.startswith("f_rest_")), key=lambda k: int(k.split("_")[-1]))
        if len(rest) != 45:
            raise ValueError(f"{path}: expected 45 f_rest_* properties (SH degree 3), found {len(rest)}")
        xyz = np.stack([v["x"], v["y"], v["z"]], axis=1).astype(np.float32)
        feat = np.zeros((n, 56), np.float32)
        rot = np.stack([v[f"rot_{i}"] for i in (1, 2, 3, 0)], axis=1).astype(np.float64)   # wxyz -> xyzw
        feat[:, 0:4] = rot / np.linalg.norm(rot, axis=1, keepdims=True)
        feat[:, 4:7] = np.stack([v[f"scale_{i}"] for i in range(3)], axis=1)
        feat[:, 7] = v["opacity"]
        extra = np.stack([v[k] for k in rest], axis=1).reshape(n, 3, 15)
        for ch in range(3):
            feat[:, 8 + 16 * ch] = v[f"f_dc_{ch}"]
            feat[:, 9 + 16 * ch: 24 + 16 * ch] = extra[:, ch]
        return GaussianPointCloudScene(xyz, config, point_cloud_features=torch.from_numpy(feat))


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
              "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
              "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def _read_ply_vertices(path: str) -> np.ndarray:
    """Minimal PLY reader: the scalar properties of the ``vertex`` element (binary little/big endian or ascii) as a
    structured array.  List properties and other elements before ``vertex`` are not supported."""
    with open(path, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex, seen_other = None, None, [], False, False
        while True:
            line = fh.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    if seen_other:
                        raise ValueError(f"{path}: elements before 'vertex' are not supported")
                    count = int(tok[2])
                elif count is None:
                    seen_other = True
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties on vertices are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt is None or count is None:
            raise ValueError(f"{path}: header lacks format or vertex element")
        if fmt == "ascii":
            rows = np.loadtxt(fh, max_rows=count, ndmin=2)
            out = np.empty(count, dtype=[(nm, "<" + tp) for nm, tp in props])
            for i, (nm, _) in enumerate(props):
                out[nm] = rows[:, i]
            return out
        order = "<" if fmt == "binary_little_endian" else ">"
        dtype = np.dtype([(nm, order + tp) for nm, tp in props])
        data = np.frombuffer(fh.read(count * dtype.itemsize), dtype=dtype, count=count)
        return data
