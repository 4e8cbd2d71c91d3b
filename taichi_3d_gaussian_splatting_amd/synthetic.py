"""Seeded synthetic scenes for parity tests and bench.py.

"Random-init point cloud": xyz ~ U([-1,1]^3); camera at z=-3 looking +z
(camera->world pose = identity rotation, t=(0,0,-3), convention of the
reference's docs/RawDataFormat.md:72-92); fx=fy=0.75*W, cx=W/2, cy=H/2;
q ~ N(0,1)^4 normalised; per-axis log-scales ~ U(log s_min, log s_max);
opacity logit ~ U(-2,2); SH DC ~ U(-2,2)/0.2820948, higher orders ~ N(0,0.3)
for "deg 3" data and exactly 0 for "deg 0" data.  Everything is drawn from a
CPU torch.Generator so the same bits are produced on every box.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

# (n_points, height, width, s_min, s_max, sh_degree, near, far, depth_scale) -- BASELINE.json configs
CONFIGS = {
    "cfg1_10k_256": dict(n=10_000, height=256, width=256, s_min=0.01, s_max=0.08, sh_degree=0),
    "cfg2_100k_800": dict(n=100_000, height=800, width=800, s_min=0.006, s_max=0.05, sh_degree=3),
    # truck-like stand-in: the real scene is not in the container (config/tat_truck_every_8_test.yaml:44-47)
    "cfg3_400k_1080p": dict(n=400_000, height=1072, width=1920, s_min=0.003, s_max=0.03, sh_degree=3,
                            near_plane=0.4, far_plane=2000.0, depth_to_sort_key_scale=10.0),
    "headline_1m_1080p": dict(n=1_000_000, height=1072, width=1920, s_min=0.002, s_max=0.02, sh_degree=3),
    "cfg4_2m_1080p": dict(n=2_000_000, height=1072, width=1920, s_min=0.0015, s_max=0.015, sh_degree=3),
}


@dataclass
class SyntheticScene:
    point_cloud: torch.Tensor            # [N,3] f32
    point_cloud_features: torch.Tensor   # [N,56] f32
    point_invalid_mask: torch.Tensor     # [N] i8
    point_object_id: torch.Tensor        # [N] i32
    camera_intrinsics: torch.Tensor      # [3,3] f32
    q_pointcloud_camera: torch.Tensor    # [1,4] f32 (x,y,z,w)
    t_pointcloud_camera: torch.Tensor    # [1,3] f32
    height: int
    width: int
    near_plane: float = 0.8
    far_plane: float = 1000.0
    depth_to_sort_key_scale: float = 100.0

    def to(self, device):
        kw = {}
        for k, v in self.__dict__.items():
            kw[k] = v.to(device) if isinstance(v, torch.Tensor) else v
        return SyntheticScene(**kw)


def make_scene(n: int, height: int, width: int, s_min: float, s_max: float, sh_degree: int = 3,
               seed: int = 0, near_plane: float = 0.8, far_plane: float = 1000.0,
               depth_to_sort_key_scale: float = 100.0, invalid_fraction: float = 0.0) -> SyntheticScene:
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(n, 3, generator=g) * 2.0 - 1.0
    feat = torch.zeros(n, 56)
    q = torch.randn(n, 4, generator=g)
    feat[:, 0:4] = q / q.norm(dim=1, keepdim=True)
    feat[:, 4:7] = math.log(s_min) + torch.rand(n, 3, generator=g) * (math.log(s_max) - math.log(s_min))
    feat[:, 7] = torch.rand(n, generator=g) * 4.0 - 2.0
    dc = (torch.rand(n, 3, generator=g) * 4.0 - 2.0) / 0.2820948
    hi = torch.randn(n, 3, 15, generator=g) * 0.3
    for ch in range(3):
        feat[:, 8 + 16 * ch] = dc[:, ch]
        if sh_degree > 0:
            feat[:, 9 + 16 * ch: 24 + 16 * ch] = hi[:, ch]
    invalid = torch.zeros(n, dtype=torch.int8)
    if invalid_fraction > 0:
        invalid = (torch.rand(n, generator=g) < invalid_fraction).to(torch.int8)
    K = torch.tensor([[0.75 * width, 0.0, width / 2.0], [0.0, 0.75 * width, height / 2.0], [0.0, 0.0, 1.0]])
    return SyntheticScene(
        point_cloud=xyz.contiguous(), point_cloud_features=feat.contiguous(), point_invalid_mask=invalid,
        point_object_id=torch.zeros(n, dtype=torch.int32), camera_intrinsics=K,
        q_pointcloud_camera=torch.tensor([[0.0, 0.0, 0.0, 1.0]]),
        t_pointcloud_camera=torch.tensor([[0.0, 0.0, -3.0]]),
        height=height, width=width, near_plane=near_plane, far_plane=far_plane,
        depth_to_sort_key_scale=depth_to_sort_key_scale)


def make_reference_stress_scene(seed: int = 0, n: int = 100_000, n_valid: int = 8000, height: int = 1088,
                                width: int = 1920) -> SyntheticScene:
    """The reference's own stress test (tests/GaussianPointCloudRasterisation_test.py:111-150): 1e5 rows of U[0,1)
    data of which only the first 8000 are valid, 1920 x 1088, f = 500, camera 0.5 behind the cloud: every Gaussian
    covers (nearly) every tile -- 4.6e7 (tile, Gaussian) pairs in the reference's binning.  SURVEY 8(d) secondary scene.
    Other sizes keep the distribution and the field of view (focal length and principal point scale with the width):
    a version the CPU oracle can check."""
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(n, 3, generator=g)
    feat = torch.rand(n, 56, generator=g)
    invalid = torch.zeros(n, dtype=torch.int8)
    invalid[n_valid:] = 1
    f = 500.0 * width / 1920.0
    return SyntheticScene(
        point_cloud=xyz, point_cloud_features=feat, point_invalid_mask=invalid,
        point_object_id=torch.zeros(n, dtype=torch.int32),
        camera_intrinsics=torch.tensor([[f, 0.0, width / 2.0], [0.0, f, 540.0 * height / 1088.0], [0.0, 0.0, 1.0]]),
        q_pointcloud_camera=torch.tensor([[0.0, 0.0, 0.0, 1.0]]), t_pointcloud_camera=torch.tensor([[0.0, 0.0, -0.5]]),
        height=height, width=width)


def make_config_scene(name: str, seed: int = 0) -> SyntheticScene:
    if name == "stress_t_ras":
        return make_reference_stress_scene(seed)
    if name == "trained_1080p":   # a scene grown by the repository's own trainer (trained_workload.py; needs a HIP device)
        from .trained_workload import load_or_make
        return load_or_make("trained_1080p")["scene"]
    if name.startswith("custom:"):   # e.g. custom:n=200000,height=960,width=960,s_min=0.005,s_max=0.04 (tuning sweeps)
        kw = dict(item.split("=") for item in name[len("custom:"):].split(","))
        ints = ("n", "height", "width", "sh_degree")
        return make_scene(seed=seed, **{k: (int(v) if k in ints else float(v)) for k, v in kw.items()})
    return make_scene(seed=seed, **CONFIGS[name])


def make_grad_image(height: int, width: int, seed: int = 1) -> torch.Tensor:
    """Fixed seeded upstream gradient dL/dimage ~ U(-1,1), exercises all channels."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(height, width, 3, generator=g) * 2.0 - 1.0
