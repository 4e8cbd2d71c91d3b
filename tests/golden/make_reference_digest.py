#!/usr/bin/env python
"""BASELINE.json configs[1] -- 1e5 Gaussians, 800 x 800, SH degree 3, forward + backward, "vs CPU ref" -- THROUGH THE
REFERENCE'S OWN SOURCES (unmodified, under tests/golden/taichi_emulation.py, like make_reference_operator_vectors.py),
stored as a DIGEST: the scene is too large for a full archive (its inputs alone are 24 MB).

Inputs are `synthetic.make_config_scene("cfg2_100k_800")` (seed 0) and are pinned by their SHA-256, which the digest
carries and the test re-computes.  Stored of the reference's outputs: the image (f32, every pixel), the per-pixel count
(u8), every fourth row of the depth image, the integer hook fields whole (ids, tile counts, affected-pixel counts), the L2
norm of every column of every gradient / hook field, and 4,096 seeded rows of each of them in full.
64 % of this scene's sort keys tie at the default depth scale, so -- as for vectors h and k -- the reference's one sort()
call is patched to sort(stable=True).

    GS_EMU_PROCS=8 python tests/golden/make_reference_digest.py        (about 2.5 hours on eight cores)
"""
import hashlib
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
os.environ.setdefault("GS_EMU_EXP", "cr")   # one definition of exp on every side: correctly rounded (tests/golden/README.md)
import taichi_emulation as E  # noqa: E402
from make_reference_operator_vectors import STABLE_SORT_PATCH  # noqa: E402
from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene, make_grad_image  # noqa: E402

WORKLOAD, BAND, GRAD_SEED, ROW_SEED, N_ROWS = "cfg2_100k_800", 3, 31, 7, 4096
# (GS_EMU_EXP=cr -- exp / log of the emulation correctly rounded -- writes a second file beside the committed one: the pair
# shows what the exponential's last bit does to the frame, see tests/test_reference_digest.py)
OUT = os.path.join(HERE, "reference_digest_cfg2_100k_800_tied_keys_stable_sort%s.npz" %
                   ("_exp_cr" if os.environ.get("GS_EMU_EXP") == "cr" else ""))


def input_hash(s, g) -> str:
    h = hashlib.sha256()
    for t in (s.point_cloud, s.point_cloud_features, s.point_invalid_mask, s.point_object_id, s.camera_intrinsics,
              s.q_pointcloud_camera, s.t_pointcloud_camera, g):
        h.update(np.ascontiguousarray(t.numpy()).tobytes())
    return h.hexdigest()


def sample_rows(m: int) -> np.ndarray:
    return np.sort(np.random.default_rng(ROW_SEED).choice(m, size=min(N_ROWS, m), replace=False))


def main():
    mods = E.load_reference("/root/reference", source_patches={"GaussianPointCloudRasterisation": [STABLE_SORT_PATCH]})
    RAS, CAM = mods["GaussianPointCloudRasterisation"], mods["Camera"]
    Op = RAS.GaussianPointCloudRasterisation
    s = make_config_scene(WORKLOAD)
    g = make_grad_image(s.height, s.width, seed=GRAD_SEED)
    digest = input_hash(s, g)
    xyz = s.point_cloud.clone().requires_grad_(True)
    feat = s.point_cloud_features.clone().requires_grad_(True)
    hook = {}
    op = Op(Op.GaussianPointCloudRasterisationConfig(near_plane=s.near_plane, far_plane=s.far_plane,
                                                     depth_to_sort_key_scale=s.depth_to_sort_key_scale),
            backward_valid_point_hook=lambda h: hook.update(h=h))
    t0 = time.time()
    image, depth, count = op(Op.GaussianPointCloudRasterisationInput(
        point_cloud=xyz, point_cloud_features=feat, point_object_id=s.point_object_id,
        point_invalid_mask=s.point_invalid_mask,
        camera_info=CAM.CameraInfo(camera_intrinsics=s.camera_intrinsics, camera_height=s.height, camera_width=s.width,
                                   camera_id=0),
        q_pointcloud_camera=s.q_pointcloud_camera, t_pointcloud_camera=s.t_pointcloud_camera, color_max_sh_band=BAND))
    t1 = time.time()
    print(f"forward {t1 - t0:.0f} s", flush=True)
    (image * g).sum().backward()
    t2 = time.time()
    print(f"backward {t2 - t1:.0f} s", flush=True)
    h = hook["h"]
    assert int(count.max()) < 256
    fields = dict(grad_xyz=xyz.grad.numpy(), grad_feat=feat.grad.numpy(), hook_grad_point=h.grad_point_in_camera.numpy(),
                  hook_grad_features=h.grad_pointfeatures_in_camera.numpy(), hook_grad_viewspace=h.grad_viewspace.numpy(),
                  hook_magnitude=h.magnitude_grad_viewspace.numpy().reshape(-1, 1),
                  hook_depth=h.point_depth.numpy().reshape(-1, 1), hook_uv=h.point_uv_in_camera.numpy(),
                  features_after_forward=feat.detach().numpy())
    out = dict(workload=np.array(WORKLOAD), band=np.array(BAND), grad_seed=np.array(GRAD_SEED), input_sha256=np.array(digest),
               stable_sort_patch=np.array(1), emulated_exp=np.array("numpy fp32" if os.environ.get("GS_EMU_EXP") != "cr" else "correctly rounded"),
               seconds=np.array([t1 - t0, t2 - t1]),
               image=image.detach().numpy(), count=count.numpy().astype(np.uint8), depth_every_4th_row=depth.detach().numpy()[::4],
               hook_point_id=h.point_id_in_camera_list.numpy(), hook_num_overlap_tiles=h.num_overlap_tiles.numpy(),
               hook_num_affected_pixels=h.num_affected_pixels.numpy(),
               hook_magnitude_image_norm=np.array(np.linalg.norm(h.magnitude_grad_viewspace_on_image.numpy().astype(np.float64))),
               hook_magnitude_image_every_4th_row=h.magnitude_grad_viewspace_on_image.numpy()[::4])
    for name, a in fields.items():
        rows = sample_rows(a.shape[0])
        out[f"{name}_column_norms"] = np.linalg.norm(a.astype(np.float64), axis=0)
        out[f"{name}_rows"], out[f"{name}_sample"] = rows, a[rows]
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {os.path.getsize(OUT) / 1e6:.1f} MB, M={len(out['hook_point_id'])}, mean image {out['image'].mean():.4f}, "
          f"max count {out['count'].max()}")


if __name__ == "__main__":
    main()
