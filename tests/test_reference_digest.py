"""BASELINE.json configs[1] (1e5 Gaussians, 800 x 800, SH degree 3, forward + backward "vs CPU ref") against the
REFERENCE'S OWN SOURCES run on exactly that scene (tests/golden/make_reference_digest.py: unmodified sources under the
Taichi emulation, sort() patched to sort(stable=True) because 85 % of the scene's keys tie).  The run is stored as a digest --
image and counts of every pixel, the integer hook fields whole, column norms and 4,096 seeded rows of every gradient and
hook field -- and the inputs are pinned by their SHA-256: the test rebuilds them with `make_config_scene` and refuses to
compare if the hash moved.

CPU: the fp32 oracle against the digest (this is the link reference -> oracle at the size the north star names; the link
oracle -> HIP operator at the same scene is tests/test_hip_parity.py::test_operator_cfg2_size_forward_backward)."""
import hashlib
import os

import numpy as np
import pytest

from oracle import gs_oracle as O
from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene, make_grad_image

PATH = os.environ.get("GS_REFERENCE_DIGEST") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                                              "reference_digest_cfg2_100k_800_tied_keys_stable_sort_exp_cr.npz")
# Two runs were made (round 4, 68 minutes each on eight cores); the committed digest is the second.
#  * Emulation with NumPy's fp32 exp (a few ulps: differs from the correctly rounded value on 39 % of the inputs): ids and
#    tile counts identical; 1 of 640,000 pixels decided differently -- its alpha is 5.1e-11 from 1/255, a quarter of an ulp;
#    364 pixels of the frame lie within 1e-8 of a threshold --; image 4.2e-7 off that pixel, 4.5e-6 on it; gradients 9.0e-6
#    (sampled rows), 1.1e-6 (column norms).
#  * Emulation with exp / log correctly rounded (GS_EMU_EXP=cr; glibc's expf, which the oracle calls, is correctly rounded
#    on 99.93 % of the inputs): EVERY one of the 640,000 pixels decided as the reference decides it, image 2.4e-7, depth
#    4.8e-7, gradients 6.6e-7 .. 1.3e-6 (rows), 4e-8 .. 1e-7 (column norms); uv, depth, normalised quaternions, ids, tile
#    counts, affected-pixel counts identical.  So most of the 9e-6 of the first run was the exponential's last bit, not
#    the order of the reference's atomics.
# A digest made with NumPy's exp (GS_REFERENCE_DIGEST=...) is held to the looser bars and may have flipped pixels.
pytestmark = pytest.mark.skipif(not os.path.exists(PATH), reason="digest not generated (tests/golden/make_reference_digest.py, 70 min)")
IMAGE_TOL = 2e-6


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="module")
def run():
    D = np.load(PATH)
    s = make_config_scene(str(D["workload"]))
    g = make_grad_image(s.height, s.width, seed=int(D["grad_seed"]))
    h = hashlib.sha256()
    for t in (s.point_cloud, s.point_cloud_features, s.point_invalid_mask, s.point_object_id, s.camera_intrinsics,
              s.q_pointcloud_camera, s.t_pointcloud_camera, g):
        h.update(np.ascontiguousarray(t.numpy()).tobytes())
    assert h.hexdigest() == str(D["input_sha256"]), "the scene generator no longer produces the inputs of the reference run"
    f = O.forward(s.point_cloud.numpy(), s.point_cloud_features.numpy(), s.point_invalid_mask.numpy(),
                  s.point_object_id.numpy(), s.camera_intrinsics.numpy(), s.q_pointcloud_camera.numpy(),
                  s.t_pointcloud_camera.numpy(), s.height, s.width, near_plane=s.near_plane, far_plane=s.far_plane,
                  depth_to_sort_key_scale=s.depth_to_sort_key_scale, want_margin=True)
    b = O.backward(f, g.numpy(), int(D["band"]))
    return D, f, b


def test_forward_of_baseline_config_2_matches_the_reference_run(run):
    D, f, b = run
    h = b["hook"]
    assert np.array_equal(h["point_id_in_camera_list"], D["hook_point_id"])
    assert np.array_equal(h["num_overlap_tiles"], D["hook_num_overlap_tiles"])
    count_differs = f["count"] != D["count"].astype(f["count"].dtype)
    image_err = np.abs(f["image"] - D["image"]).max(axis=2)
    depth_err = np.abs(f["depth"][::4] - D["depth_every_4th_row"])
    print(f"[parity] reference_digest.forward: pixels={count_differs.size}, count_differs={int(count_differs.sum())}, "
          f"image_linf={float(image_err.max()):.3e}, depth_linf={float(depth_err.max()):.3e}, "
          f"pixels_within_5e-8_of_a_threshold={int((f['margin'] < 5e-8).sum())}, closest={float(f['margin'].min()):.2e}, "
          f"flipped={int((count_differs | (image_err > IMAGE_TOL)).sum())}")
    # the reference's skip / stop decision on every pixel (bars by the exponential the run was made with, see above)
    flipped = count_differs | (image_err > IMAGE_TOL)
    n_flipped = int(flipped.sum())
    if str(D["emulated_exp"]) == "correctly rounded":
        assert n_flipped == 0
    else:
        assert not (flipped & (f["margin"] >= 1e-8)).any() and n_flipped <= 8 and float(image_err.max()) <= 5e-3
    assert float(depth_err.max()) <= 1e-4 * max(1.0, float(np.abs(D["depth_every_4th_row"]).max()))
    assert int(np.abs(h["num_affected_pixels"].astype(np.int64) - D["hook_num_affected_pixels"].astype(np.int64)).sum()) <= \
        8 * n_flipped      # (a pixel stopped one Gaussian later or earlier changes the counts of the Gaussians behind it)


def test_backward_of_baseline_config_2_matches_the_reference_run(run):
    D, f, b = run
    h = b["hook"]
    fields = dict(grad_xyz=b["grad_xyz"], grad_feat=b["grad_feat"], hook_grad_point=h["grad_point_in_camera"],
                  hook_grad_features=h["grad_pointfeatures_in_camera"], hook_grad_viewspace=h["grad_viewspace"],
                  hook_magnitude=h["magnitude_grad_viewspace"].reshape(-1, 1), hook_depth=h["point_depth"].reshape(-1, 1),
                  hook_uv=h["point_uv_in_camera"], features_after_forward=f["feat"])
    worst = {}
    for name, a in fields.items():
        rows, sample, norms = D[f"{name}_rows"], D[f"{name}_sample"], D[f"{name}_column_norms"]
        grad_tol = 1e-5 if str(D["emulated_exp"]) == "correctly rounded" else 5e-5
        tol = grad_tol if "grad" in name or "magnitude" in name else 1e-6
        d_rows = _rel(a[rows], sample)
        mine = np.linalg.norm(a.astype(np.float64), axis=0)
        d_norms = float(np.abs(mine - norms).max() / max(float(norms.max()), 1e-30))
        worst[name] = (d_rows, d_norms)
        assert d_rows <= tol and d_norms <= tol, (name, d_rows, d_norms)
    mag = h["magnitude_grad_viewspace_on_image"]
    d_mag = _rel(mag[::4], D["hook_magnitude_image_every_4th_row"])
    norm_reference = float(D["hook_magnitude_image_norm"])
    assert d_mag <= grad_tol and abs(float(np.linalg.norm(mag.astype(np.float64))) - norm_reference) <= grad_tol * norm_reference
    print("[parity] reference_digest.backward: " + ", ".join(f"{k}: rows {a:.2e} column_norms {c:.2e}" for k, (a, c) in worst.items()) +
          f", magnitude_image {d_mag:.2e}")


@pytest.mark.gpu
def test_hip_operator_matches_the_reference_run_of_baseline_config_2():
    """The north star's sentence literally: the HIP operator's outputs against the reference's own run of BASELINE
    configs[1], within 1e-4 L-inf on EVERY pixel, every pixel's count equal (none of the 1,983 pixels that lie within 5e-8
    of a threshold goes the other way since round 5), integer fields exact, gradients 1e-4."""
    import torch
    from tests.test_reference_operator import _hip_outputs
    D = np.load(PATH)
    s = make_config_scene(str(D["workload"]))
    g = make_grad_image(s.height, s.width, seed=int(D["grad_seed"]))
    cfg = dict(near_plane=s.near_plane, far_plane=s.far_plane, depth_to_sort_key_scale=s.depth_to_sort_key_scale)
    got = _hip_outputs({"grad_image": g.numpy()}, s, cfg, int(D["band"]))
    torch.cuda.synchronize()
    margin = O.forward(s.point_cloud.numpy(), s.point_cloud_features.numpy(), s.point_invalid_mask.numpy(),
                       s.point_object_id.numpy(), s.camera_intrinsics.numpy(), s.q_pointcloud_camera.numpy(),
                       s.t_pointcloud_camera.numpy(), s.height, s.width, want_margin=True, **cfg)["margin"]
    assert np.array_equal(got["hook_point_id"], D["hook_point_id"])
    assert np.array_equal(got["hook_num_overlap_tiles"], D["hook_num_overlap_tiles"])
    image_err = np.abs(got["image"] - D["image"]).max(axis=2)
    flipped = (got["count"] != D["count"].astype(got["count"].dtype)) | (image_err > 1e-4)
    print(f"[parity] reference_digest.hip: flipped={int(flipped.sum())}, image_linf={float(image_err.max()):.3e}, "
          f"image_linf_off_flips={float(image_err[~flipped].max()):.3e}")
    assert int(flipped.sum()) == 0 and float(image_err.max()) <= 1e-4
    assert np.array_equal(got["hook_num_affected_pixels"], D["hook_num_affected_pixels"])
    assert float(np.abs(got["depth"][::4] - D["depth_every_4th_row"]).max()) <= 2e-4
    fields = dict(grad_xyz=got["grad_xyz"], grad_feat=got["grad_feat"], hook_grad_point=got["hook_grad_point"],
                  hook_grad_features=got["hook_grad_features"], hook_grad_viewspace=got["hook_grad_viewspace"],
                  hook_magnitude=got["hook_magnitude"].reshape(-1, 1), hook_depth=got["hook_depth"].reshape(-1, 1),
                  hook_uv=got["hook_uv"], features_after_forward=got["features"])
    for name, a in fields.items():
        tol = 1e-4 if "grad" in name or "magnitude" in name else 1e-6
        d_rows = _rel(a[D[f"{name}_rows"]], D[f"{name}_sample"])
        norms = D[f"{name}_column_norms"]
        d_norms = float(np.abs(np.linalg.norm(a.astype(np.float64), axis=0) - norms).max() / max(float(norms.max()), 1e-30))
        print(f"[parity] reference_digest.hip.{name}: rows={d_rows:.3e}, column_norms={d_norms:.3e}")
        assert d_rows <= tol and d_norms <= tol, (name, d_rows, d_norms)
