"""The TRAINED scene at full size (trained_workload.py: a cloud grown tenfold by the repository's own trainer and adaptive
controller at 1920 x 1072 -- densified, anisotropic, with an opacity-reset history; the stand-in for BASELINE configs 3 and
5 while the Truck data is absent) against the CPU oracle: forward and backward, masked and all-pixel figures, and the
operator's own choices (list layout, cull, speculation) on it."""
import numpy as np
import pytest
import torch

from oracle import gs_oracle as O
from tests.helpers import FRAGILE_MARGIN, oracle_forward, rel_l2, report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def trained():
    assert torch.cuda.is_available()
    from taichi_3d_gaussian_splatting_amd.trained_workload import load_or_make
    made = load_or_make("trained_1080p")
    report("trained_scene.stats", **{k: v for k, v in made["stats"].items() if k != "growth"})
    return made


def test_trained_scene_is_what_it_says(trained):
    st = trained["stats"]
    assert st["live_gaussians"] >= 300_000 and st["densifications"] >= 20 and st["iterations"] > 3000   # past the opacity reset
    assert st["anisotropy_p99"] > 5.0
    s = trained["scene"]
    assert (s.height, s.width) == (1072, 1920)


def test_trained_scene_against_the_oracle(trained):
    from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Op
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image
    s = trained["scene"]
    f = oracle_forward(s)
    keys = f["keys"]
    per_tile = f["tile_end"] - f["tile_start"]
    report("trained_scene.sizes", N=s.point_cloud.shape[0], M=len(f["ids"]), K_reference_binning=len(keys),
           longest_tile_list=int(per_tile.max()), mean_tile_list=float(per_tile.mean()),
           saturated_pixel_fraction=float((f["acc_alpha"] > 0.9999).mean()), mean_acc_alpha=float(f["acc_alpha"].mean()))
    fragile = f["margin"] < FRAGILE_MARGIN
    g = make_grad_image(s.height, s.width)
    g_masked = g * torch.from_numpy(~fragile)[:, :, None]
    d = s.to("cuda")
    hooks = []
    op = Op(Op.GaussianPointCloudRasterisationConfig(near_plane=s.near_plane, far_plane=s.far_plane,
                                                     depth_to_sort_key_scale=s.depth_to_sort_key_scale),
            backward_valid_point_hook=hooks.append)
    cam = CameraInfo(camera_intrinsics=d.camera_intrinsics, camera_height=s.height, camera_width=s.width, camera_id=0)

    def run(gi):
        xyz = d.point_cloud.clone().requires_grad_(True)
        feat = d.point_cloud_features.clone().requires_grad_(True)
        image, depth, count = op(Op.GaussianPointCloudRasterisationInput(
            point_cloud=xyz, point_cloud_features=feat, point_object_id=d.point_object_id,
            point_invalid_mask=d.point_invalid_mask, camera_info=cam, q_pointcloud_camera=d.q_pointcloud_camera,
            t_pointcloud_camera=d.t_pointcloud_camera, color_max_sh_band=3))
        image.backward(gi.to("cuda"))
        return image.detach(), depth.detach(), count, xyz.grad, feat.grad

    frames = [run(g_masked) for _ in range(3)]      # the third frame runs with the layout and capacities the operator chose
    image, depth, count, gx, gf = frames[-1]
    for other in frames[:-1]:                        # whatever the frame's history: the same bits
        assert torch.equal(other[0], image) and torch.equal(other[2], count)
    diff = np.abs(image.cpu().numpy().astype(np.float64) - f["image"].astype(np.float64)).max(axis=2)
    ok = ~fragile
    report("trained_scene.image", linf_nonfragile=float(diff[ok].max()), linf_all_pixels=float(diff.max()),
           fragile_fraction=float(fragile.mean()), over_1e4_all_pixels=int((diff > 1e-4).sum()),
           bin_shift=op.list_layout(s.height, s.width).bin_shift, exact_tile_cull=op.exact_tile_cull,
           speculation=dict(op.speculation_stats))
    assert diff[ok].max() <= 1e-4                    # north star, off the pixels whose decisions sit on a threshold
    assert diff.max() <= 1e-2                        # one flipped Gaussian at most
    assert np.array_equal(count.cpu().numpy()[ok], f["count"][ok])
    ob = O.backward(f, g_masked.numpy(), 3)
    r_feat, r_xyz = rel_l2(gf.cpu().numpy(), ob["grad_feat"]), rel_l2(gx.cpu().numpy(), ob["grad_xyz"])
    image_a, _, _, gx_a, gf_a = run(g)
    ob_a = O.backward(f, g.numpy(), 3)
    r_feat_a, r_xyz_a = rel_l2(gf_a.cpu().numpy(), ob_a["grad_feat"]), rel_l2(gx_a.cpu().numpy(), ob_a["grad_xyz"])
    report("trained_scene.gradients", masked_rel_l2_feat=r_feat, masked_rel_l2_xyz=r_xyz, all_pixels_rel_l2_feat=r_feat_a,
           all_pixels_rel_l2_xyz=r_xyz_a)
    assert r_feat <= 1e-4 and r_xyz <= 1e-4          # fuzz bars (tests/test_fuzz_gpu.py): masked
    assert r_feat_a <= 1e-3 and r_xyz_a <= 1e-3      # all pixels: a flipped pair is a discrete change
    h = hooks[-1]
    assert np.array_equal(h.point_id_in_camera_list.cpu().numpy(), f["ids"])
    assert np.array_equal(h.num_overlap_tiles.cpu().numpy(), f["num_overlap_tiles"])


def test_the_operator_picks_the_backward_form_by_the_walk_lengths(trained):
    """frame_path.walk_skew: on the trained scene (a few tiles walk ten times the mean) the operator's sample of the forward's
    recorded walk lengths arrives within a few frames and selects the register form of the two-wave backward kernel
    (GS_BLEND_SKEWED_WALKS); on the evenly loaded headline-like scene it does not.  The two forms take the same decisions
    and agree to rounding: the gradients of a frame rendered before and after the switch are the same to 1e-6."""
    from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Op
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image, make_scene

    def run(scene, frames, by_skew=True):
        s = scene.to("cuda")
        g = make_grad_image(s.height, s.width).cuda()
        op = Op(Op.GaussianPointCloudRasterisationConfig(near_plane=s.near_plane, far_plane=s.far_plane,
                                                         depth_to_sort_key_scale=s.depth_to_sort_key_scale))
        op.backward_form_by_walk_skew = by_skew
        grads = []
        for _ in range(frames):
            xyz = s.point_cloud.clone().requires_grad_(True)
            feat = s.point_cloud_features.clone().requires_grad_(True)
            inp = Op.GaussianPointCloudRasterisationInput(
                point_cloud=xyz, point_cloud_features=feat, point_object_id=s.point_object_id,
                point_invalid_mask=s.point_invalid_mask, camera_info=CameraInfo(s.camera_intrinsics, s.height, s.width, 0),
                q_pointcloud_camera=s.q_pointcloud_camera, t_pointcloud_camera=s.t_pointcloud_camera, color_max_sh_band=3)
            image, _, _ = op(inp)
            (image * g).sum().backward()
            torch.cuda.synchronize()
            grads.append(feat.grad.clone())
        probes = op.__dict__.get("_walk_skew", {})
        return grads, [p["skewed"] for p in probes.values()]

    grads, skewed = run(trained["scene"], 5)
    assert skewed == [True]
    plain, none = run(trained["scene"], 2, by_skew=False)
    assert none == []                                            # (switched off: no probe at all)
    rel = rel_l2(grads[-1].cpu().numpy(), plain[-1].cpu().numpy())
    report("trained_scene.backward_forms", rel_l2_of_feature_gradients=rel)
    assert rel < 1e-6
    even = make_scene(n=200_000, height=1072, width=1920, s_min=0.004, s_max=0.02, seed=11)
    _, skewed = run(even, 4)
    assert skewed == [False]
