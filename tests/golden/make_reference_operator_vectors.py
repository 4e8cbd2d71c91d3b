#!/usr/bin/env python
"""Generates tests/golden/reference_operator_*.npz by EXECUTING THE REFERENCE'S OWN OPERATOR -- the unmodified source
of /root/reference/taichi_3d_gaussian_splatting/{GaussianPointCloudRasterisation,GaussianPoint3D,SphericalHarmonics,
utils,Camera}.py: its seven Taichi kernels, its torch glue, its autograd Function and its backward hook -- on tiny
seeded scenes, with Taichi replaced by the emulation in tests/golden/taichi_emulation.py (fp32 NumPy scalars, the two
tile blend kernels run block by block on 256 OS threads with real barriers).

Run in the build container only (the GPU box has no /root/reference; takes a few minutes):
    GS_EMU_PROCS=8 python tests/golden/make_reference_operator_vectors.py [scene names]

Scenes are chosen so that no two sort keys tie (asserted): torch.sort's tie order is the one thing the reference
leaves undefined (RAS:947), everything else is then a function of the inputs.
"""
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
from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image, make_scene  # noqa: E402

SCENES = {
    # name: (make_scene kwargs, color_max_sh_band, rasteriser config overrides, opacity-logit override or None)
    "a_40pts_32x48_sh3": (dict(n=40, height=32, width=48, s_min=0.05, s_max=0.3, sh_degree=3, seed=26), 3, {}, None),
    "b_70pts_48x32_sh3_band1_invalid": (dict(n=70, height=48, width=32, s_min=0.03, s_max=0.2, sh_degree=3, seed=3,
                                              invalid_fraction=0.15), 1,
                                         dict(near_plane=0.4, far_plane=2000.0, depth_to_sort_key_scale=1000.0), None),
    # sixteen tiles, default planes and depth scale (quantised depths 200..400: a tie-free seed had to be searched)
    "e_70pts_64x64_default_config": (dict(n=70, height=64, width=64, s_min=0.03, s_max=0.15, sh_degree=3, seed=34,
                                           invalid_fraction=0.05), 3, {}, None),
    # two objects with their own rotated poses (point_object_id selects the pose, RAS:56-59,272-275)
    "d_60pts_32x32_two_objects_rotated": (dict(n=60, height=32, width=32, s_min=0.05, s_max=0.25, sh_degree=3, seed=4), 3,
                                           dict(depth_to_sort_key_scale=1000000.0), "two_objects"),
    # many large, nearly opaque Gaussians: pixels saturate (T' < 1e-4, RAS:458-460) and alpha clamps at 0.99
    "c_90pts_32x32_opaque_saturating": (dict(n=90, height=32, width=32, s_min=0.15, s_max=0.5, sh_degree=3, seed=2), 2,
                                         dict(depth_to_sort_key_scale=100000.0), 6.0),
    # MULTI-BATCH staging: 600 large, faint Gaussians over two tiles -> > 512 entries in one tile, so the forward's
    # group loop (RAS:382-386) and the backward's block loop with its clamp (RAS:574-585) run 3 batches of 256
    "f_600pts_16x32_three_batches": (dict(n=600, height=16, width=32, s_min=0.2, s_max=0.8, sh_degree=3, seed=11), 3,
                                      dict(depth_to_sort_key_scale=1000000.0), -3.0),
    # 128 x 128 = 64 tiles, Gaussians of mixed size, a few invalid rows
    "g_160pts_128x128": (dict(n=160, height=128, width=128, s_min=0.02, s_max=0.2, sh_degree=3, seed=7,
                              invalid_fraction=0.05), 2, dict(depth_to_sort_key_scale=100000.0), None),
    # TIED sort keys (default depth scale, 1/100 units): the reference's torch.sort leaves their order undefined
    # (RAS:947), so THIS scene is generated with that one call patched to sort(stable=True) -- the tie rule of the
    # oracle and of the HIP path.  Shows that the tie rule is the only difference on scenes with ties.
    "h_200pts_32x32_tied_keys_stable_sort": (dict(n=200, height=32, width=32, s_min=0.05, s_max=0.3, sh_degree=3, seed=5),
                                             3, dict(depth_to_sort_key_scale=20.0), None),
    # NOT A TOY (round 4): 2,400 Gaussians over a 320 x 320 image = 400 tiles (about 250 of them non-empty, 5,869 list
    # entries, a few invalid rows); tie-free at a depth scale of 1e7.  90 s with GS_EMU_PROCS=8 (hours before the
    # emulation ran the per-point loop of the backward kernel once instead of once per pixel thread).
    "i_2400pts_320x320_400_tiles": (dict(n=2400, height=320, width=320, s_min=0.005, s_max=0.03, sh_degree=3, seed=41,
                                         invalid_fraction=0.03), 3, dict(depth_to_sort_key_scale=1.0e7), None),
    # DEPTH COMPLEXITY at scale (round 4): 6,000 Gaussians of opacity 0.73 over 384 x 384 = 576 tiles, 89,765 list entries,
    # 142 tiles with more than 256 entries (the longest 614: three batches of the reference's staging), 42 % of the
    # pixels stop at T' < 1e-4 (RAS:458-460), up to 100 blended Gaussians per pixel; tie-free at a depth scale of 1e7
    "j_6000pts_384x384_deep_lists": (dict(n=6000, height=384, width=384, s_min=0.02, s_max=0.12, sh_degree=3, seed=43,
                                          invalid_fraction=0.02), 3, dict(depth_to_sort_key_scale=1.0e7), 1.0),
    # BASELINE.json configs[0] AS STATED (round 4): 10,000 Gaussians, 256 x 256, SH-degree-0 data, colour band 0, default
    # planes and depth scale -- the kwargs are synthetic.CONFIGS["cfg1_10k_256"] with seed 0, i.e. the scene of
    # bench.py --workload cfg1_10k_256 and of test_operator_cfg1_as_stated.  48,317 list entries, lists of up to 787
    # (four batches); 64 % of the keys tie at the default scale, so -- as for vector h -- the reference's sort() call is
    # patched to sort(stable=True), the one thing it leaves undefined.
    "k_cfg1_10k_256x256_sh0_tied_keys_stable_sort": (dict(n=10_000, height=256, width=256, s_min=0.01, s_max=0.08,
                                                          sh_degree=0, seed=0), 0, {}, None),
    # DRAWS OF THE RANDOMISED PARITY SUITE (round 4; tests/test_fuzz_gpu.py:random_scene -- rotated off-centre cameras with
    # unequal focal lengths, several objects under their own poses, un-normalised quaternions, invalid rows, needles:
    # Gaussians with one axis 10-100 x the others, whose conics fp32 barely resolves).  Planes, depth scale and band are
    # the draw's; their keys tie (scales 10 .. 1000), hence the stable-sort patch.
    #   draw 17: 144 x 192, 2,182 rows (half of them invalid), 2 objects, needles, near 2 / far 10, 32,768 list entries
    "l_fuzz17_144x192_needles_two_objects_tied_keys_stable_sort": ("fuzz:17", None, None, None),
    #   draw 20: 144 x 192, 1,744 rows, needles seen from as close as 0.1 (a hundredfold magnification), lists of up to 721
    "m_fuzz20_144x192_needles_close_ups_tied_keys_stable_sort": ("fuzz:20", None, None, None),
    #   draw 9: 240 x 48, 4,454 rows, 3 objects, 42,968 list entries in 45 tiles: lists of up to 2,189 = nine batches
    "n_fuzz9_240x48_three_objects_nine_batches_tied_keys_stable_sort": ("fuzz:9", None, None, None),
    # THE REFERENCE'S OWN STRESS DISTRIBUTION (T_RAS:111-150: rows of U[0,1) data, camera 0.5 behind the cloud, every
    # Gaussian over nearly every tile) at 256 x 144 with 300 valid rows of 600 (synthetic.make_reference_stress_scene)
    "o_stress_distribution_256x144_tied_keys_stable_sort": ("stress:600:300:144:256", 3, {}, None),
    # THE TRUCK CONFIGURATION'S RASTERISER PARAMETERS (round 5; config/tat_truck_every_8_test.yaml:44-47: near 0.4, far 2000,
    # depth-to-sort-key scale 10 -- quantised depths of 20..40 over the whole cloud, so nearly every key of a tile ties):
    # 6,000 Gaussians over 256 x 256 = 256 tiles, stable-sort patch
    "p_truck_params_6000pts_256x256_tied_keys_stable_sort": (
        dict(n=6000, height=256, width=256, s_min=0.01, s_max=0.06, sh_degree=3, seed=47, invalid_fraction=0.02), 3,
        dict(near_plane=0.4, far_plane=2000.0, depth_to_sort_key_scale=10.0), None),
}


def build_scene(spec):
    """-> (scene, band, config kwargs, kwargs recorded in the archive) for a string-valued SCENES entry."""
    kind, *args = spec.split(":")
    if kind == "fuzz":
        sys.path.insert(0, ROOT)
        from tests.test_fuzz_gpu import random_scene
        scene, band, _needles, _options = random_scene(int(args[0]))
        cfg = dict(near_plane=float(scene.near_plane), far_plane=float(scene.far_plane),
                   depth_to_sort_key_scale=float(scene.depth_to_sort_key_scale))
        return scene, band, cfg, dict(height=scene.height, width=scene.width, builder=spec)
    if kind == "stress":
        from taichi_3d_gaussian_splatting_amd.synthetic import make_reference_stress_scene
        n, n_valid, height, width = (int(a) for a in args)
        scene = make_reference_stress_scene(0, n=n, n_valid=n_valid, height=height, width=width)
        return scene, None, None, dict(height=height, width=width, builder=spec)
    raise ValueError(spec)
# forward outputs and integer fields of a regenerated archive must be bit-identical to the committed one (the gradients
# are sums of fp32 atomic adds in OS-thread order: reproducible to ~1e-6 only, as on a GPU)
BITWISE_FIELDS = ("image", "depth", "count", "features_after_forward", "hook_point_id", "hook_num_overlap_tiles",
                  "hook_num_affected_pixels", "hook_depth", "hook_uv")
STABLE_SORT_PATCH = ("point_in_camera_sort_key.sort()", "point_in_camera_sort_key.sort(stable=True)")


def main():
    from oracle import gs_oracle as O   # only to assert that the scene has no sort-key ties
    only = sys.argv[1:]
    for name, (kw, band, cfg_kw, opacity) in SCENES.items():
        if only and name not in only:
            continue
        tied = "tied_keys_stable_sort" in name
        mods = E.load_reference("/root/reference", source_patches={
            "GaussianPointCloudRasterisation": [STABLE_SORT_PATCH]} if tied else None)
        RAS, CAM = mods["GaussianPointCloudRasterisation"], mods["Camera"]
        Op = RAS.GaussianPointCloudRasterisation
        if isinstance(kw, str):
            s, draw_band, draw_cfg, kw = build_scene(kw)
            band = draw_band if band is None else band
            cfg_kw = draw_cfg if cfg_kw is None else cfg_kw
        else:
            s = make_scene(**kw)
        if opacity == "two_objects":
            gq = torch.Generator().manual_seed(99)
            q = torch.tensor([[0.0, 0.0, 0.0, 1.0]]).repeat(2, 1) + 0.15 * torch.randn(2, 4, generator=gq)
            s.q_pointcloud_camera = q / q.norm(dim=1, keepdim=True)
            s.t_pointcloud_camera = torch.tensor([[0.0, 0.0, -3.0]]).repeat(2, 1) + 0.2 * torch.randn(2, 3, generator=gq)
            s.point_object_id = torch.randint(0, 2, (kw["n"],), generator=gq, dtype=torch.int32)
            opacity = None
        if opacity is not None:
            s.point_cloud_features[:, 7] = opacity
        f = O.forward(s.point_cloud.numpy(), s.point_cloud_features.numpy(), s.point_invalid_mask.numpy(),
                      s.point_object_id.numpy(), s.camera_intrinsics.numpy(), s.q_pointcloud_camera.numpy(),
                      s.t_pointcloud_camera.numpy(), s.height, s.width, **cfg_kw)
        ties = float((f["keys"][1:] == f["keys"][:-1]).mean())
        assert tied or ties == 0, f"{name}: tied sort keys, pick another seed"
        assert not tied or ties > 0.02, f"{name}: wanted tied keys"
        per_tile = f["tile_end"] - f["tile_start"]
        print(name, "K", len(f["keys"]), "longest tile list", int(per_tile.max()), "tie fraction", ties,
              "saturated pixels", float((f["acc_alpha"] > 0.9999).mean()), flush=True)
        xyz = s.point_cloud.clone().requires_grad_(True)
        feat = s.point_cloud_features.clone().requires_grad_(True)
        hook = {}
        op = Op(Op.GaussianPointCloudRasterisationConfig(**cfg_kw), backward_valid_point_hook=lambda h: hook.update(h=h))
        t0 = time.time()
        image, depth, count = op(Op.GaussianPointCloudRasterisationInput(
            point_cloud=xyz, point_cloud_features=feat, point_object_id=s.point_object_id,
            point_invalid_mask=s.point_invalid_mask,
            camera_info=CAM.CameraInfo(camera_intrinsics=s.camera_intrinsics, camera_height=s.height,
                                       camera_width=s.width, camera_id=0),
            q_pointcloud_camera=s.q_pointcloud_camera, t_pointcloud_camera=s.t_pointcloud_camera,
            color_max_sh_band=band))
        t1 = time.time()
        g = make_grad_image(s.height, s.width, seed=31)
        (image * g).sum().backward()
        t2 = time.time()
        h = hook["h"]
        out = dict(
            kwargs=np.array(repr(kw)), band=np.array(band), stable_sort_patch=np.array(int(tied)), config=np.array(repr(cfg_kw)), grad_image=g.numpy(),
            opacity_override=np.array(np.nan if opacity is None else opacity),
            in_xyz=s.point_cloud.numpy(), in_feat=s.point_cloud_features.numpy(), in_invalid=s.point_invalid_mask.numpy(),
            in_object_id=s.point_object_id.numpy(), in_K=s.camera_intrinsics.numpy(), in_q=s.q_pointcloud_camera.numpy(),
            in_t=s.t_pointcloud_camera.numpy(),
            features_after_forward=feat.detach().numpy().copy(),      # q normalised in place (RAS:196-205)
            image=image.detach().numpy(), depth=depth.detach().numpy(), count=count.numpy(),
            grad_xyz=xyz.grad.numpy(), grad_feat=feat.grad.numpy(),
            hook_point_id=h.point_id_in_camera_list.numpy(), hook_grad_point=h.grad_point_in_camera.numpy(),
            hook_grad_features=h.grad_pointfeatures_in_camera.numpy(), hook_grad_viewspace=h.grad_viewspace.numpy(),
            hook_magnitude=h.magnitude_grad_viewspace.numpy(),
            hook_magnitude_image=h.magnitude_grad_viewspace_on_image.numpy(),
            hook_num_overlap_tiles=h.num_overlap_tiles.numpy(), hook_num_affected_pixels=h.num_affected_pixels.numpy(),
            hook_depth=h.point_depth.numpy(), hook_uv=h.point_uv_in_camera.numpy())
        # GS_EMU_EXP=cr (taichi_emulation.py: exp / log correctly rounded instead of NumPy's fp32 routines) writes a second
        # archive beside the default one: the pair brackets what the choice of the exponential does to the reference
        exp_cr = os.environ.get("GS_EMU_EXP") == "cr"
        out["emulated_exp"] = np.array("correctly rounded" if exp_cr else "numpy fp32")
        path = os.path.join(HERE, f"reference_operator_{name}{'_exp_cr' if exp_cr else ''}.npz")
        # GS_EMU_OUT_DIR (make_arithmetic_residue.py): another arithmetic of the emulation (GS_EMU_FMA / GS_EMU_RCP_DIV /
        # NumPy's exp) writes elsewhere and is not compared with the committed archive here
        if os.environ.get("GS_EMU_OUT_DIR"):
            out["emulated_arithmetic"] = np.array(f"fma={int(E.EMU_FMA)} rcp_div={int(E.EMU_RCP_DIV)} exp={'cr' if exp_cr else 'numpy'}")
            path = os.path.join(os.environ["GS_EMU_OUT_DIR"], os.path.basename(path))
        elif E.EMU_FMA or E.EMU_RCP_DIV:
            raise SystemExit("GS_EMU_FMA / GS_EMU_RCP_DIV change the reference's arithmetic: set GS_EMU_OUT_DIR (the committed "
                             "archives are the IEEE run)")
        if os.path.exists(path):
            old = np.load(path)
            for key in BITWISE_FIELDS:
                assert np.array_equal(old[key], out[key]), f"{name}: {key} differs from the committed archive"
            for key in ("grad_xyz", "grad_feat"):
                d = np.abs(old[key] - out[key]).max()
                assert d <= 1e-4 * max(1.0, np.abs(old[key]).max()), (name, key, d)
            print(f"{name}: forward and integer fields bit-identical to the committed archive")
        np.savez_compressed(path, **out)
        print(f"{name}: forward {t1 - t0:.1f} s, backward {t2 - t1:.1f} s, M={len(out['hook_point_id'])}, "
              f"mean image {out['image'].mean():.4f}, max count {out['count'].max()}")


if __name__ == "__main__":
    main()
