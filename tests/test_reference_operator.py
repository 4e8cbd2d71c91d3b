"""Oracle (CPU) and HIP path (GPU) against outputs of THE REFERENCE'S OWN OPERATOR.

tests/golden/reference_operator_*.npz were produced by executing the unmodified reference sources (its Taichi
kernels, torch glue, autograd Function and backward hook) under the Taichi emulation of tests/golden/
taichi_emulation.py -- see tests/golden/make_reference_operator_vectors.py.  The scenes either have no tied sort keys,
the only thing the reference leaves undefined, or were generated with its one sort() call patched to sort(stable=True)
(their names say so); every output is then a function of the inputs and is compared here: image, depth, per-pixel
counts, the in-place normalised features, dense gradients and all ten hook fields.

ONE DEFINITION OF exp (round 5).  Every archive is `*_exp_cr.npz`: generated with GS_EMU_EXP=cr, i.e. the emulation's exp
and log evaluated in double and rounded once -- the correctly rounded fp32 functions.  The oracle's fp32 build does the
same (oracle/gs_oracle.c R_EXP), and so does the HIP library wherever an exponential decides something discrete (scale
activation, opacity sigmoid, the exact re-evaluation of a weight next to a threshold: csrc/gs_common.h).  The archives
made with NumPy's fp32 exp (whose last bit differs from the correctly rounded value on 39 % of the inputs) are gone.

Tolerances.  Observed: fp32 oracle vs reference image L-inf 1.2e-7 .. 4.9e-7, every discrete output identical on every
vector (per-pixel counts included), gradients 2e-7 .. 1e-6 relative L2 on the small vectors and up to 2e-5 on the large
ones (_grad_tol).  The bars: image 2e-6 for the oracle and the north star's 1e-4 for the HIP path ON EVERY PIXEL -- since
round 5 no pixel is admitted as "fragile" for either: both take the reference's skip (RAS:451 / RAS:631) and stop (RAS:458)
decision on every pixel; discrete outputs (visible ids, tile counts, per-pixel counts, affected-pixel counts) identical;
gradients 2e-5 / 5e-5 relative L2 -- the reference accumulates with fp32 atomics (emulated in thread order), the oracle in
double, the HIP path in a fixed fp32 order.

Round 2 added: ``f_600pts_16x32_three_batches`` (598 entries in one tile: the reference's 256-entry shared-memory staging
runs three batches forward and backward, incl. the clamp of RAS:579-585), ``g_160pts_128x128`` (64 tiles) and
``h_200pts_32x32_tied_keys_stable_sort`` (69 % tied keys; generated with the reference's ``sort()`` call patched to
``sort(stable=True)`` -- the tie rule is the only thing the reference leaves open).

Round 4 added three vectors that are not toys (the emulation now runs a kernel's top-level loops as the separate offloads
they are, and deals tiles to worker processes): ``i_2400pts_320x320_400_tiles``; ``j_6000pts_384x384_deep_lists`` (89,765
list entries, lists of up to 614, 42 % of the pixels stop at T' < 1e-4); and ``k_cfg1_10k_256x256_sh0_tied_keys_stable_sort``
-- BASELINE.json's configs[0] exactly as stated (the scene of ``bench.py --workload cfg1_10k_256``), 64 % tied keys, the
stable-sort patch of vector h.  The fp32 oracle takes the reference's skip / stop decision on EVERY pixel of all of them.
Then four more: ``n`` (draw 9 of tests/test_fuzz_gpu.py: three posed objects, lists of up to 2,189), ``o`` (the reference's own
stress distribution, T_RAS:111-150) and the needle scenes ``l``, ``m`` with tests of their own at the end of this file.
BASELINE configs[1] is in tests/test_reference_digest.py.
"""
import ast
import glob
import os

import numpy as np
import pytest
import torch

from oracle import gs_oracle as O
from taichi_3d_gaussian_splatting_amd.synthetic import SyntheticScene   # (a plain container for the archived inputs)

ALL_FILES = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_operator_*_exp_cr.npz")))
# Needle scenes (Gaussians with one axis 10-100 x the others) are ill-conditioned in fp32 and have tests of their own below
NEEDLE_FILES = [p for p in ALL_FILES if "needles" in os.path.basename(p)]
FILES = [p for p in ALL_FILES if p not in NEEDLE_FILES]
NEEDLE_MARGIN, SPEC_FACTOR = 4e-5, 4.0     # as tests/test_fuzz_gpu.py
IMAGE_TOL_ORACLE, IMAGE_TOL_F64_SPEC, IMAGE_TOL_HIP = 2e-6, 2e-5, 1e-4
MAX_FLIPPED_PIXELS, FLIP_GRAD_TOL = 8, 2e-4    # (the f64 SPEC build only: it rounds differently by design; fp32 oracle and HIP: none)


def _load(path):
    """Inputs come from the archive itself (every archive carries its `in_*` fields since round 4; the generator asserts
    that regenerated forward outputs are bit-identical to the committed ones): nothing of the product builds them."""
    V = np.load(path)
    kw = ast.literal_eval(str(V["kwargs"]))
    s = SyntheticScene(
        point_cloud=torch.from_numpy(V["in_xyz"]), point_cloud_features=torch.from_numpy(V["in_feat"]),
        point_invalid_mask=torch.from_numpy(V["in_invalid"]), point_object_id=torch.from_numpy(V["in_object_id"]),
        camera_intrinsics=torch.from_numpy(V["in_K"]), q_pointcloud_camera=torch.from_numpy(V["in_q"]),
        t_pointcloud_camera=torch.from_numpy(V["in_t"]), height=int(kw["height"]), width=int(kw["width"]))
    return V, s, ast.literal_eval(str(V["config"])), int(V["band"])


def _rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) /
                 max(np.linalg.norm(np.asarray(b, np.float64)), 1e-30))


def _grad_tol(V):
    """Relative-L2 bar of the gradients against a reference-run vector.  The reference adds its per-Gaussian sums with
    fp32 atomics (emulated in thread order: an undefined order on a GPU too), the oracle in double, the HIP path in a
    fixed fp32 tree -- so the distance is the REFERENCE's own fp32 summation noise and grows with the terms per sum.  All
    vectors are made with the correctly rounded fp32 exp (GS_EMU_EXP=cr, the definition the oracle and the kernels share:
    tests/golden/README.md; with NumPy's fp32 exp, whose last bit differs on 39 % of the inputs, the same distances were
    3-10x larger and those archives are gone).  Observed: 2e-7 .. 1e-6 on the vectors whose Gaussians touch up to 1,129
    pixels (bar 2e-5); 7e-7 .. 8e-6 on j (up to 8,274 pixels per Gaussian), k (BASELINE config 1: 1,550), n (6,223) and o
    (the reference's stress distribution: every Gaussian over all 36,864 pixels, |grad uv| sums of 36,864 terms) -- bar 5e-5
    from 1,500 pixels per Gaussian on."""
    return 2e-5 if int(V["hook_num_affected_pixels"].max()) < 1500 else 5e-5


def _check(V, got, grad_tol, image_tol, fragile=None):
    """got: dict with image, depth, count, features, grad_xyz, grad_feat and the ten hook fields.
    fragile (f64 spec build only; None = none admitted: the fp32 oracle and the HIP path): bool[H, W], pixels on which the CPU
    build evaluated an alpha or a T' within 1e-5 of its threshold (RAS:451, RAS:458) -- the admission rule of
    tests/test_fuzz_gpu.py: two correct implementations may decide such a pixel differently (seen once at a margin of
    1.3e-7 in 1,800 random frames).  A pixel that misses a bar must be one of them, there may be at
    most MAX_FLIPPED_PIXELS, and they stay within one blended Gaussian (5e-3); everything else holds on every pixel.
    Observed: the fp32 oracle takes the reference's decision on EVERY pixel of every vector -- also on the 296 pixels of
    vector j that come within 5e-8 of a threshold (the closest: 3.6e-11); the f64 build flips four pixels of j (margins
    2e-10 .. 2e-8) and none elsewhere."""
    assert np.array_equal(got["hook_point_id"], V["hook_point_id"])
    assert np.array_equal(got["hook_num_overlap_tiles"], V["hook_num_overlap_tiles"])
    image_err = np.abs(got["image"] - V["image"]).max(axis=2)
    depth_err = np.abs(got["depth"] - V["depth"])
    flipped = (image_err > image_tol) | (got["count"] != V["count"]) | (depth_err > 1e-4 * max(1.0, np.abs(V["depth"]).max()))
    n_flipped = int(flipped.sum())
    if fragile is None:
        assert n_flipped == 0, (n_flipped, float(image_err.max()))
    else:
        print(f"[parity] reference_vector.flips: fragile_pixels={int(fragile.sum())}, flipped_pixels={n_flipped}, "
              f"image_linf={float(image_err.max()):.3e}")
        assert not (flipped & ~fragile).any() and n_flipped <= MAX_FLIPPED_PIXELS and float(image_err.max()) <= 5e-3
    affected = np.abs(got["hook_num_affected_pixels"].astype(np.int64) - V["hook_num_affected_pixels"].astype(np.int64))
    # identical -- for the f64 spec build: unless a forward decision flipped, or (the backward evaluates alpha by another
    # expression, UTL:331-348) a pair sits within the same margin of 1/255 there
    assert int(affected.sum()) <= (0 if fragile is None else max(n_flipped, int(fragile.sum())))
    assert np.abs(got["features"] - V["features_after_forward"]).max() <= 2e-7      # in-place q normalisation
    assert np.abs(got["hook_uv"] - V["hook_uv"]).max() <= 1e-4 and np.abs(got["hook_depth"] - V["hook_depth"]).max() <= 1e-5
    if n_flipped:
        grad_tol = max(grad_tol, FLIP_GRAD_TOL)  # a flipped (pixel, Gaussian) pair is a discrete change, not rounding
    for key in ("grad_xyz", "grad_feat", "hook_grad_point", "hook_grad_features", "hook_grad_viewspace", "hook_magnitude",
                "hook_magnitude_image"):
        assert _rel(got[key], V[key]) <= grad_tol, (key, _rel(got[key], V[key]))
    # rows of invisible / invalid points carry exactly zero gradient, band clearing is exact (RAS:1167-1182)
    if n_flipped == 0:
        assert np.array_equal(got["grad_feat"] == 0, V["grad_feat"] == 0)
        assert np.array_equal(got["grad_xyz"] == 0, V["grad_xyz"] == 0)


def _oracle_outputs(V, s, cfg, band, precision):
    f = O.forward(s.point_cloud.numpy(), s.point_cloud_features.numpy(), s.point_invalid_mask.numpy(),
                  s.point_object_id.numpy(), s.camera_intrinsics.numpy(), s.q_pointcloud_camera.numpy(),
                  s.t_pointcloud_camera.numpy(), s.height, s.width, precision=precision, want_margin=True, **cfg)
    b = O.backward(f, V["grad_image"].astype(f["image"].dtype), band)
    h = b["hook"]
    got = dict(image=f["image"], depth=f["depth"], count=f["count"], features=f["feat"], grad_xyz=b["grad_xyz"],
               grad_feat=b["grad_feat"], hook_point_id=h["point_id_in_camera_list"],
               hook_grad_point=h["grad_point_in_camera"], hook_grad_features=h["grad_pointfeatures_in_camera"],
               hook_grad_viewspace=h["grad_viewspace"], hook_magnitude=h["magnitude_grad_viewspace"],
               hook_magnitude_image=h["magnitude_grad_viewspace_on_image"],
               hook_num_overlap_tiles=h["num_overlap_tiles"], hook_num_affected_pixels=h["num_affected_pixels"],
               hook_depth=h["point_depth"], hook_uv=h["point_uv_in_camera"])
    return got, f["margin"]


def _hip_outputs(V, s, cfg, band):
    from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Op
    dev = torch.device("cuda:0")
    s = s.to(dev)
    xyz = s.point_cloud.clone().requires_grad_(True)
    feat = s.point_cloud_features.clone().requires_grad_(True)
    hook = {}
    op = Op(Op.GaussianPointCloudRasterisationConfig(**cfg), backward_valid_point_hook=lambda h: hook.update(h=h))
    image, depth, count = op(Op.GaussianPointCloudRasterisationInput(
        point_cloud=xyz, point_cloud_features=feat, point_object_id=s.point_object_id,
        point_invalid_mask=s.point_invalid_mask,
        camera_info=CameraInfo(camera_intrinsics=s.camera_intrinsics, camera_height=s.height, camera_width=s.width,
                               camera_id=0),
        q_pointcloud_camera=s.q_pointcloud_camera, t_pointcloud_camera=s.t_pointcloud_camera, color_max_sh_band=band))
    (image * torch.from_numpy(V["grad_image"]).to(dev)).sum().backward()
    h = hook["h"]
    n = lambda t: t.detach().cpu().numpy()   # noqa: E731
    return dict(image=n(image), depth=n(depth), count=n(count), features=n(feat), grad_xyz=n(xyz.grad),
                grad_feat=n(feat.grad), hook_point_id=n(h.point_id_in_camera_list),
                hook_grad_point=n(h.grad_point_in_camera), hook_grad_features=n(h.grad_pointfeatures_in_camera),
                hook_grad_viewspace=n(h.grad_viewspace), hook_magnitude=n(h.magnitude_grad_viewspace),
                hook_magnitude_image=n(h.magnitude_grad_viewspace_on_image),
                hook_num_overlap_tiles=n(h.num_overlap_tiles), hook_num_affected_pixels=n(h.num_affected_pixels),
                hook_depth=n(h.point_depth), hook_uv=n(h.point_uv_in_camera))


def test_vectors_exist_and_cover_the_branches():
    assert len(FILES) >= 3
    saturating = clamped = tied_covered = False
    longest = largest = most_points_over_many_tiles = 0
    for path in FILES:
        V, s, cfg, band = _load(path)
        f = O.forward(s.point_cloud.numpy(), s.point_cloud_features.numpy(), s.point_invalid_mask.numpy(),
                      s.point_object_id.numpy(), s.camera_intrinsics.numpy(), s.q_pointcloud_camera.numpy(),
                      s.t_pointcloud_camera.numpy(), s.height, s.width, **cfg)
        patched = "stable_sort_patch" in V.files and int(V["stable_sort_patch"]) == 1
        ties = float((f["keys"][1:] == f["keys"][:-1]).mean())
        # no ties: outputs are well defined -- except in the vectors generated with the reference's sort call patched
        # to sort(stable=True) (the oracle's / HIP path's tie rule), which exist to cover tied keys
        assert (ties > 0.02) if patched else (ties == 0)
        longest = max(longest, int((f["tile_end"] - f["tile_start"]).max()))
        if (s.height // 16) * (s.width // 16) >= 256:
            most_points_over_many_tiles = max(most_points_over_many_tiles, int(V["hook_point_id"].shape[0]))
        largest = max(largest, s.height * s.width)
        tied_covered |= patched
        ends = f["tile_end"][(np.arange(s.height)[:, None] // 16) * (s.width // 16) + np.arange(s.width)[None] // 16]
        saturating |= bool(((1 - f["acc_alpha"] < 1e-2) & (f["last_eff"] < ends)).any())
        clamped |= bool((f["alpha"] > 0.99).any())
    assert saturating and clamped   # the T < 1e-4 stop and the 0.99 clamp are both exercised
    # the reference's 256-entry staging loops run >= 3 batches in one tile (forward RAS:382-386, backward RAS:574-585),
    # one scene has >= 64 tiles, one has tied keys under the stable rule
    assert longest > 512 and largest >= 128 * 128 and tied_covered
    # and the pin is not only on toy sizes: a reference run with >= 2,000 visible Gaussians over >= 256 tiles
    assert most_points_over_many_tiles >= 2000


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[19:-4] for p in FILES])
@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_oracle_matches_reference_operator(path, precision):
    V, s, cfg, band = _load(path)
    got, margin = _oracle_outputs(V, s, cfg, band, precision)
    f = {"margin": margin}
    # the f64 spec build differs from the fp32 reference by the reference's own rounding (the backward recovers T by
    # division, RAS:643, which amplifies it on saturating pixels): 5e-5 observed there, 7e-7 for the fp32 build
    # (image: the fp32 build is 1.2e-7 .. 3.6e-7 from the reference on every vector incl. the 2,400- and the 6,000-Gaussian
    # ones and takes the same decision on every pixel; the f64 build is 7.3e-6 away on vector i -- the reference's own
    # fp32 rounding of 2,321 small conics -- and decides four pixels of vector j differently.  Gradients, fp32 build:
    # 2e-7 .. 1e-6, and 8e-6 on vector j, where the reference's fp32 atomics add up to 100 terms per pixel and 614 per tile)
    _check(V, got, grad_tol=_grad_tol(V) if precision == "f32" else 2e-4,
           image_tol=IMAGE_TOL_ORACLE if precision == "f32" else IMAGE_TOL_F64_SPEC,
           fragile=None if precision == "f32" else f["margin"] < 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[19:-4] for p in FILES])
def test_hip_operator_matches_reference_operator(path):
    """Every pixel within 1e-4 of the reference's run, every per-pixel count and every per-Gaussian affected-pixel count
    identical: no pixel is admitted as fragile (round 5: a comparison inside the proven distance between the kernels'
    rounding and the reference's is re-evaluated in the reference's expression order, csrc/gs_common.h)."""
    V, s, cfg, band = _load(path)
    got = _hip_outputs(V, s, cfg, band)
    assert np.array_equal(got["count"], V["count"])
    assert np.array_equal(got["hook_num_affected_pixels"], V["hook_num_affected_pixels"])
    print(f"[parity] reference_vector.hip.{os.path.basename(path)[19:-11]}: image_linf={float(np.abs(got['image'] - V['image']).max()):.3e}, "
          f"flipped_pixels=0, grad_feat={_rel(got['grad_feat'], V['grad_feat']):.2e}")
    _check(V, got, grad_tol=_grad_tol(V), image_tol=IMAGE_TOL_HIP, fragile=None)


# ------------------------------------------------------------------------------------------------ needle scenes
# Draws of the randomised suite with needles (one axis of a Gaussian 10-100 x the others), run through the reference's
# sources with the emulation's exp / log correctly rounded (GS_EMU_EXP=cr, `*_exp_cr.npz`).  What the runs taught:
#  * a needle's conic amplifies the last bit of the scale activation a thousandfold.  With NumPy's fp32 exp in the
#    emulation (a few ulps; draws 17 and 20) the fp32 oracle and the reference run were 1.1e-4 / 1.4e-4 apart on the image with
#    4 / 10 flipped pixels, 2.6e-4 / 1.6e-4 in the position gradients -- and the reference run itself 2.5e-3 / 2.3e-3
#    (image, with 11 / 21 flipped pixels) from the f64 build: on such scenes "the reference" is defined by its libm to
#    1e-4 .. 1e-3.  With exp correctly rounded on both sides everything below agrees as on ordinary scenes: image
#    1.2e-7, every pixel's count, position gradients 5e-7 -- the oracle's formulas are the reference's on needles too;
#  * EXCEPT the q and s columns of the feature gradient (0.2 % / 4 % apart, relative L2): the reference adds the three
#    covariance gradients of a Gaussian with fp32 atomics, and a needle's Jacobians (GP3:237-331) cancel over four decades,
#    so the sums' last bits come out in the second digit of dL/ds.  The oracle adds in double and its fp32 build is 4.5e-4
#    from its f64 build; the reference run is 2.3e-3 (q) / 4.2e-2 (s) from it.  Those columns are held to the float64
#    yardstick: at least as close to the f64 build as the reference run is.
def _without_q_s_columns(d):
    return dict(d, grad_feat=d["grad_feat"][:, 7:], hook_grad_features=d["hook_grad_features"][:, 7:])


@pytest.mark.parametrize("path", NEEDLE_FILES, ids=[os.path.basename(p)[19:-4] for p in NEEDLE_FILES])
def test_oracle_matches_reference_operator_on_needles(path):
    V, s, cfg, band = _load(path)
    V = {k: V[k] for k in V.files}
    assert str(V["emulated_exp"]) == "correctly rounded"
    got, _ = _oracle_outputs(V, s, cfg, band, "f32")
    spec, _ = _oracle_outputs(V, s, cfg, band, "f64")
    _check(_without_q_s_columns(V), _without_q_s_columns(got), grad_tol=_grad_tol(V), image_tol=IMAGE_TOL_ORACLE)
    for key in ("grad_feat", "hook_grad_features"):
        for name, cols in (("q", slice(0, 4)), ("s", slice(4, 7))):
            d_reference, d_oracle = _rel(V[key][:, cols], spec[key][:, cols]), _rel(got[key][:, cols], spec[key][:, cols])
            print(f"[parity] needle_vector.{key}.{name}: reference_run_to_f64={d_reference:.3e}, fp32_oracle_to_f64={d_oracle:.3e}, "
                  f"fp32_oracle_to_reference_run={_rel(got[key][:, cols], V[key][:, cols]):.3e}")
            assert d_oracle <= d_reference + 1e-5, (key, name, d_oracle, d_reference)


@pytest.mark.gpu
@pytest.mark.parametrize("path", NEEDLE_FILES, ids=[os.path.basename(p)[19:-4] for p in NEEDLE_FILES])
def test_hip_operator_on_needle_vectors(path):
    """Needle scenes (one axis 10-100x the others).  The HIP operator evaluates the reference's own exponent, bit for bit,
    and takes every skip / stop decision as the reference takes it, so the integer outputs are
    exact -- every pixel's count and every Gaussian's affected-pixel count included.  The CONTINUOUS outputs of such scenes
    are ill-conditioned in fp32 whoever computes them: the reference's run and the fp32 oracle are themselves 1e-5 .. 1e-4
    from the float64 build.  Their yardstick is therefore the float64 build: the HIP operator must be as close to it as the
    reference run is (x SPEC_FACTOR, + the 1e-4 north-star floor), on the pixels where reference run and f64 build agree
    with NEEDLE_MARGIN to spare -- the yardstick of tests/test_fuzz_gpu.py, with the REFERENCE's run in the place of the fp32
    oracle.  (This is the one place where the image bar is not a flat 1e-4 against the reference run on every pixel; the
    last line holds every pixel to 5e-3, one blended Gaussian.)"""
    V, s, cfg, band = _load(path)
    V = {k: V[k] for k in V.files}
    got = _hip_outputs(V, s, cfg, band)
    o32, margin32 = _oracle_outputs(V, s, cfg, band, "f32")
    spec, margin64 = _oracle_outputs(V, s, cfg, band, "f64")
    assert np.array_equal(got["hook_point_id"], V["hook_point_id"])
    assert np.array_equal(got["hook_num_overlap_tiles"], V["hook_num_overlap_tiles"])
    assert np.abs(got["features"] - V["features_after_forward"]).max() <= 2e-7
    assert np.abs(got["hook_uv"] - V["hook_uv"]).max() <= 1e-4 and np.abs(got["hook_depth"] - V["hook_depth"]).max() <= 1e-5
    keep = (V["count"] == spec["count"]) & (margin32 >= NEEDLE_MARGIN) & (margin64 >= NEEDLE_MARGIN)
    assert keep.mean() > 0.5      # (0.94 on vector l, 0.80 on vector m: close-ups put many pixels near a threshold)
    assert np.array_equal(got["count"], V["count"])
    assert np.array_equal(got["hook_num_affected_pixels"], V["hook_num_affected_pixels"])
    linf = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())   # noqa: E731
    report = {}
    for name, dist, floor, pick in (("image", linf, 1e-4, lambda d: d["image"][keep]), ("depth", linf, 2e-4, lambda d: d["depth"][keep]),
                                    ("grad_xyz", _rel, 1e-4, lambda d: d["grad_xyz"]), ("grad_feat", _rel, 1e-4, lambda d: d["grad_feat"]),
                                    ("hook_grad_viewspace", _rel, 1e-4, lambda d: d["hook_grad_viewspace"]),
                                    ("hook_magnitude", _rel, 1e-4, lambda d: d["hook_magnitude"]),
                                    ("hook_magnitude_image", _rel, 1e-4, lambda d: d["hook_magnitude_image"])):
        d_hip, d_reference = dist(pick(got), pick(spec)), dist(pick(V), pick(spec))
        report[name] = (d_hip, d_reference, dist(pick(got), pick(V)))
        assert d_hip <= SPEC_FACTOR * d_reference + floor, (name, d_hip, d_reference)
    print("[parity] needle_vector.hip: kept_pixels=%.4f, " % keep.mean() +
          ", ".join(f"{k}: to_f64={a:.2e} reference_run_to_f64={b:.2e} to_reference_run={c:.2e}" for k, (a, b, c) in report.items()))
    assert linf(got["image"], V["image"]) <= 5e-3      # a flipped pair stays within one blended Gaussian
