"""CPU tests of the host logic around the rasteriser (rows F1/F2 of SURVEY 8(f)): YAML configuration, SSIM and
loss, the adaptive controller's decision rules (with the two device kernels injected), the trainer's resize rule,
and the oracle restatements of the controller kernels (GP3:375-406) against an independent numpy/scipy formula."""
import math
import warnings

import numpy as np
import pytest
import torch
from scipy.ndimage import correlate1d
from scipy.spatial.transform import Rotation

from oracle import gs_oracle as O
from taichi_3d_gaussian_splatting_amd.Camera import CameraInfo
from taichi_3d_gaussian_splatting_amd.GaussianPointAdaptiveController import GaussianPointAdaptiveController as ADC
from taichi_3d_gaussian_splatting_amd.GaussianPointCloudRasterisation import GaussianPointCloudRasterisation as RAS
from taichi_3d_gaussian_splatting_amd.GaussianPointTrainer import GaussianPointCloudTrainer as TRN
from taichi_3d_gaussian_splatting_amd.LossFunction import LossFunction, ssim


# ---------------------------------------------------------------------------------------------- YAML config
def test_train_config_yaml_round_trip_and_key_styles(tmp_path):
    cfg = TRN.TrainConfig(num_iterations=123, position_learning_rate=3e-5)
    cfg.adaptive_controller_config.num_iterations_warm_up = 7
    path = tmp_path / "c.yaml"
    cfg.to_yaml_file(str(path))
    text = path.read_text()
    assert "num-iterations: 123" in text and "adaptive-controller-config:" in text   # kebab-case on disk
    assert TRN.TrainConfig.from_yaml_file(str(path)) == cfg
    mixed = """
num_iterations: 9
val-interval: 4
adaptive-controller-config:
  densification-view-space-position-gradients-threshold: 3e-6
  under_reconstructed_move_factor: 10.
rasterisation-config:
  near-plane: 0.4
position_learning_rateo: 0.5
"""
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        got = TRN.TrainConfig.from_yaml(mixed)
    assert any("position_learning_rateo" in str(w.message) for w in caught)   # dropped, like the reference does
    assert got.num_iterations == 9 and got.val_interval == 4 and got.position_learning_rate == 1e-5
    assert got.adaptive_controller_config.densification_view_space_position_gradients_threshold == 3e-6
    assert got.adaptive_controller_config.under_reconstructed_move_factor == 10.0
    assert got.rasterisation_config.near_plane == 0.4 and got.rasterisation_config.far_plane == 1000.0


# ---------------------------------------------------------------------------------------------- loss
def _numpy_ssim(X, Y):
    k = np.arange(11) - 5
    g = np.exp(-k ** 2 / (2 * 1.5 ** 2)); g /= g.sum()

    def blur(a):
        a = correlate1d(correlate1d(a, g, axis=-2, mode="constant"), g, axis=-1, mode="constant")
        return a[..., 5:-5, 5:-5]
    mx, my = blur(X), blur(Y)
    vx, vy, cxy = blur(X * X) - mx * mx, blur(Y * Y) - my * my, blur(X * Y) - mx * my
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mx * my + c1) / (mx * mx + my * my + c1)) * ((2 * cxy + c2) / (vx + vy + c2))).mean()


def test_ssim_matches_independent_formula_and_loss_composition():
    torch.manual_seed(1)
    x = torch.rand(2, 3, 40, 52, dtype=torch.float64)
    y = (x + 0.1 * torch.randn_like(x)).clamp(0, 1)
    assert abs(float(ssim(x, y)) - _numpy_ssim(x.numpy(), y.numpy())) < 1e-12
    assert abs(float(ssim(x, x)) - 1.0) < 1e-12
    loss_fn = LossFunction(LossFunction.LossFunctionConfig(lambda_value=0.2, enable_regularization=True,
                                                           regularization_weight=2.0))
    feat = torch.zeros(5, 56, dtype=torch.float64); feat[:, 4:7] = math.log(0.5)
    invalid = torch.tensor([0, 0, 1, 0, 1], dtype=torch.int8)
    total, l1, d_ssim = loss_fn(x[0], y[0], point_invalid_mask=invalid, pointcloud_features=feat)
    want_l1 = (x[0] - y[0]).abs().mean()
    want_reg = math.sqrt(3 * 0.25)
    assert torch.allclose(l1, want_l1) and abs(float(d_ssim) - (1 - _numpy_ssim(x[:1].numpy(), y[:1].numpy()))) < 1e-12
    assert abs(float(total) - (0.8 * float(want_l1) + 0.2 * float(d_ssim) + 2.0 * want_reg)) < 1e-12
    total_no_reg, _, _ = loss_fn(x[0], y[0])
    assert abs(float(total_no_reg) - (0.8 * float(want_l1) + 0.2 * float(d_ssim))) < 1e-12


# ---------------------------------------------------------------------------------------------- oracle of K9 / K10
def test_oracle_controller_kernels_against_scipy():
    rng = np.random.default_rng(0)
    n = 500
    f = rng.normal(size=(n, 56)); f[:, :4] /= np.linalg.norm(f[:, :4], axis=1, keepdims=True)
    f[:, 4:7] = rng.uniform(-3, 0, (n, 3))
    xyz, u = rng.normal(size=(n, 3)), 1 - rng.random((n, 4))
    R = Rotation.from_quat(f[:, :4]).as_matrix()
    e = np.exp(f[:, 4:7])
    axis = np.where((f[:, 4] < f[:, 5]) & (f[:, 5] > f[:, 6]), 1, np.where((f[:, 4] < f[:, 6]) & (f[:, 5] < f[:, 6]), 2, 0))
    foci = np.sqrt(e.max(1) ** 2 - e.min(1) ** 2)[:, None] * R[np.arange(n), :, axis]
    assert np.abs(O.ellipsoid_offsets(f, "f64") - foci).max() < 1e-12
    r1, r2 = np.sqrt(-2 * np.log(u[:, 0])), np.sqrt(-2 * np.log(u[:, 2]))
    z = np.stack([r1 * np.cos(2 * np.pi * u[:, 1]), r1 * np.sin(2 * np.pi * u[:, 1]), r2 * np.cos(2 * np.pi * u[:, 3])], 1)
    want = xyz + np.einsum("nij,nj->ni", R, e * z)
    assert np.abs(O.sample_from_points(xyz, f, u, "f64") - want).max() < 1e-9   # pi literal of GP3:92 has 12 digits
    assert np.abs(O.sample_from_points(xyz, f, u, "f32") - want).max() < 1e-4


# ---------------------------------------------------------------------------------------------- controller
def _controller(n=12, live=6, **cfg_kw):
    xyz = torch.nn.Parameter(torch.arange(n * 3, dtype=torch.float32).view(n, 3))
    feat = torch.zeros(n, 56); feat[:, 3] = 1.0; feat[:, 7] = 0.05   # below reset_alpha_value: iteration 0 is a reset iteration
    feat = torch.nn.Parameter(feat)
    invalid = torch.zeros(n, dtype=torch.int8); invalid[live:] = 1
    obj = torch.arange(n, dtype=torch.int32) % 2
    cfg = ADC.GaussianPointAdaptiveControllerConfig(num_iterations_warm_up=0, num_iterations_densify=1,
                                                    num_iterations_reset_alpha=10 ** 6, **cfg_kw)
    calls = []

    def fake_sample(p, f):
        calls.append(p.clone())
        return p + 1000.0 * len(calls)
    ctrl = ADC(cfg, ADC.GaussianPointAdaptiveControllerMaintainedParameters(xyz, feat, invalid, obj),
               sample_from_point=fake_sample, ellipsoid_offset=lambda f: torch.ones(f.shape[0], 3))
    return ctrl, calls


def _hook(ids, magnitude, pixels, grad_xyz=None, depth=None):
    m = len(ids)
    return RAS.BackwardValidPointHookInput(
        point_id_in_camera_list=torch.tensor(ids, dtype=torch.int32),
        grad_point_in_camera=grad_xyz if grad_xyz is not None else torch.zeros(m, 3),
        grad_pointfeatures_in_camera=torch.zeros(m, 56), grad_viewspace=torch.zeros(m, 2),
        magnitude_grad_viewspace=torch.tensor(magnitude, dtype=torch.float32),
        magnitude_grad_viewspace_on_image=torch.zeros(16, 16, 2),
        num_overlap_tiles=torch.ones(m, dtype=torch.int32),
        num_affected_pixels=torch.tensor(pixels, dtype=torch.int32),
        point_depth=depth if depth is not None else torch.full((m,), 5.0), point_uv_in_camera=torch.zeros(m, 2))


def test_controller_clone_split_remove_and_capacity():
    ctrl, calls = _controller()
    mp = ctrl.maintained_parameters
    with torch.no_grad():
        mp.pointcloud_features[4, 7] = -1.0          # transparent (< -0.5)
        mp.pointcloud_features[5, 8] = float("nan")  # broken
    before = mp.pointcloud.detach().clone()
    grad = torch.zeros(5, 3); grad[1] = torch.tensor([0.0, 0.02, 0.0])
    # 0: big gradient + many pixels -> split; 1: big gradient + few pixels -> clone; 2: small gradient -> keep;
    # 4: transparent but big gradient -> removed, not densified; 3: zero pixels, nan average must not select it
    ctrl.update(_hook([0, 1, 2, 3, 4], [1e-3, 1e-3, 1e-9, 0.0, 1e-3], [600, 10, 600, 0, 50], grad_xyz=grad))
    info = ctrl.densify_point_info
    assert info.densify_point_id.tolist() == [0, 1]
    assert sorted(info.transparent_point_id.tolist()) == [4, 5] and info.floater_point_id.numel() == 0
    assert torch.allclose(info.densify_size_reduction_factor.flatten(), torch.tensor([math.log(1.6), 0.0]))
    with torch.no_grad():
        mp.pointcloud += 0.5    # the optimiser step between backward and refinement
    ctrl.refinement()
    inv = mp.point_invalid_mask
    # rows 4,5 freed first, then refilled (lowest free rows first): live set is again 6 rows
    assert inv.tolist() == [0, 0, 0, 0, 0, 0] + [1] * 6
    f, p = mp.pointcloud_features.detach(), mp.pointcloud.detach()
    assert torch.allclose(f[0, 4:7], torch.full((3,), -math.log(1.6))) and torch.allclose(f[4, 4:7], f[0, 4:7])
    assert torch.all(f[1, 4:7] == 0) and torch.all(f[5, 4:7] == 0)
    assert f[4, 7] == 0.05 and f[5, 7] == 0.05 and not torch.isnan(f[5]).any()       # features copied from parents
    assert mp.point_object_id[4] == mp.point_object_id[0] and mp.point_object_id[5] == mp.point_object_id[1]
    # split: child and parent both re-sampled around the parent's post-step position, two independent draws
    assert len(calls) == 2 and torch.equal(calls[0], calls[1]) and torch.allclose(calls[0][0], before[0] + 0.5)
    assert torch.allclose(p[4], before[0] + 0.5 + 1000.0) and torch.allclose(p[0], before[0] + 0.5 + 2000.0)
    # clone: child = parent's pre-step position + mean positional gradient * move factor; parent untouched
    assert torch.allclose(p[5], before[1] + grad[1] * 100.0) and torch.allclose(p[1], before[1] + 0.5)
    assert ctrl.densify_point_info is None and int(ctrl.accumulated_num_in_camera.sum()) == 0

    # capacity: 3 candidates, only 1 free row -> exactly one is filled
    ctrl2, _ = _controller(n=5, live=4)
    ctrl2.update(_hook([0, 1, 2], [1.0, 1.0, 1.0], [5, 5, 5]))
    ctrl2.refinement()
    assert ctrl2.maintained_parameters.point_invalid_mask.tolist() == [0] * 5


def test_controller_schedule_floaters_offset_and_alpha_reset():
    ctrl, _ = _controller(iteration_start_remove_floater=-1, floater_near_camrea_num_pixels_threshold=100,
                          floater_depth_threshold=3.0, enable_ellipsoid_offset=True, enable_sample_from_point=False)
    mp = ctrl.maintained_parameters
    before = mp.pointcloud.detach().clone()
    depth = torch.tensor([1.0, 10.0, 1.0])
    ctrl.update(_hook([0, 1, 2], [1e-3, 1e-3, 1e-3], [500, 500, 50], depth=depth))   # 0: floater (near + huge)
    assert ctrl.densify_point_info.floater_point_id.tolist() == [0]
    assert ctrl.densify_point_info.densify_point_id.tolist() == [1, 2]
    ctrl.refinement()
    assert mp.point_invalid_mask.tolist()[:8] == [0, 0, 0, 0, 0, 0, 0, 1]   # row 0 freed and reused, row 6 filled
    p = mp.pointcloud.detach()
    assert torch.allclose(p[0], before[1] + 1.0) and torch.allclose(p[1], before[1] - 1.0)   # foci offsets
    assert torch.allclose(p[6], before[2] + 1.0) and torch.allclose(p[2], before[2] - 1.0)

    # warm-up and interval gating; alpha reset clamps from above only
    xyz = torch.nn.Parameter(torch.zeros(4, 3)); feat = torch.nn.Parameter(torch.zeros(4, 56))
    with torch.no_grad():
        feat[:, 7] = torch.tensor([-3.0, 0.05, 0.5, 4.0])
    cfg = ADC.GaussianPointAdaptiveControllerConfig(num_iterations_warm_up=3, num_iterations_densify=2,
                                                    num_iterations_reset_alpha=4, reset_alpha_value=0.1,
                                                    transparent_alpha_threshold=-100.0)
    c = ADC(cfg, ADC.GaussianPointAdaptiveControllerMaintainedParameters(
        xyz, feat, torch.zeros(4, dtype=torch.int8), torch.zeros(4, dtype=torch.int32)),
        sample_from_point=lambda p, f: p)
    selected = []
    for it in range(7):
        c.update(_hook([0, 1], [0.0, 0.0], [1, 1]))
        selected.append(c.densify_point_info is not None)
        c.refinement()
        if it < 4:
            assert feat[3, 7] == 4.0
    assert selected == [False, False, False, False, True, False, True]   # iterations 4 and 6 (>= warm-up, even)
    assert feat[:, 7].tolist() == pytest.approx([-3.0, 0.05, 0.1, 0.1])    # reset at iteration 4


# ---------------------------------------------------------------------------------------------- trainer helpers
def test_downsample_rule_and_colour_map():
    info = CameraInfo(camera_intrinsics=torch.tensor([[400.0, 0, 330.0], [0, 420.0, 250.0], [0, 0, 1]]),
                      camera_height=500, camera_width=660, camera_id=3)
    image = torch.rand(3, 500, 660)
    small, small_info = TRN._downsample_image_and_camera_info(image, info, 4)
    # 500//4 = 125 -> 112, 660//4 = 165 -> 160; intrinsics / 4, not rescaled for the crop (TRN:96-118)
    assert small.shape == (3, 112, 160) and (small_info.camera_height, small_info.camera_width) == (112, 160)
    assert torch.allclose(small_info.camera_intrinsics, torch.tensor([[100.0, 0, 82.5], [0, 105.0, 62.5], [0, 0, 1]]))
    assert info.camera_intrinsics[0, 0] == 400.0 and small_info.camera_id == 3
    flat = torch.full((3, 64, 64), 0.25)
    out, _ = TRN._downsample_image_and_camera_info(flat, CameraInfo(torch.eye(3), 64, 64, 0), 2)
    assert torch.allclose(out, torch.full((3, 32, 32), 0.25), atol=1e-6)
    cm = TRN._easy_cmap(torch.tensor([[0.0, 5.0], [35.0, 260.0]]))
    assert torch.allclose(cm[:, 0, 0], torch.ones(3)) and torch.allclose(cm[:, 1, 1], torch.zeros(3))
    assert torch.allclose(cm[:, 0, 1], torch.tensor([0.5, 1.0, 1.0])) and torch.allclose(cm[:, 1, 0], torch.tensor([0.0, 0.5, 1.0]))


# ---------------------------------------------------------------------------------------------- oracle of the loss
@pytest.mark.parametrize("hwc,clamp", [(True, True), (False, False)])
def test_oracle_loss_and_hand_derived_gradient_match_torch_autograd(hwc, clamp):
    """Two independent derivations: the oracle's analytic reverse sweep (C, double) vs autograd through the
    PyTorch restatement of pytorch_msssim's definition."""
    torch.manual_seed(3)
    H, W = 29, 37
    gt = torch.rand(3, H, W, dtype=torch.float64)
    raw = (gt + 0.3 * torch.randn(3, H, W, dtype=torch.float64)).requires_grad_(True)   # some values leave [0,1]
    x = raw.clamp(0, 1) if clamp else raw
    l1 = (x - gt).abs().mean()
    d_ssim = 1 - ssim(x[None], gt[None])
    total = 0.8 * l1 + 0.2 * d_ssim
    g_total, g_l1, g_ds = 1.3, -0.4, 0.7
    (g_total * total + g_l1 * l1 + g_ds * d_ssim).backward()
    pred_np = raw.detach().permute(1, 2, 0).contiguous().numpy() if hwc else raw.detach().numpy()
    out, grad = O.l1_ssim(pred_np, gt.numpy(), hwc=hwc, clamp=clamp, lambda_value=0.2, g_total=g_total, g_l1=g_l1,
                          g_dssim=g_ds)
    assert np.allclose(out, [total.item(), l1.item(), d_ssim.item()], rtol=0, atol=1e-13)
    want = raw.grad.permute(1, 2, 0).numpy() if hwc else raw.grad.numpy()
    assert np.abs(grad - want).max() < 1e-13
    if clamp:
        assert (grad[(pred_np < 0) | (pred_np > 1)] == 0).all()


def test_in_place_regulariser_equals_autograd_path():
    torch.manual_seed(5)
    feat = torch.nn.Parameter(torch.randn(40, 56, dtype=torch.float64))
    invalid = (torch.rand(40) < 0.3).to(torch.int8)
    loss_fn = LossFunction(LossFunction.LossFunctionConfig(regularization_weight=2.0))
    x, y = torch.rand(3, 16, 16, dtype=torch.float64), torch.rand(3, 16, 16, dtype=torch.float64)
    total, _, _ = loss_fn(x, y, point_invalid_mask=invalid, pointcloud_features=feat)
    total.backward()
    want_grad, want_total = feat.grad.clone(), total.item()
    feat.grad = torch.full_like(feat, 0.5)                       # an existing (rasteriser) gradient is kept
    image_only, _, _ = loss_fn(x, y)
    reg = loss_fn.add_regularization_gradient_(invalid, feat)
    assert abs(image_only.item() + reg.item() - want_total) < 1e-12
    assert torch.allclose(feat.grad - 0.5, want_grad, atol=1e-13)
    assert (feat.grad[invalid == 1] == 0.5).all() and (feat.grad[:, :4] == 0.5).all()
    off = LossFunction(LossFunction.LossFunctionConfig(enable_regularization=False))
    assert off.add_regularization_gradient_(invalid, feat) is None
