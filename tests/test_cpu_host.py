"""CPU-only tests: the C-ABI library loads and exports every symbol include/gsplat_hip.h declares (no
compute calls without a GPU), host-side logic, and that the product path fails loudly without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "gsplat_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__
    __graft_entry__.build()
    from taichi_3d_gaussian_splatting_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(lib):
    declared = _declared_symbols()
    assert len(declared) >= 15
    cdll = ctypes.CDLL(lib.LIB_PATH)
    for name in declared:
        assert hasattr(cdll, name), f"libgsplat_hip.so does not export {name}"
    assert set(declared) == set(lib.EXPORTED_SYMBOLS), "python binding table out of sync with the header"
    assert lib.load().gs_abi_version() == lib.ABI_VERSION


def test_argument_validation_without_gpu(lib):
    """Entry points validate their arguments before touching the device."""
    l = lib.load()
    assert l.gs_pose_inverse(None, None, None, None, 0, None) == -1
    assert b"n_obj" in l.gs_last_error()
    assert l.gs_sort_pairs(None, None, None, None, 10, None, 0, 70, 3, 0, None, None) == -1
    assert l.gs_blend_forward(None, None, None, None, 100, 64, 0, 1, 4, 2, 3, None, None, None, None, None, 0, None, None, None, None, None, None) == -1
    assert b"multiple of 16" in l.gs_last_error()
    # lists that cover several tiles (bins) cannot be blended without the tile-box filter
    assert l.gs_blend_forward(None, None, None, None, 128, 64, 0, 1, 4, 2, 0, None, None, None, None, None, 0, None, None, None, None, None, None) == -1
    assert b"box filter" in l.gs_last_error()
    assert l.gs_sort_pairs(None, None, None, None, 1, None, 0, 17, 13, 0, None, None) == 0  # n <= 1: nothing to do
    assert l.gs_sort_pairs(None, None, None, None, 10, None, 25, 25, 13, 0, None, None) == -1  # 25+13 bits > 32
    assert l.gs_sort_workspace_bytes(10_000_000) > 256 * 4 * (10_000_000 // 4096)


def test_stamped_size_words_host_side(lib):
    """gs_wait_stamped_sizes (the host half of GsFrame.size_stamp): the four 64-bit words {stamp << 32 | value} are valid
    only when ALL of them carry the frame's stamp; a stale word (the previous frame's stamp) keeps the host waiting."""
    lib = lib.load()
    words = (ctypes.c_uint64 * 4)()
    sizes = (ctypes.c_int32 * 4)()
    assert lib.gs_wait_stamped_sizes(words, 7, 2000, sizes) == 1             # nothing has arrived: timeout, nothing written
    assert list(sizes) == [0, 0, 0, 0]
    for k, v in enumerate((977848, 2877171, 9524086, 400)):
        words[k] = (7 << 32) | v
    assert lib.gs_wait_stamped_sizes(words, 7, 2000, sizes) == 0
    assert list(sizes) == [977848, 2877171, 9524086, 400]
    words[2] = (6 << 32) | 123                                               # one word still from the frame before
    assert lib.gs_wait_stamped_sizes(words, 7, 2000, sizes) == 1
    words[2] = (7 << 32) | 0x7FFFFFFF                                        # a saturated count travels as it is
    assert lib.gs_wait_stamped_sizes(words, 7, 2000, sizes) == 0 and sizes[2] == 0x7FFFFFFF
    assert lib.gs_wait_stamped_sizes(words, 0, 2000, sizes) < 0              # stamp 0 is "no stamp": an argument error


def test_product_path_has_no_cpu_fallback():
    from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Op
    from taichi_3d_gaussian_splatting_amd.synthetic import make_scene
    s = make_scene(n=10, height=32, width=32, s_min=0.01, s_max=0.05)
    op = Op(Op.GaussianPointCloudRasterisationConfig())
    inp = Op.GaussianPointCloudRasterisationInput(
        point_cloud=s.point_cloud, point_cloud_features=s.point_cloud_features, point_object_id=s.point_object_id,
        point_invalid_mask=s.point_invalid_mask,
        camera_info=CameraInfo(s.camera_intrinsics, s.height, s.width, 0),
        q_pointcloud_camera=s.q_pointcloud_camera, t_pointcloud_camera=s.t_pointcloud_camera)
    with pytest.raises(RuntimeError, match="no CPU path"):
        op(inp)
    with pytest.raises(AssertionError):  # RAS:1193-1194
        inp.camera_info = CameraInfo(s.camera_intrinsics, 30, 32, 0)
        op(inp)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "taichi_3d_gaussian_splatting_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "gs_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_config_frozen_factors_are_class_attributes():
    # RAS:782-786: un-annotated => not dataclass fields; constructor kwargs for them are rejected/ignored
    from dataclasses import fields
    from taichi_3d_gaussian_splatting_amd import GaussianPointCloudRasterisation as Op
    C = Op.GaussianPointCloudRasterisationConfig
    assert [f.name for f in fields(C)] == ["near_plane", "far_plane", "depth_to_sort_key_scale", "rgb_only"]
    c = C()
    assert (c.near_plane, c.far_plane, c.depth_to_sort_key_scale, c.rgb_only) == (0.8, 1000., 100., False)
    assert (c.grad_color_factor, c.grad_high_order_color_factor, c.grad_s_factor, c.grad_q_factor,
            c.grad_alpha_factor) == (5., 1., 0.5, 1., 20.)
    I = Op.GaussianPointCloudRasterisationInput
    assert [f.name for f in fields(I)] == ["point_cloud", "point_cloud_features", "point_object_id",
                                           "point_invalid_mask", "camera_info", "q_pointcloud_camera",
                                           "t_pointcloud_camera", "color_max_sh_band"]
    H = Op.BackwardValidPointHookInput
    assert [f.name for f in fields(H)] == ["point_id_in_camera_list", "grad_point_in_camera",
                                           "grad_pointfeatures_in_camera", "grad_viewspace",
                                           "magnitude_grad_viewspace", "magnitude_grad_viewspace_on_image",
                                           "num_overlap_tiles", "num_affected_pixels", "point_depth",
                                           "point_uv_in_camera"]


def test_sort_key_bits():
    from taichi_3d_gaussian_splatting_amd.hip_ops import sort_key_bits
    assert sort_key_bits(0.8, 1000., 100., 8040) == (17, 13)   # defaults @1920x1072: 1e5 < 2^17, 8040 tiles
    assert sort_key_bits(0.4, 2000., 10., 8040) == (15, 13)    # truck config
    assert sort_key_bits(-1.0, 10., 100., 4)[0] == 64          # negative depth: full signed key
    assert sort_key_bits(0.0, 1e9, 100., 1) == (64, 0)         # quantised depth may overflow int32
    from taichi_3d_gaussian_splatting_amd.hip_ops import key_layout
    assert key_layout(0.8, 1000., 100., 8040) == (17, 17, 13)  # 30 bits: compressed 32-bit keys
    assert key_layout(0.8, 1e6, 100., 8040) == (0, 27, 13)     # 40 bits: reference 64-bit layout
    assert key_layout(-1.0, 10., 100., 4) == (0, 64, 2)


def test_pose_helpers_match_oracle_and_scipy():
    from scipy.spatial.transform import Rotation
    from oracle import gs_oracle as O
    from taichi_3d_gaussian_splatting_amd.utils import (SE3_to_quaternion_and_translation_torch, inverse_SE3_qt_torch,
                                                        quaternion_to_rotation_matrix_torch)
    rng = np.random.default_rng(0)
    q = rng.normal(size=(64, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    t = rng.normal(size=(64, 3))
    T = np.tile(np.eye(4), (64, 1, 1)); T[:, :3, :3] = Rotation.from_quat(q).as_matrix(); T[:, :3, 3] = t
    q2, t2 = SE3_to_quaternion_and_translation_torch(torch.tensor(T))
    R2 = quaternion_to_rotation_matrix_torch(q2).numpy()
    assert np.allclose(R2, T[:, :3, :3], atol=1e-12) and np.allclose(t2.numpy(), t)
    qi, ti = inverse_SE3_qt_torch(torch.tensor(q), torch.tensor(t))
    qo, to = O.inverse_se3_qt(q, t, "f64")
    assert np.allclose(qi.numpy(), qo, atol=1e-12) and np.allclose(ti.numpy(), to, atol=1e-12)


def test_synthetic_scene_is_deterministic_and_matches_survey_sizes():
    from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene
    from tests.helpers import oracle_forward
    a, b = make_config_scene("cfg1_10k_256"), make_config_scene("cfg1_10k_256")
    assert torch.equal(a.point_cloud, b.point_cloud) and torch.equal(a.point_cloud_features, b.point_cloud_features)
    f = oracle_forward(a, want_margin=False)
    # SURVEY.md section 8 preamble: cfg1 M = 1e4, K ~ 4.8e4
    assert len(f["ids"]) == 10_000 and 4.5e4 < len(f["keys"]) < 5.1e4
    assert not a.point_cloud_features[:, 9:24].any()  # "SH degree 0" = higher orders are zero in the data


def test_list_layout_object():
    from taichi_3d_gaussian_splatting_amd.hip_ops import FILTER_BOX, FILTER_CULL, PER_TILE_LISTS, ListLayout
    d = ListLayout()
    assert (d.bin_shift, d.exact_cull, d.filter, d.sharded) == (0, True, 0, False)   # per-tile lists, staged as they are
    b = ListLayout(bin_shift=2)
    assert b.filter == FILTER_BOX | FILTER_CULL and ListLayout(bin_shift=2, exact_cull=False).filter == FILTER_BOX
    assert b.num_bins(1920, 1072) == 30 * 17 and d.num_bins(1920, 1072) == 120 * 67
    assert PER_TILE_LISTS.filter == 0 and PER_TILE_LISTS.bin_shift == 0 and not PER_TILE_LISTS.exact_cull
    band = ListLayout(row_begin=8, row_end=20)
    assert band.sharded and list(band.owned_rows(1072)) == list(range(8, 20))
    assert list(ListLayout(row_begin=1, row_step=3).owned_rows(160)) == [1, 4, 7]
    assert list(ListLayout(row_begin=60, row_end=100).owned_rows(1072)) == list(range(60, 67))


def test_automatic_list_layout_rule():
    """hip_ops.next_bin_shift: the one rule behind the operator's and the owner mode's automatic layout (thresholds of
    GaussianPointCloudRasterisation._sizes_arrived's comment), with its hysteresis -- and the case that went wrong in round 6:
    a G = 8 band of the headline frame (300k keys of 140k-178k record slots: about two keys per record) must not look like
    the stress distribution because its key count was scaled to the whole image and its record count was not."""
    from taichi_3d_gaussian_splatting_amd.hip_ops import next_bin_shift as nxt
    m = 1_000_000
    assert nxt(0, 1_900_000, 1_900_000, m) == 0 and nxt(0, 2_000_000, 2_000_000, m) == 1     # per tile -> 2 x 2 from 2e6 keys
    assert nxt(1, 700_000, 700_000, m) == 1 and nxt(1, 699_999, 699_999, m) == 0             # ... back below 0.7e6 bin keys
    assert nxt(0, 64 * 20_000, 64 * 20_000, 20_000) == 2 and nxt(1, 16 * 20_000, 16 * 20_000, 20_000) == 2   # stress distribution
    assert nxt(2, 3 * 20_000 - 1, 3 * 20_000 - 1, 20_000) == 1 and nxt(2, 3 * 20_000, 3 * 20_000, 20_000) == 2
    assert nxt(1, 5e6, 5e6, 0) == 1 and nxt(2, 0, 0, 0) == 2                                  # nothing on screen: keep
    # the headline frame on one GPU (5.65e6 tile keys / 2.88e6 bin keys of 977,848 Gaussians) settles on 2 x 2 bins
    assert nxt(0, 5_652_545, 5_652_545, 977_848) == 1 and nxt(1, 2_877_171, 2_877_171, 977_848) == 1
    # one of eight bands of it in the owner mode: 301,626 keys of 178,696 record slots, 67 / 9 tile rows
    band_keys, band_records = 301_626, 178_696
    assert nxt(1, band_keys * 67 / 9, band_keys, band_records) == 1
    # (with the scaled count on both sides the 16-keys-per-Gaussian threshold trips at 16 * 9 / 67 = 2.1 keys per record:
    #  the same band with the 140,000 record slots of a smaller chunk capacity went to 4 x 4-tile bins)
    assert nxt(1, band_keys * 67 / 9, band_keys * 67 / 9, 140_000) == 2 and nxt(1, band_keys * 67 / 9, band_keys, 140_000) == 1


def test_host_affinity_picks_one_l3_complex_per_local_rank(monkeypatch):
    """host_affinity on a made-up two-socket box (no real affinity call): rank r's threads go to the r-th L3 complex of the
    GPU's NUMA node, PyTorch's intra-op pool is cut to its cores, unpin restores both."""
    import os
    import torch
    from taichi_3d_gaussian_splatting_amd import host_affinity as h
    assert h._parse_cpu_list("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    # 2 sockets x 2 complexes x 4 cores, SMT siblings at +16; the GPU hangs off socket 1 (cpus 8-15, 24-31)
    def fake_read(path):
        if path.endswith("local_cpulist"):
            return "8-15,24-31"
        cpu = int(path.split("/cpu/cpu")[1].split("/")[0])
        core = cpu % 16
        if path.endswith("shared_cpu_list"):
            base = core // 4 * 4
            return f"{base}-{base + 3},{base + 16}-{base + 19}"
        if path.endswith("thread_siblings_list"):
            return f"{core},{core + 16}"
        return None
    calls = []
    monkeypatch.setattr(h, "_read", fake_read)
    monkeypatch.setattr(h, "gpu_local_cpus", lambda i: h._parse_cpu_list(fake_read("local_cpulist")))
    monkeypatch.setattr(h, "_all_thread_ids", lambda: [101, 102])
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(32)), raising=False)
    monkeypatch.setattr(os, "sched_setaffinity", lambda tid, mask: calls.append((tid, set(mask))), raising=False)
    monkeypatch.setattr(h, "_original_mask", None)
    monkeypatch.setattr(h, "_original_torch_threads", None)
    before = torch.get_num_threads()
    try:
        monkeypatch.setenv("LOCAL_RANK", "1")
        chosen = h.pin_host_threads(0)
        assert chosen == {12, 13, 14, 15, 28, 29, 30, 31}           # second complex of the GPU's socket
        assert calls == [(101, chosen), (102, chosen)]
        assert torch.get_num_threads() == min(before, 4)            # four cores behind the eight hardware threads
        assert h.pin_host_threads(0, slot=0) == {8, 9, 10, 11, 24, 25, 26, 27}
        monkeypatch.delenv("LOCAL_RANK")
        assert h.pin_host_threads(1) == chosen and h.pin_host_threads(0) == {8, 9, 10, 11, 24, 25, 26, 27}   # no launcher: by device
        del calls[:]
        h.unpin_host_threads()
        assert calls == [(101, set(range(32))), (102, set(range(32)))] and torch.get_num_threads() == before
        monkeypatch.setenv("GS_PIN_HOST_THREADS", "0")
        assert h.pin_host_threads(0) is None
        # a pod that shows its one GPU as "0" (HIP_VISIBLE_DEVICES=0 in every container of the host): the slot is the
        # device's PCI bus number, not the list's "0" -- two pods of one socket do not share a complex
        monkeypatch.delenv("GS_PIN_HOST_THREADS")
        monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
        import types
        monkeypatch.setattr(torch.cuda, "get_device_properties", lambda i: types.SimpleNamespace(pci_bus_id=0x0B))
        assert h._stable_device_slot(0) == 0x0B
        assert h.pin_host_threads(0) == chosen                       # 11 % 2 complexes of the GPU's socket = the second one
        monkeypatch.setattr(torch.cuda, "get_device_properties", lambda i: types.SimpleNamespace())
        assert h._stable_device_slot(0) == 0                         # no bus number: the visibility list's entry
    finally:
        torch.set_num_threads(before)
