#!/usr/bin/env python
"""Executes the drop-in recipe of INTEGRATION.md section 1 against the REFERENCE'S OWN host code.

Build-container only (needs /root/reference); run as a script in its own process because it registers stand-ins for
modules the container lacks (``taichi`` -> tests/golden/taichi_emulation.py; ``dataclass_wizard``, ``pytorch_msssim``,
``torchvision``, ``tensorboard``, ``plyfile`` -> import-time stubs).  TEST INFRASTRUCTURE.

What it proves:
  1. the python code block of INTEGRATION.md (extracted and exec'd verbatim) makes the reference's
     ``GaussianPointAdaptiveController`` (ADC:4 imports ``load_point_cloud_row_into_gaussian_point_3d``),
     ``ImagePoseDataset`` (DST:11 imports TILE_WIDTH/TILE_HEIGHT) and ``GaussianPointTrainer`` modules import and bind
     the drop-in operator;
  2. the reference's controller kernel ``compute_ellipsoid_offset`` (ADC:10-25) runs on rows loaded by the drop-in's
     ``load_point_cloud_row_into_gaussian_point_3d`` and reproduces the f64 oracle's foci offsets;
  3. the reference's controller consumes the drop-in's ``BackwardValidPointHookInput``;
  4. row F4: the reference's ``ImagePoseDataset`` (``__getitem__`` and ``_autoscale_image_and_camera_info``, DST:41-96),
     run with a ``torchvision`` stand-in that implements torchvision's documented resize rule, returns the same image
     sizes, crops and intrinsics as the package's ``ImagePoseDataset`` on landscape / portrait / oversize images.
"""
import importlib
import json
import os
import re
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REFERENCE = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "golden"))


def torchvision_resized_hw(h, w, size, max_size):
    """torchvision.transforms.functional.resize's output size for an int ``size`` (shorter edge) with ``max_size``
    (its documented rule: the shorter edge becomes ``size``; if the longer edge then exceeds ``max_size`` the image is
    scaled so that the longer edge equals ``max_size``; all roundings are int() truncations)."""
    short, long_ = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long_ / short)
    if new_long > max_size:
        new_short, new_long = int(max_size * new_short / new_long), max_size
    new_w, new_h = (new_short, new_long) if w <= h else (new_long, new_short)
    return new_h, new_w


def install_stubs():
    import taichi_emulation as E
    ti, tm = E.build_taichi_module()
    sys.modules["taichi"], sys.modules["taichi.math"] = ti, tm
    wizard = types.ModuleType("dataclass_wizard")
    wizard.YAMLWizard = type("YAMLWizard", (), {})
    sys.modules["dataclass_wizard"] = wizard
    msssim = types.ModuleType("pytorch_msssim")
    msssim.ssim = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub"))
    sys.modules["pytorch_msssim"] = msssim
    ply = types.ModuleType("plyfile")
    ply.PlyData = ply.PlyElement = object
    sys.modules["plyfile"] = ply
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = type("SummaryWriter", (), {"__init__": lambda self, *a, **k: None})
    sys.modules["torch.utils.tensorboard"] = tb

    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")
    tvu = types.ModuleType("torchvision.utils")

    def resize(img, size, max_size=None, antialias=None):
        # tensor path of torchvision: bilinear interpolate with antialias, output size by the rule above
        assert isinstance(size, int) and antialias
        new_h, new_w = torchvision_resized_hw(img.shape[-2], img.shape[-1], size, max_size)
        return torch.nn.functional.interpolate(img[None], size=(new_h, new_w), mode="bilinear", antialias=True,
                                               align_corners=False)[0]

    def to_tensor(pil_image):
        arr = np.asarray(pil_image, dtype=np.float32) / 255.0
        return torch.from_numpy(arr).permute(2, 0, 1).contiguous()

    tvf.resize, tvf.to_tensor = resize, to_tensor
    tvt.functional = tvf
    tvu.make_grid = lambda *a, **k: None
    tv.transforms, tv.utils = tvt, tvu
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": tvf,
                        "torchvision.utils": tvu})
    import matplotlib
    matplotlib.use("Agg")


def run_recipe():
    """exec the python block of INTEGRATION.md section 1, verbatim."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# sitecustomize.*?)```", text, re.S).group(1)
    exec(compile(block, "INTEGRATION.md", "exec"), {})
    return block


def main():
    assert os.path.isdir(REFERENCE), "build container only"
    install_stubs()
    sys.path.insert(0, REFERENCE)     # the reference package as its user has it (a namespace package: no __init__.py)
    block = run_recipe()
    amd = importlib.import_module("taichi_3d_gaussian_splatting_amd.GaussianPointCloudRasterisation")

    adc_mod = importlib.import_module("taichi_3d_gaussian_splatting.GaussianPointAdaptiveController")
    dst_mod = importlib.import_module("taichi_3d_gaussian_splatting.ImagePoseDataset")
    trn_mod = importlib.import_module("taichi_3d_gaussian_splatting.GaussianPointTrainer")
    for mod in (adc_mod, trn_mod):
        assert mod.__file__.startswith(REFERENCE), mod.__file__
        assert mod.GaussianPointCloudRasterisation is amd.GaussianPointCloudRasterisation
    assert adc_mod.load_point_cloud_row_into_gaussian_point_3d is amd.load_point_cloud_row_into_gaussian_point_3d
    assert (dst_mod.TILE_WIDTH, dst_mod.TILE_HEIGHT) == (16, 16)
    assert trn_mod.CameraInfo is importlib.import_module("taichi_3d_gaussian_splatting_amd.Camera").CameraInfo
    print("recipe: reference controller / dataset / trainer modules import against the drop-in")

    # 2. the reference's Taichi kernel on rows loaded by the drop-in's loader, vs the f64 oracle
    from oracle import gs_oracle as O
    g = np.random.default_rng(5)
    n = 37
    xyz = torch.tensor(g.normal(size=(n, 3)), dtype=torch.float32)
    feat = torch.tensor(g.normal(size=(n, 56)) * 0.6, dtype=torch.float32)
    feat[:, :4] /= feat[:, :4].norm(dim=1, keepdim=True)
    row = amd.load_point_cloud_row_into_gaussian_point_3d(xyz.numpy(), feat.numpy(), 3)
    gp3 = importlib.import_module("taichi_3d_gaussian_splatting.GaussianPoint3D").GaussianPoint3D
    assert isinstance(row, gp3), type(row)
    assert np.array_equal(np.asarray(list(row.color_b)), feat[3, 40:56].numpy()) and float(row.alpha) == float(feat[3, 7])
    off = torch.zeros(n, 3)
    adc_mod.compute_ellipsoid_offset(xyz, feat, off)
    ref = O.ellipsoid_offsets(feat.numpy().astype(np.float64), "f64")
    err = float(np.abs(off.numpy() - ref).max() / np.abs(ref).max())
    print(f"compute_ellipsoid_offset through the drop-in loader vs f64 oracle: rel max err {err:.2e}")
    assert err < 1e-5

    # 3. the reference's controller consumes the drop-in's hook payload
    ADC = adc_mod.GaussianPointAdaptiveController
    invalid = torch.zeros(n, dtype=torch.int8)
    ctrl = ADC(ADC.GaussianPointAdaptiveControllerConfig(num_iterations_warm_up=0, num_iterations_densify=10 ** 6),
               ADC.GaussianPointAdaptiveControllerMaintainedParameters(
                   pointcloud=xyz, pointcloud_features=feat, point_invalid_mask=invalid,
                   point_object_id=torch.zeros(n, dtype=torch.int32)))
    m = 11
    ids = torch.arange(0, 2 * m, 2, dtype=torch.int32)
    ctrl.update(amd.GaussianPointCloudRasterisation.BackwardValidPointHookInput(
        point_id_in_camera_list=ids, grad_point_in_camera=torch.ones(m, 3), grad_pointfeatures_in_camera=torch.ones(m, 56),
        grad_viewspace=torch.ones(m, 2), magnitude_grad_viewspace=torch.ones(m),
        magnitude_grad_viewspace_on_image=torch.zeros(16, 16, 2), num_overlap_tiles=torch.ones(m, dtype=torch.int32),
        num_affected_pixels=torch.full((m,), 5, dtype=torch.int32), point_depth=torch.ones(m),
        point_uv_in_camera=torch.zeros(m, 2)))
    assert int(ctrl.accumulated_num_in_camera[ids.long()].sum()) == m and int(ctrl.accumulated_num_pixels.sum()) == 5 * m
    print("reference controller update() accepted the drop-in's BackwardValidPointHookInput")

    # 4. F4: the reference's dataset vs the package's, on the same files
    import PIL.Image
    from taichi_3d_gaussian_splatting_amd.ImagePoseDataset import ImagePoseDataset as Mine, _resized_hw
    RefDataset = dst_mod.ImagePoseDataset
    sizes = [(120, 200), (203, 131), (1080, 1920), (1700, 1100), (2000, 3000), (1601, 1601), (1600, 1600), (900, 1616)]
    with tempfile.TemporaryDirectory() as tmp:
        records = []
        for i, (h, w) in enumerate(sizes):
            arr = g.integers(0, 256, (h, w, 3)).astype(np.uint8)
            path = os.path.join(tmp, f"img{i}.png")
            PIL.Image.fromarray(arr).save(path)
            T = np.eye(4); T[:3, 3] = g.normal(size=3)
            records.append(dict(image_path=path, T_pointcloud_camera=T.tolist(),
                                camera_intrinsics=[[0.9 * w, 0.0, w / 2 + 1.5], [0.0, 0.8 * w, h / 2 - 2.0], [0, 0, 1.0]],
                                camera_height=h + (i % 2) * 7, camera_width=w + (i % 3) * 5, camera_id=i))
        jpath = os.path.join(tmp, "data.json")
        with open(jpath, "w") as fh:
            json.dump(records, fh)
        ref_ds, my_ds = RefDataset(jpath), Mine(jpath)
        assert len(ref_ds) == len(my_ds) == len(sizes)
        for i, (h, w) in enumerate(sizes):
            ri, rq, rt, rc = ref_ds[i]
            mi, mq, mt, mc = my_ds[i]
            assert (rc.camera_height, rc.camera_width) == (mc.camera_height, mc.camera_width), (i, rc, mc)
            assert tuple(ri.shape) == tuple(mi.shape) == (3, rc.camera_height, rc.camera_width)
            assert rc.camera_height % 16 == 0 and rc.camera_width % 16 == 0
            kerr = float((rc.camera_intrinsics.double() - mc.camera_intrinsics.double()).abs().max())
            ierr = float((ri - mi).abs().max())
            assert kerr < 1e-3 * max(h, w) * 1e-3 + 1e-4, (i, kerr)
            assert ierr < 1e-6, (i, ierr)
            assert torch.allclose(rq.float(), mq.float(), atol=1e-6) and torch.allclose(rt.float(), mt.float(), atol=1e-6)
            if max(h - h % 16, w - w % 16) > 1600:
                hh, ww = h - h % 16, w - w % 16
                assert _resized_hw(hh, ww) == torchvision_resized_hw(hh, ww, 1024, 1600)
            print(f"dataset {h}x{w}: -> {rc.camera_height}x{rc.camera_width}, intrinsics err {kerr:.1e}, image err {ierr:.1e}")
    print("OK")
    return block


if __name__ == "__main__":
    main()
