#!/usr/bin/env python
"""Regenerates tests/golden/*.npz from the CPU oracle (run from the repo root, CPU only).

These are REGRESSION vectors: outputs of this repo's oracle (fp32 build) on small seeded scenes, so that
an accidental change of the oracle (the definition of parity for everything the reference's own tests do
not pin) is caught on CPU, and so that the GPU box can check the HIP path against committed numbers
without re-deriving them.  They are not outputs of the reference implementation: that cannot run here
(taichi is absent; its rasterisation kernels are CUDA-only).  The reference's own known-answer vectors
(tile ranges, single-Gaussian alpha/gradients, 2x2 covariance, quaternion->R, SE(3) inverse) are restated
in tests/test_oracle_pins.py with file:line citations.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gs_oracle as O  # noqa: E402
from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image, make_scene  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SCENES = {
    # name: make_scene kwargs
    "scene_a_400pts_64x64_sh3": dict(n=400, height=64, width=64, s_min=0.02, s_max=0.12, sh_degree=3, seed=11),
    "scene_b_300pts_48x112_sh0_invalid": dict(n=300, height=48, width=112, s_min=0.02, s_max=0.1, sh_degree=0,
                                              seed=12, invalid_fraction=0.1),
}


def main():
    for name, kw in SCENES.items():
        s = make_scene(**kw)
        f = O.forward(s.point_cloud.numpy(), s.point_cloud_features.numpy(), s.point_invalid_mask.numpy(),
                      s.point_object_id.numpy(), s.camera_intrinsics.numpy(), s.q_pointcloud_camera.numpy(),
                      s.t_pointcloud_camera.numpy(), s.height, s.width, want_margin=True)
        g = make_grad_image(s.height, s.width, seed=21).numpy()
        b = O.backward(f, g, color_max_sh_band=2)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), kwargs=np.array(repr(kw)), ids=f["ids"],
            num_overlap_tiles=f["num_overlap_tiles"], keys=f["keys"], payload=f["payload"],
            tile_start=f["tile_start"], tile_end=f["tile_end"], image=f["image"], depth=f["depth"],
            count=f["count"], margin=f["margin"], grad_xyz=b["grad_xyz"], grad_feat=b["grad_feat"],
            num_affected_pixels=b["hook"]["num_affected_pixels"])
        print(name, "M", len(f["ids"]), "K", len(f["keys"]), "mean count", f["count"].mean())


if __name__ == "__main__":
    main()
