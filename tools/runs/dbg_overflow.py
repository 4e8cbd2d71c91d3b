import sys, torch
sys.path.insert(0, "/root/repo")
from tests.test_hip_parity import _run_operator
from taichi_3d_gaussian_splatting_amd import GaussianPointCloudRasterisation as Op
from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image, make_scene
g = make_grad_image(256, 256)
small = make_scene(n=2000, height=256, width=256, s_min=0.01, s_max=0.05, seed=1)
big = make_scene(n=20000, height=256, width=256, s_min=0.01, s_max=0.08, seed=2)
for fe in (True, False):
    op = Op(Op.GaussianPointCloudRasterisationConfig()); op.bin_shift = 0; op.frame_entry_points = fe
    _run_operator(small, g, op=op); _run_operator(small, g, op=op)
    print("guesses", op._size_guesses)
    got = _run_operator(big, g, op=op)
    print("fe", fe, op.speculation_stats, "guesses", list(op._size_guesses.values()))
    fresh = Op(Op.GaussianPointCloudRasterisationConfig()); fresh.bin_shift = 0
    ref = _run_operator(big, g, op=fresh)
    ref2 = _run_operator(big, g, op=Op(Op.GaussianPointCloudRasterisationConfig()))
    for i in range(3):
        d = (got[i].float() - ref[i].float()).abs()
        print(i, "mismatch px", int((d > 0).sum()), "max", float(d.max()), "fresh-vs-fresh", int(((ref[i].float()-ref2[i].float()).abs()>0).sum()))
    print("grad equal", torch.equal(got[4].grad, ref[4].grad))
