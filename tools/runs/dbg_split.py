import sys, torch
sys.path.insert(0, "/root/repo")
from taichi_3d_gaussian_splatting_amd import hip_ops as ops
from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image, make_scene
from tests.test_hip_parity import _stages_to_ranges
for size, n, bs in ((256, 10_000, 0), (640, 60_000, 0), (256, 10_000, 1)):
    s = make_scene(n=n, height=size, width=size, s_min=0.01, s_max=0.08, seed=size + bs).to("cuda")
    layout = ops.ListLayout(bin_shift=bs)
    st = _stages_to_ranges(ops, s, layout)
    emit = bs > 0
    n_list = st["payload"].shape[0] << (2 * bs if emit else 0)
    nbytes = ops.boundary_states_bytes(n_list, size, size, layout, emit)
    boundary = torch.full((nbytes,), 0x7f, dtype=torch.uint8, device="cuda")
    work = torch.empty(ops.num_owned_tiles(size, size, layout), dtype=torch.int32, device="cuda")
    args = (st["start"], st["end"], st["payload"], st["attrs"], size, size, layout)
    plain = ops.blend_forward(*args)
    a = ops.blend_forward(*args, ordered=True, tile_work=work)
    b = ops.blend_forward(*args, boundary=boundary)
    c = ops.blend_forward(*args, ordered=True, tile_work=work, emit_walked_lists=emit, boundary=boundary)
    for name, o in (("ordered", a), ("boundary", b), ("both", c)):
        print(size, bs, name, [bool(torch.equal(o[i], plain[i])) for i in range(5)], "K", st["payload"].shape[0], "nbytes", nbytes)
