"""scratch: is the backward blend's partial-sum output reproducible run to run, and identical between debug / product kernels?"""
import hashlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from taichi_3d_gaussian_splatting_amd import hip_ops as ops
from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene, make_grad_image
from test_hip_parity import _stages_to_ranges
s = make_config_scene(sys.argv[1] if len(sys.argv) > 1 else "headline_1m_1080p").to("cuda")
st = _stages_to_ranges(ops, s, ops.ListLayout(bin_shift=0))
image, depth, acc_alpha, last_eff, count = ops.blend_forward(st["start"], st["end"], st["payload"], st["attrs"], s.width, s.height, st["layout"])
g = make_grad_image(s.height, s.width).cuda()
outs = []
for dbg in (False, False, True, False):
    r = ops.blend_backward_partials(st["start"], st["payload"], st["attrs"], g, acc_alpha, last_eff, st["slot_offsets"], st["n_slots"],
                                    s.width, s.height, st["layout"], debug_hits=dbg)
    p, f = r[0], r[1]
    raised = f.bool()
    outs.append(p[raised].contiguous().view(torch.int32).clone())
    print("debug" if dbg else "plain", hashlib.sha256(outs[-1].cpu().numpy().tobytes()).hexdigest()[:16], int(raised.sum()))
for i in range(1, len(outs)):
    d = (outs[i] != outs[0])
    rows = d.any(dim=1)
    print("run", i, "differs from run 0 in", int(rows.sum()), "records; columns", d.sum(dim=0).tolist())
