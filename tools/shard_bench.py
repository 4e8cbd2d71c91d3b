"""Rank time of a tile-row sharded frame on ONE GPU (no collectives): the compute a rank does at world size G, and -- with
GS_SHARD_EXCHANGE=1 -- the device side of the sparse accumulator exchange (compaction of the produced rows, the host read
of the list length, the merge of G gathered lists; the wire time of the all-gather is NOT in it: the other ranks' lists are
stand-ins built from this rank's own, shifted to other row ids).  usage: python tools/shard_bench.py [workload]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Op, hip_ops
from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene, make_grad_image
from taichi_3d_gaussian_splatting_amd import host_affinity
if os.environ.get("GS_NO_PIN") != "1":
    host_affinity.pin_host_threads(0)   # launching threads on one L3 complex next to the GPU (as bench.py)
s = make_config_scene(sys.argv[1] if len(sys.argv) > 1 else "headline_1m_1080p").to("cuda"); g = make_grad_image(s.height, s.width).to("cuda")
MODE = os.environ.get("GS_SHARD_MODE", "bands")
RANK = int(os.environ.get("GS_SHARD_RANK", "-1"))   # -1: the middle band (the heaviest one under perspective)
WORLDS = tuple(int(x) for x in os.environ.get("GS_SHARD_WORLDS", "1,2,4,8").split(","))
for G in WORLDS:
    op = Op(Op.GaussianPointCloudRasterisationConfig())
    op.shard = (G // 2 if RANK < 0 else min(RANK, G - 1), G, MODE)
    if G > 1:   # as under torch.distributed: outputs allocated for an in-place gather, other ranks' rows not zero-filled
        op.image_gather = lambda tensors: None
    if os.environ.get("GS_BIN_SHIFT"):
        op.bin_shift = int(os.environ["GS_BIN_SHIFT"])
    xyz = s.point_cloud.clone().requires_grad_(True); feat = s.point_cloud_features.clone().requires_grad_(True)
    inp = Op.GaussianPointCloudRasterisationInput(point_cloud=xyz, point_cloud_features=feat, point_object_id=s.point_object_id,
        point_invalid_mask=s.point_invalid_mask, camera_info=CameraInfo(s.camera_intrinsics, s.height, s.width, 0),
        q_pointcloud_camera=s.q_pointcloud_camera, t_pointcloud_camera=s.t_pointcloud_camera, color_max_sh_band=3)
    saved = {}
    if G > 1 and os.environ.get("GS_SHARD_EXCHANGE") == "1":   # keep one frame's accumulators for the exchange timing below
        op.grad_accumulator_reduce = lambda acc, nk: (saved.update(acc=acc.clone(), nk=nk.clone()), acc)[1]
    def step():
        xyz.grad = None; feat.grad = None
        image, _, _ = op(inp); image.backward(g)
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    print(f"G={G} {MODE} rank {op.shard[0]} bin_shift {op.list_layout(s.height, s.width).bin_shift}: compute per step (no collectives) "
          f"{(time.perf_counter()-t0)/20*1e3:.3f} ms")
    if saved:
        acc, nk = saved["acc"], saved["nk"]; m = acc.shape[0]
        def ev(): return torch.cuda.Event(enable_timing=True)
        def send_list():
            ids, rows, count = hip_ops.compact_rows(acc, nk)
            n = int(count.item())                      # the exchange's one host synchronisation (list lengths)
            cap = max(4, -(-n // 4) * 4)
            send = torch.empty(13 * cap, dtype=torch.int32, device=acc.device)
            send[:n].copy_(ids[:n]); send[cap:cap + 12 * n].view(torch.float32).copy_(rows[:n].reshape(-1))
            return send, n, cap
        send, n, cap = send_list()
        recv = torch.empty((G, 13 * cap), dtype=torch.int32, device=acc.device)
        for r in range(G):   # stand-ins for the other ranks' lists: the same rows under other (ascending) ids
            shifted, order = torch.sort((send[:n] + r * (m // G)) % m)
            recv[r, :n] = shifted
            recv[r, cap:cap + 12 * n].view(torch.float32).view(n, 12).copy_(send[cap:cap + 12 * n].view(torch.float32).view(n, 12)[order])
        counts = torch.full((G,), n, dtype=torch.int32, device=acc.device)
        t = {"compact + length read + pack": [], "merge": []}
        for _ in range(12):
            a, b, c = ev(), ev(), ev()
            a.record(); send_list(); b.record()
            hip_ops.merge_rows(recv.view(-1), 13 * cap, cap, counts, G, m); c.record()
            torch.cuda.synchronize()
            t["compact + length read + pack"].append(a.elapsed_time(b)); t["merge"].append(b.elapsed_time(c))
        print(f"    sparse exchange, device side: rows sent {n} of {m} ({52 * n / 1e6:.1f} MB; dense {48 * m / 1e6:.1f} MB), " +
              ", ".join(f"{k} {sorted(v)[len(v) // 2]:.3f} ms" for k, v in t.items()))
