#!/usr/bin/env python
"""Per-stage timing of the HIP path on one workload (development tool; run through gpurun).
usage: python tools/stage_bench.py [workload] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_3d_gaussian_splatting_amd import hip_ops  # noqa: E402
from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene, make_grad_image  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "headline_1m_1080p"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
s = make_config_scene(workload).to("cuda")
g = make_grad_image(s.height, s.width).to("cuda")
times = {}
CULL = os.environ.get('GS_CULL', '1') == '1'
AB = os.environ.get('GS_AB', '0') == '1'
ORDERED = os.environ.get('GS_TILE_ORDER', '0') == '1'
ARMS = os.environ.get('GS_ARMS', '0') == '1'
arm_out, arm_acc = {}, {}
LAYOUT = hip_ops.ListLayout(bin_shift=int(os.environ.get('GS_BIN_SHIFT', '0')), exact_cull=CULL)


def timed(name, fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); out = fn(); b.record()
    times.setdefault(name, []).append((a, b))
    return out


num_tiles = LAYOUT.num_bins(s.width, s.height)
kdb, db, tb = hip_ops.key_layout(s.near_plane, s.far_plane, s.depth_to_sort_key_scale, num_tiles)
q_cp, t_cp = hip_ops.pose_inverse(s.q_pointcloud_camera, s.t_pointcloud_camera)
feat = s.point_cloud_features.clone()
for _ in range(reps + 2):
    vmask, ids, counters = timed("filter_compact", lambda: hip_ops.filter_compact(
        s.point_cloud, s.point_invalid_mask, s.point_object_id, s.camera_intrinsics, q_cp, t_cp, s.near_plane,
        s.far_plane, s.width, s.height))
    attrs, ntiles, nowned, bsums, bsums_full = timed("preprocess", lambda: hip_ops.preprocess(
        s.point_cloud, feat, s.point_object_id, s.camera_intrinsics, q_cp, t_cp, ids, s.width, s.height, LAYOUT,
        s.depth_to_sort_key_scale, counters))
    k, n_slots, max_dq, _m = timed("scan_block_sums", lambda: hip_ops.scan_block_sums(bsums, counters, bsums_full))
    kdb, db, tb = hip_ops.key_layout(s.near_plane, s.far_plane, s.depth_to_sort_key_scale, num_tiles, max_dq)
    keys, payload, slot_off = timed("make_keys", lambda: hip_ops.make_keys(attrs, nowned, bsums, k, s.width, s.height,
                                                                 s.depth_to_sort_key_scale, LAYOUT, kdb, ntiles, bsums_full))
    keys, payload = timed("sort_pairs", lambda: hip_ops.sort_pairs(keys, payload, db, tb, kdb, in_place=False,
                                                                        bins_in_any_order=True))
    start, end = timed("tile_ranges", lambda: hip_ops.tile_ranges(keys, num_tiles, kdb))
    image, depth, acc_alpha, last_eff, count = timed("blend_forward", lambda: hip_ops.blend_forward(
        start, end, payload, attrs, s.width, s.height, LAYOUT))
    if ORDERED:   # the library's own dispatch order (longest list first) + the walk lengths for the backward
        tile_work = torch.empty(hip_ops.num_owned_tiles(s.width, s.height, LAYOUT), dtype=torch.int32, device="cuda")
        out_o = timed("blend_forward_ordered", lambda: hip_ops.blend_forward(
            start, end, payload, attrs, s.width, s.height, LAYOUT, ordered=True, tile_work=tile_work))
    if AB:   # A/B arm: inference forward (image only)
        timed("blend_forward_rgb_nostate", lambda: hip_ops.blend_forward(
            start, end, payload, attrs, s.width, s.height, LAYOUT, rgb_only=True, need_state=False))
    partials, flags, mag = timed("blend_backward", lambda: hip_ops.blend_backward_partials(
        start, payload, attrs, g, acc_alpha, last_eff, slot_off, n_slots, s.width, s.height, LAYOUT))
    if ORDERED:   # A/B arm: tiles dispatched longest walk first (order computed with torch, outside the timed region)
        tw_, th_ = s.width // 16, s.height // 16
        side_ = 1 << LAYOUT.bin_shift
        bins_u_ = (tw_ + side_ - 1) // side_
        tb_ = (torch.arange(th_, device="cuda")[:, None] // side_) * bins_u_ + (torch.arange(tw_, device="cuda")[None, :] // side_)
        walked_ = last_eff.view(th_, 16, tw_, 16).amax(dim=(1, 3)) - start[tb_.long()]
        order = torch.argsort(walked_.flatten(), descending=True, stable=True).to(torch.int32)
        partials2, flags2, mag2 = timed("blend_backward_lpt", lambda: hip_ops.blend_backward_partials(
            start, payload, attrs, g, acc_alpha, last_eff, slot_off, n_slots, s.width, s.height, LAYOUT, tile_order=order))
        partials3, flags3, mag3 = timed("blend_backward_ordered", lambda: hip_ops.blend_backward_partials(
            start, payload, attrs, g, acc_alpha, last_eff, slot_off, n_slots, s.width, s.height, LAYOUT, tile_work=tile_work))
        if LAYOUT.filter != 0:   # the operator's path on binned layouts: forward emits the walked per-tile lists
            work_w = torch.empty_like(tile_work)
            out_w = timed("blend_forward_ordered_emitting", lambda: hip_ops.blend_forward(
                start, end, payload, attrs, s.width, s.height, LAYOUT, ordered=True, tile_work=work_w, emit_walked_lists=True))
            partials4, flags4, mag4 = timed("blend_backward_ordered_walked", lambda: hip_ops.blend_backward_partials(
                out_w[5], out_w[6], attrs, g, out_w[2], out_w[3], slot_off, n_slots, s.width, s.height,
                hip_ops.walked_layout(LAYOUT), tile_work=work_w))
            walked_same = bool(torch.equal(partials4[flags4.bool()], partials[flags.bool()]) and torch.equal(mag4, mag))
    acc = timed("reduce_partials", lambda: hip_ops.reduce_partials(slot_off, ntiles, flags, partials))
    unfused = timed("point_backward", lambda: hip_ops.point_backward(
        s.point_cloud, feat, s.point_object_id, s.camera_intrinsics, q_cp, t_cp, s.t_pointcloud_camera, ids, acc, attrs, 3,
        1.0, 0.5, 20.0, 5.0, 1.0, False, vmask, nowned))
    if ARMS and LAYOUT.bin_shift == 0:   # the forms of the blend kernels on the same per-tile lists
        for arm_ in ("two_waves", "four_waves"):
            arm_out[arm_] = timed(f"blend_backward[{arm_}]", lambda: hip_ops.blend_backward_partials(
                start, payload, attrs, g, acc_alpha, last_eff, slot_off, n_slots, s.width, s.height, LAYOUT, arm=arm_,
                tile_work=tile_work if ORDERED else None))
            timed(f"blend_forward[{arm_}]", lambda: hip_ops.blend_forward(
                start, end, payload, attrs, s.width, s.height, LAYOUT, arm=arm_, ordered=ORDERED))
            arm_acc[arm_] = hip_ops.reduce_partials(slot_off, ntiles, arm_out[arm_][1], arm_out[arm_][0]).clone()
    if AB:   # A/B arm: slot reduction fused into the per-point kernel
        fused = timed("point_backward_fused", lambda: hip_ops.point_backward(
            s.point_cloud, feat, s.point_object_id, s.camera_intrinsics, q_cp, t_cp, s.t_pointcloud_camera, ids, None, attrs, 3,
            1.0, 0.5, 20.0, 5.0, 1.0, False, vmask, nowned, slots=(slot_off, ntiles, flags, partials), width=s.width,
            height=s.height))
torch.cuda.synchronize()
print(f"workload={workload} M={ids.shape[0]} K={k} cull={CULL} bin_shift={LAYOUT.bin_shift}")
tot = 0.0
for name, pairs in times.items():
    ms = sum(a.elapsed_time(b) for a, b in pairs[2:]) / len(pairs[2:])
    tot += 0.0 if name in ("blend_forward_rgb_nostate", "blend_backward_lpt", "blend_forward_ordered",
                           "blend_backward_ordered", "point_backward_fused") or "[" in name else ms
    print(f"  {name:16s} {ms:8.4f} ms")
print(f"  {'sum':16s} {tot:8.4f} ms")
lens = (end - start).float()
side = 16 << LAYOUT.bin_shift
bins_u = (s.width + side - 1) // side
tile_bin = (torch.arange(s.height // 16, device="cuda")[:, None] >> LAYOUT.bin_shift) * bins_u + \
    (torch.arange(s.width // 16, device="cuda")[None, :] >> LAYOUT.bin_shift)
walked = (last_eff.view(s.height // 16, 16, s.width // 16, 16).amax(dim=(1, 3)) - start[tile_bin.long()]).float().flatten()
hits = acc[:, 10].contiguous().view(torch.int32).double().sum().item()
print(f"  checksums: acc.sum={acc[:, :10].double().sum().item():.9e} acc.abs={acc[:, :10].double().abs().sum().item():.9e} "
      f"mag.sum={mag.double().sum().item():.9e}; (pixel, Gaussian) hits={hits:.0f} = "
      f"{hits / max(walked.sum().item() * 256.0, 1.0):.4f} of the visited (tile entry x 256 pixel) pairs")
for arm_, a_ in arm_acc.items():
    ref_ = arm_acc["two_waves"]
    scale_ = ref_[:, :10].abs().amax(dim=0).clamp_min(1e-30)
    print(f"  arm {arm_}: max scaled difference of the slot sums vs two_waves "
          f"{float(((a_[:, :10] - ref_[:, :10]).abs() / scale_).max()):.3e}, pixel counts equal "
          f"{bool(torch.equal(a_[:, 10].contiguous().view(torch.int32), ref_[:, 10].contiguous().view(torch.int32)))}, "
          f"magnitude image equal {bool(torch.equal(arm_out[arm_][2], arm_out['two_waves'][2]))}")
if AB:
    print(f"  fused slot reduction identical: {all(torch.equal(a, b) for a, b in zip(fused[:2], unfused[:2]))}")
if ORDERED:
    print(f"  lpt arm identical: {bool(torch.equal(partials2[flags2.bool()], partials[flags.bool()]) and torch.equal(mag2, mag))}"
          f"; library-ordered arms identical: backward "
          f"{bool(torch.equal(partials3[flags3.bool()], partials[flags.bool()]) and torch.equal(mag3, mag))}, forward "
          f"{all(torch.equal(a, b) for a, b in zip(out_o, (image, depth, acc_alpha, last_eff, count)))}; "
          f"tile_work == walked: {bool(torch.equal(tile_work.long(), walked.long()))}"
          + (f"; backward on walked lists identical: {walked_same}" if LAYOUT.filter != 0 else ""))
print(f"  bin list mean={lens.mean():.1f} max={lens.max():.0f}; list positions walked per tile (to max last) "
      f"mean={walked.mean():.1f} max={walked.max():.0f}; blended per pixel mean={count.float().mean():.2f}")
