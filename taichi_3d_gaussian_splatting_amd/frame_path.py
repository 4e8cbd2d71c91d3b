"""The operator's forward and backward pass through ONE C call each (``gs_frame_forward`` / ``gs_frame_backward``,
include/gsplat_hip.h "One entry point per pass").

A frame is ~25 launches forward and ~5 backward.  Issued stage by stage from Python (``hip_ops``: ~10 us of interpreter,
``ctypes`` marshalling and ``torch.empty`` per stage) the HOST bounds every small frame: the reference's 4x / 2x
down-sampled first iterations (TRN:139-148), BASELINE config 1, a rank of a sharded frame.  Here the host does, per pass:
one slab allocation for everything that lives until the backward pass (carved up by offsets, no per-buffer tensors), the
three output tensors, one ``GsFrame`` fill and one foreign call; the library issues the launches back to back.  The kernels,
their arguments and therefore all results are those of the stage-by-stage path (``tests/test_hip_parity.py``:
options that must not change a bit).

Only the speculative case is handled here (capacities and key layout learnt from the previous frame, sizes read back behind
an event while the GPU works): a frame that does not fit is redone by the stage-by-stage path with exact sizes.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import _lib, hip_ops

_ALIGN = 256
S = _lib.STAGES
FORWARD_STAGES = (S["GS_FWD_POSE_INVERSE"] | S["GS_FWD_FILTER_COMPACT"] | S["GS_FWD_PREPROCESS"] | S["GS_FWD_SCAN"] |
                  S["GS_FWD_READ_SIZES"] | S["GS_FWD_MAKE_KEYS"] | S["GS_FWD_SORT"] | S["GS_FWD_RANGES"] | S["GS_FWD_BLEND"])


class Slab:
    """One allocation, many buffers: ``add`` reserves 256-byte aligned ranges, ``allocate`` makes the tensor, ``ptr`` /
    ``tensor`` hand out raw addresses / typed views (views are only built for what Python itself looks at)."""

    def __init__(self):
        self.offsets, self.total, self.buf, self.base = {}, 0, None, 0

    def add(self, name: str, nbytes: int) -> None:
        self.offsets[name] = (self.total, int(nbytes))
        self.total = (self.total + int(nbytes) + _ALIGN - 1) // _ALIGN * _ALIGN

    def allocate(self, device) -> "Slab":
        self.buf = torch.empty(max(self.total, _ALIGN), dtype=torch.uint8, device=device)
        self.base = self.buf.data_ptr()
        return self

    def ptr(self, name: str) -> int:
        return self.base + self.offsets[name][0] if name in self.offsets else 0

    def tensor(self, name: str, dtype: torch.dtype, shape) -> torch.Tensor:
        off, nbytes = self.offsets[name]
        return self.buf[off:off + nbytes].view(dtype).view(shape)


class FrameState:
    """What the backward pass needs of a forward pass that went through ``gs_frame_forward``."""
    __slots__ = ("frame", "slab", "layout", "layout_bwd", "m", "n_slots", "walked", "width", "height", "n_keys", "split")


def _ws_bytes(ws: hip_ops.Workspaces, name: str, nbytes: int, device) -> int:
    return ws.get(name, max(int(nbytes), 16), torch.uint8, device).data_ptr()


def aux_stream_of(outer, device):
    """The operator's second stream on ``device`` and the two events gs_frame_forward forks / joins it with (created once
    per operator and device; GS_FWD_COLOUR_ASYNC)."""
    aux = outer._aux_streams.get(device)
    if aux is None:
        stream = torch.cuda.Stream(device)
        fork, join = torch.cuda.Event(), torch.cuda.Event()
        with torch.cuda.device(device):
            fork.record(torch.cuda.current_stream(device))   # (creates the events the library records by handle)
            join.record(torch.cuda.current_stream(device))
        aux = outer._aux_streams[device] = (stream, fork, join)
    return aux


def forward(outer, xyz, features, invalid, obj, intrinsics, q_pc, t_pc, camera_info, color_max_sh_band, need_state,
            layout, guess, gathered_rows, readback):
    """-> (image, depth, count, state, host counters).  All launches of the forward pass are enqueued by one call; the host
    then waits for the frame's sizes (which arrive while the GPU is still working) and the CALLER checks that the
    speculative capacities held."""
    cfg = outer.config
    dev = xyz.device
    n = xyz.shape[0]
    width, height = camera_info.camera_width, camera_info.camera_height
    cap, depth_guess = int(guess[0]), int(guess[1])
    num_bins = layout.num_bins(width, height)
    n_bins = (num_bins + 1) & ~1
    kdb, depth_bits, tile_bits = hip_ops.key_layout(cfg.near_plane, cfg.far_plane, cfg.depth_to_sort_key_scale, num_bins,
                                                    depth_guess)
    key_bytes = 4 if kdb > 0 else 8
    pixels = width * height
    tiles = (width // hip_ops.TILE_WIDTH) * (height // hip_ops.TILE_HEIGHT)
    owned_tiles = hip_ops.num_owned_tiles(width, height, layout)
    rgb_only = bool(cfg.rgb_only)
    ordered = bool(outer.ordered_dispatch)
    shift2 = 2 * layout.bin_shift
    emit = bool(need_state and outer.backward_on_walked_lists and layout.filter != 0 and
                hip_ops.can_emit_walked_lists(cap, layout.bin_shift))
    n_obj = q_pc.shape[0]
    slab = Slab()
    slab.add("q_cp", 16 * n_obj)
    slab.add("t_cp", 12 * n_obj)
    slab.add("counters", 4 * hip_ops.NUM_COUNTERS)
    slab.add("visible_mask", n)
    slab.add("ids", 4 * n)
    slab.add("attrs", 64 * n)
    slab.add("ntiles", 4 * n)
    slab.add("nkeys", 4 * n)
    slab.add("ranges", 8 * n_bins)
    if need_state:
        slab.add("slot_offsets", 4 * n)
        slab.add("acc_alpha", 4 * pixels)
        slab.add("last_eff", 4 * pixels)
        if ordered:
            slab.add("tile_work", 4 * owned_tiles)
    if emit:
        slab.add("walked_list", 4 * (max(cap, 1) << shift2))
        slab.add("walked_start", 4 * tiles)
    else:   # the sorted payload itself is what the backward pass walks
        slab.add("payload", 4 * max(cap, 1))
        slab.add("payload_alt", 4 * max(cap, 1))
    split = 0
    if need_state and outer.split_small_grid_backward:   # boundary states for a split backward pass (small grids)
        split = hip_ops.boundary_states_bytes(max(cap, 1) << (shift2 if emit else 0), width, height, layout, emit)
        if split:
            slab.add("boundary", split)
    slab.allocate(dev)
    ws = outer._scratch
    lib = _lib.load()
    # outputs (allocated as hip_ops.blend_forward does: un-owned rows zero, or padded for an in-place all-gather)
    if gathered_rows >= height:
        image = torch.empty((gathered_rows, width, 3), dtype=torch.float32, device=dev)[:height]
        depth = None if rgb_only else torch.empty((gathered_rows, width), dtype=torch.float32, device=dev)[:height]
        count = None if rgb_only else torch.empty((gathered_rows, width), dtype=torch.int32, device=dev)[:height]
    else:
        alloc = torch.zeros if layout.sharded else torch.empty
        image = alloc((height, width, 3), dtype=torch.float32, device=dev)
        depth = None if rgb_only else alloc((height, width), dtype=torch.float32, device=dev)
        count = None if rgb_only else alloc((height, width), dtype=torch.int32, device=dev)

    f = _lib.GsFrame()
    f.n_points, f.n_objects, f.width, f.height = n, n_obj, width, height
    f.tile_row_begin, f.tile_row_step, f.tile_row_end = layout.row_begin, layout.row_step, layout.row_end
    f.bin_shift, f.exact_tile_cull = layout.bin_shift, int(layout.exact_cull)
    f.always_store_rotation = int(bool(outer.always_store_normalised_rotation))
    f.key_depth_bits, f.depth_bits, f.tile_bits = kdb, depth_bits, tile_bits
    f.blend_flags = (hip_ops.BLEND_RGB_ONLY if rgb_only else 0) | (0 if need_state else hip_ops.BLEND_NO_STATE)
    # frames whose walk lengths are skewed (walk_skew below: the backward pass of an earlier frame of this size sampled them):
    # the launch lasts as long as its few long tiles' chains, and four waves per tile -- one pixel per lane, half the work per
    # wave and entry -- shorten them (trained 1920 x 1072 scene: forward blend 0.274 -> 0.236 ms; bit-identical outputs)
    if layout.bin_shift == 0 and layout.filter == 0 and known_walk_skew(outer, width, height, dev):
        f.blend_flags |= hip_ops.BLEND_ARMS["four_waves"]
    f.need_state = int(bool(need_state))
    f.color_max_sh_band = int(color_max_sh_band)
    f.near_plane, f.far_plane, f.depth_scale = cfg.near_plane, cfg.far_plane, cfg.depth_to_sort_key_scale
    f.grad_q_factor, f.grad_s_factor, f.grad_alpha_factor = cfg.grad_q_factor, cfg.grad_s_factor, cfg.grad_alpha_factor
    f.grad_color_factor, f.grad_high_order_color_factor = cfg.grad_color_factor, cfg.grad_high_order_color_factor
    f.n_keys_capacity = cap
    f.xyz, f.features, f.invalid_mask, f.object_id = xyz.data_ptr(), features.data_ptr(), invalid.data_ptr(), obj.data_ptr()
    f.intrinsics, f.q_pointcloud_camera, f.t_pointcloud_camera = intrinsics.data_ptr(), q_pc.data_ptr(), t_pc.data_ptr()
    f.q_camera_pointcloud, f.t_camera_pointcloud = slab.ptr("q_cp"), slab.ptr("t_cp")
    f.visible_mask, f.ids, f.counters = slab.ptr("visible_mask"), slab.ptr("ids"), slab.ptr("counters")
    f.host_counters_pinned, f.size_event = readback.host.data_ptr(), readback.event.cuda_event
    f.size_stamp = readback.next_stamp()   # != 0: the sizes arrive as stamped words, no event is recorded
    f.attrs, f.num_overlap_tiles, f.num_keys = slab.ptr("attrs"), slab.ptr("ntiles"), slab.ptr("nkeys")
    nblk = (n + 255) // 256
    f.block_sums = _ws_bytes(ws, "f_block_sums", 4 * nblk, dev)
    f.block_sums_full = _ws_bytes(ws, "f_block_sums_full", 4 * nblk, dev)
    f.keys = _ws_bytes(ws, "f_keys", key_bytes * cap, dev)
    f.keys_alt = _ws_bytes(ws, "f_keys_alt", key_bytes * cap, dev)
    if emit:
        f.payload = _ws_bytes(ws, "f_payload", 4 * cap, dev)
        f.payload_alt = _ws_bytes(ws, "f_payload_alt", 4 * cap, dev)
    else:
        f.payload, f.payload_alt = slab.ptr("payload"), slab.ptr("payload_alt")
    f.slot_offsets = slab.ptr("slot_offsets")
    f.bin_ranges, f.n_bins = slab.ptr("ranges"), n_bins
    f.image = image.data_ptr()
    f.depth = 0 if depth is None else depth.data_ptr()
    f.valid_count = 0 if count is None else count.data_ptr()
    f.acc_alpha, f.last_effective = slab.ptr("acc_alpha"), slab.ptr("last_eff")
    f.tile_order = _ws_bytes(ws, "f_order_fwd", 4 * owned_tiles, dev) if ordered else 0
    f.tile_work = slab.ptr("tile_work")
    f.walked_list, f.walked_start = slab.ptr("walked_list"), slab.ptr("walked_start")
    f.boundary_states = slab.ptr("boundary")
    f.split_workspace = hip_ops.split_workspace(ws, width, height, dev).data_ptr() if split else 0
    fsplit = hip_ops.forward_split_workspace(ws, width, height, layout, dev) if getattr(outer, "split_small_grid_forward", False) else None
    f.forward_split_workspace = 0 if fsplit is None else fsplit.data_ptr()
    f.filter_workspace = _ws_bytes(ws, "f_filter", lib.gs_filter_workspace_bytes(n), dev)
    f.sort_workspace = _ws_bytes(ws, "f_sort", lib.gs_sort_workspace_bytes(cap), dev)
    stages = FORWARD_STAGES
    if getattr(outer, "colours_beside_list_stages", False):
        aux, fork, join = aux_stream_of(outer, dev)
        f.aux_stream, f.aux_event_fork, f.aux_event_join = aux.cuda_stream, fork.cuda_event, join.cuda_event
        stages |= S["GS_FWD_COLOUR_ASYNC"]
    _lib.check(lib.gs_frame_forward(ctypes.addressof(f), stages, _lib.current_stream(dev)), "gs_frame_forward")
    host = readback.wait(f.size_stamp)
    state = FrameState()
    state.frame, state.slab, state.layout, state.walked = f, slab, layout, emit
    state.split = bool(split)   # (the split backward reads the forward's image: the caller saves it for the backward pass)
    state.layout_bwd = hip_ops.walked_layout(layout) if emit else layout
    state.width, state.height = width, height
    state.m, state.n_keys = host[hip_ops.COUNTER_NUM_VISIBLE], host[hip_ops.COUNTER_NUM_KEYS]
    state.n_slots = host[hip_ops.COUNTER_NUM_SLOTS]
    return image, depth, count, state, host


SKEW_PROBE_EVERY = 16   # frames of one image size between two looks at the walk lengths
SKEW_RATIO = 5.0        # longest walk / mean walk above which a frame counts as skewed (headline scene 2.6, trained scene 10.4)


def known_walk_skew(outer, width: int, height: int, device) -> bool:
    """What walk_skew last found for this image size (False until a backward pass has looked)."""
    if getattr(outer, "backward_form_by_walk_skew", True) is False:
        return False
    p = outer.__dict__.get("_walk_skew", {}).get((width, height, device))
    return bool(p and p["skewed"])


def walk_skew(outer, slab, layout, width: int, height: int, device) -> bool:
    """Are this image size's walk lengths skewed?  Every SKEW_PROBE_EVERY-th frame (max, sum) of the forward's tile_work goes
    to pinned host memory with an asynchronous copy; the answer in force is the last one that has ARRIVED (an event query,
    never a wait): a frame or two late, which is fine for a property of the scene.  Until the first answer: not skewed."""
    if getattr(outer, "backward_form_by_walk_skew", True) is False:
        return False
    probes = outer.__dict__.setdefault("_walk_skew", {})
    p = probes.get((width, height, device))
    if p is None:
        p = probes[(width, height, device)] = {"frames": 0, "skewed": False, "pending": None,
                                               "host": torch.empty(2, dtype=torch.int64, pin_memory=True),
                                               "event": torch.cuda.Event()}
    if p["pending"] is not None and p["event"].query():
        mx, total, tiles = int(p["host"][0]), int(p["host"][1]), p["pending"]
        p["skewed"] = bool(total > 0 and mx * tiles > SKEW_RATIO * total)
        p["pending"] = None
        if os.environ.get("GS_DEBUG_SKEW"):
            print(f"[walk_skew] frame {p['frames']} {width}x{height}: max {mx} sum {total} tiles {tiles} -> {p['skewed']}", flush=True)
    if p["frames"] % SKEW_PROBE_EVERY == 0 and p["pending"] is None:
        tiles = hip_ops.num_owned_tiles(width, height, layout)
        if tiles > 0:
            work = slab.tensor("tile_work", torch.int32, (tiles,))
            p["host"].copy_(torch.stack((work.max().to(torch.int64), work.sum(dtype=torch.int64))), non_blocking=True)
            p["event"].record(torch.cuda.current_stream(device))
            p["pending"] = tiles
    p["frames"] += 1
    return p["skewed"]


def backward(outer, state: FrameState, grad_image: torch.Tensor, hook, hook_input_type, want_feature_copy: bool):
    """-> (grad_point_cloud, grad_point_cloud_features); calls the hook (RAS:1127-1142)."""
    f, slab = state.frame, state.slab
    dev = grad_image.device
    n, m, width, height = f.n_points, int(state.m), state.width, state.height
    n_slots = max(int(state.n_slots), 1)
    ws = outer._scratch
    lib = _lib.load()
    layout_bwd = state.layout_bwd
    grad_image = grad_image.contiguous()
    if grad_image.dtype != torch.float32:
        raise TypeError("grad_rasterized_image must be float32")
    f.n_visible, f.n_slots = m, int(state.n_slots)
    f.backward_bin_shift, f.backward_filter = layout_bwd.bin_shift, layout_bwd.filter
    if state.walked:
        f.list_start, f.list_payload = f.walked_start, f.walked_list
    else:
        f.list_start = f.bin_ranges
        f.list_payload = f.payload_alt if f.sorted_in_alt else f.payload
    f.grad_image = grad_image.data_ptr()
    f.partials = _ws_bytes(ws, "f_partials", 48 * n_slots, dev)
    f.slot_flags = _ws_bytes(ws, "f_slot_flags", (n_slots + 15) & ~15, dev)
    alloc = torch.zeros if state.layout.sharded else torch.empty
    magnitude = alloc((height, width, 2), dtype=torch.float32, device=dev)
    f.magnitude_image = magnitude.data_ptr()
    f.tile_order_backward = _ws_bytes(ws, "f_order_bwd", 4 * hip_ops.num_owned_tiles(width, height, state.layout), dev) \
        if f.tile_work else 0
    grad_xyz = torch.empty((n, 3), dtype=torch.float32, device=dev)
    grad_feat = torch.empty((n, hip_ops.FEATURE_DIM), dtype=torch.float32, device=dev)
    f.grad_xyz, f.grad_features = grad_xyz.data_ptr(), grad_feat.data_ptr()
    gx_vis = gf_vis = fields = None
    f.grad_xyz_visible = f.grad_features_visible = f.hook_compact = 0
    if hook is not None:
        gx_vis = torch.empty((m, 3), dtype=torch.float32, device=dev)
        compact = torch.empty(7 * m, dtype=torch.float32, device=dev)
        f.grad_xyz_visible, f.hook_compact = gx_vis.data_ptr(), compact.data_ptr()
        if want_feature_copy:
            gf_vis = torch.empty((m, hip_ops.FEATURE_DIM), dtype=torch.float32, device=dev)
            f.grad_features_visible = gf_vis.data_ptr()
        fields = dict(grad_viewspace=compact[0:2 * m].view(m, 2), magnitude_grad_viewspace=compact[2 * m:3 * m],
                      num_affected_pixels=compact[3 * m:4 * m].view(torch.int32), point_depth=compact[4 * m:5 * m],
                      point_uv_in_camera=compact[5 * m:7 * m].view(m, 2))
    stream = _lib.current_stream(dev)
    S_ = _lib.STAGES
    # the form of the two-wave backward kernel: skewed walk lengths (a few tiles walk many times the mean: trained scenes) want
    # the one with the shorter chain per hit entry (include/gsplat_hip.h GS_BLEND_SKEWED_WALKS) -- decided from the walk
    # lengths the forward pass recorded, sampled every few frames (walk_skew below: no host wait, ever)
    skewed = walk_skew(outer, slab, state.layout, width, height, dev) if f.tile_work else False
    f.blend_flags = (f.blend_flags & ~(hip_ops.BLEND_SKEWED_WALKS | hip_ops.BLEND_ARMS["four_waves"])) | \
        (hip_ops.BLEND_SKEWED_WALKS if skewed else 0)   # (the forward's four-wave choice is the forward's)
    reduce_hook = outer.grad_accumulator_reduce
    if reduce_hook is None:
        f.acc = _ws_bytes(ws, "f_acc", 48 * max(m, 1), dev)
        _lib.check(lib.gs_frame_backward(ctypes.addressof(f), S_["GS_BWD_BLEND"] | S_["GS_BWD_REDUCE"] | S_["GS_BWD_POINTS"],
                                         stream), "gs_frame_backward")
    else:   # multi-GPU with a replicated point cloud: the accumulators are summed over the ranks between the two halves
        acc = torch.empty((m, hip_ops.ACC_STRIDE), dtype=torch.float32, device=dev)
        f.acc = acc.data_ptr()
        _lib.check(lib.gs_frame_backward(ctypes.addressof(f), S_["GS_BWD_BLEND"] | S_["GS_BWD_REDUCE"], stream),
                   "gs_frame_backward")
        acc = reduce_hook(acc, slab.tensor("nkeys", torch.int32, (n,))[:m])
        f.acc = acc.data_ptr()
        _lib.check(lib.gs_frame_backward(ctypes.addressof(f), S_["GS_BWD_POINTS"], stream), "gs_frame_backward")
    if hook is not None:
        hook(hook_input_type(
            point_id_in_camera_list=slab.tensor("ids", torch.int32, (n,))[:m], grad_point_in_camera=gx_vis,
            grad_pointfeatures_in_camera=gf_vis, magnitude_grad_viewspace_on_image=magnitude,
            num_overlap_tiles=slab.tensor("ntiles", torch.int32, (n,))[:m], **fields))
    return grad_xyz, grad_feat
