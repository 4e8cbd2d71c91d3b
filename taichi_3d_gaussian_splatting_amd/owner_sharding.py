"""Multi-GPU with OWNER-SHARDED Gaussians: tile-row bands for the pixels + a routed exchange for the Gaussians.

``distributed.py`` shards the image (rank g renders a band of tile rows) but replicates the point cloud: every rank projects
all N points, back-propagates all M visible ones and steps all parameters -- at eight GPUs that replicated per-Gaussian work
is most of a rank's frame.  Here nothing per-Gaussian is replicated:

  * rank g OWNS a contiguous block of point-cloud rows (its inputs are that block: ``point_cloud[N_g,3]``,
    ``point_cloud_features[N_g,56]``, ...; the ranks' blocks in rank order are the whole cloud, so the concatenation of
    their visible lists is the un-sharded visible list and the stable tie order of the sort is unchanged);
  * forward: it filters / projects its own rows (RAS:31-78, RAS:239-315: the reference's per-point kernels, unchanged),
    ROUTES every projected 64-B record to the band(s) its tile box reaches (``gs_route_count`` / ``gs_route_scatter``,
    one all-to-all with equal splits: fixed-size chunks with a count in the header), bins / sorts / blends the records it
    RECEIVES on its band (RAS:131-193, RAS:318-485: the same kernels, the received buffer serving as the ``attrs`` array),
    and the image rows are all-gathered in place as in ``distributed.py`` (the all-gather BASELINE.json names);
  * backward: it back-propagates its band's pixels (RAS:531-705), returns one 48-B accumulator row per received record to
    the record's owner through the same all-to-all, the owner adds the rows of a record that went to several bands in band
    order (``gs_gather_returned_rows``: a fixed order, bitwise reproducible) and runs the per-point backward (RAS:707-772)
    on ITS rows: dense gradients, the hook's fields, Adam state and the parameters themselves never cross a link.

Per rank and frame at G ranks: project N/G points, send ~1.1 M/G records of 64 B, blend 1/G of the tiles, return
~1.1 M/G rows of 48 B, back-propagate M/G points.

The frame is written as FOUR PHASES separated by the two exchanges (``OwnerShardedRasteriser``), so that the same code
runs (a) under ``torch.distributed`` (``OwnerShardedRasterisation``: an ``nn.Module`` with the operator's ``forward``
signature, one process per GPU, RCCL) and (b) in ONE process that plays all ranks in lockstep on one GPU
(``simulate_frame``: what the GPU tests and ``tools/shard_bench.py`` use -- every rank's device work exactly as in (a),
the exchanges as device copies).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist

from . import _lib, hip_ops
from .distributed import padded_image_rows, uniform_band_rows
from .GaussianPointCloudRasterisation import GaussianPointCloudRasterisation as _Op

TILE = 16


@dataclass
class _Frame:
    """What one rank keeps between the phases of a frame."""
    width: int = 0
    height: int = 0
    # owner side
    xyz: torch.Tensor = None
    features: torch.Tensor = None
    obj: torch.Tensor = None
    intrinsics: torch.Tensor = None
    q_cp: torch.Tensor = None
    t_cp: torch.Tensor = None
    t_pc: torch.Tensor = None
    visible_mask: torch.Tensor = None
    ids: torch.Tensor = None
    counters: torch.Tensor = None
    attrs: torch.Tensor = None
    num_overlap_tiles: torch.Tensor = None
    num_keys: torch.Tensor = None
    counts: torch.Tensor = None          # i32[world] on the device: records for every band
    route_ws: torch.Tensor = None
    pos: torch.Tensor = None
    capacity: int = 0
    n_visible: int = -1
    color_max_sh_band: int = 0
    # band side
    records: torch.Tensor = None         # f32[world, capacity + 1, 16] as received
    band_layout: hip_ops.ListLayout = None
    layout_bwd: hip_ops.ListLayout = None
    band_num_overlap_tiles: torch.Tensor = None
    band_num_keys: torch.Tensor = None
    slot_offsets: torch.Tensor = None
    n_slots: int = 0
    n_keys: int = 0
    list_start: torch.Tensor = None
    payload: torch.Tensor = None
    acc_alpha: torch.Tensor = None
    last_eff: torch.Tensor = None
    tile_work: torch.Tensor = None
    outputs: tuple = ()
    stats: dict = field(default_factory=dict)


class OwnerShardedRasteriser:
    """One rank's device work of an owner-sharded frame, phase by phase (see the module docstring).  Options as the
    single-GPU operator's: ``bin_shift`` (None = chosen per frame from the previous frame's sizes), ``exact_tile_cull``,
    ``ordered_dispatch``, ``backward_on_walked_lists``, ``hook_feature_gradients``."""

    def __init__(self, config: "_Op.GaussianPointCloudRasterisationConfig", rank: int, world: int,
                 backward_valid_point_hook: Optional[Callable] = None):
        if not (0 <= rank < world <= 64):
            raise ValueError("rank / world (at most 64 bands)")
        self.config, self.rank, self.world = config, rank, world
        self.hook = backward_valid_point_hook
        self.bin_shift: Optional[int] = None
        self._auto_bin_shift = 0
        self.exact_tile_cull = True
        self.ordered_dispatch = True
        self.backward_on_walked_lists = True
        self.hook_feature_gradients = True
        self.always_store_normalised_rotation = False
        self._scratch = hip_ops.Workspaces()

    # ------------------------------------------------------------------ geometry of the bands
    def band_rows(self, height: int) -> range:
        th = height // TILE
        block = uniform_band_rows(th, self.world)
        return range(min(self.rank * block, th), min((self.rank + 1) * block, th))

    def _layout(self, height: int, sharded: bool) -> hip_ops.ListLayout:
        shift = self._auto_bin_shift if self.bin_shift is None else self.bin_shift
        if not sharded:
            return hip_ops.ListLayout(bin_shift=shift, exact_cull=self.exact_tile_cull)
        rows = self.band_rows(height)
        return hip_ops.ListLayout(bin_shift=shift, exact_cull=self.exact_tile_cull, row_begin=rows.start, row_step=1,
                                  row_end=rows.stop)

    # ------------------------------------------------------------------ phase A1: project the rank's own rows, count
    def project(self, input_data, need_state: bool = True) -> _Frame:
        cfg = self.config
        cam = input_data.camera_info
        width, height = cam.camera_width, cam.camera_height
        assert width % TILE == 0 and height % TILE == 0    # RAS:1193-1194
        feats = input_data.point_cloud_features
        if not input_data.point_cloud.is_cuda:
            raise RuntimeError("the rasteriser needs tensors on a HIP device (no CPU path)")
        if not feats.is_contiguous() or feats.dtype != torch.float32 or feats.shape[1] != 56:
            raise TypeError("point_cloud_features must be a contiguous float32 [N,56] tensor (normalised in place)")
        f = _Frame(width=width, height=height, color_max_sh_band=input_data.color_max_sh_band)
        f.xyz = input_data.point_cloud.detach().contiguous()
        f.features = feats.detach()
        f.obj = input_data.point_object_id.to(torch.int32).contiguous()
        invalid = input_data.point_invalid_mask.to(torch.int8).contiguous()
        f.intrinsics = cam.camera_intrinsics.to(device=f.xyz.device, dtype=torch.float32).contiguous()
        f.t_pc = input_data.t_pointcloud_camera.to(torch.float32).contiguous()
        f.q_cp, f.t_cp = hip_ops.pose_inverse(input_data.q_pointcloud_camera.to(torch.float32).contiguous(), f.t_pc)
        f.visible_mask, f.ids, f.counters = hip_ops.filter_compact(
            f.xyz, invalid, f.obj, f.intrinsics, f.q_cp, f.t_cp, cfg.near_plane, cfg.far_plane, width, height,
            sync=False, ws=self._scratch)
        # full-image ownership: a record is completed (conic, colour) iff the Gaussian emits a key ANYWHERE on the image
        f.attrs, f.num_overlap_tiles, f.num_keys, _, _ = hip_ops.preprocess(
            f.xyz, f.features, f.obj, f.intrinsics, f.q_cp, f.t_cp, f.ids, width, height, self._layout(height, False),
            cfg.depth_to_sort_key_scale, f.counters, n_visible_on_device=True,
            always_store_rotation=self.always_store_normalised_rotation, ws=self._scratch)
        rows_per_band = uniform_band_rows(height // TILE, self.world)
        f.counts, f.route_ws = hip_ops.route_count(f.attrs, f.num_keys, f.counters, width, height, rows_per_band,
                                                   self.world)
        return f

    # ------------------------------------------------------------------ phase A2: the send chunks
    def pack(self, f: _Frame, capacity: int, n_visible: int) -> torch.Tensor:
        """-> send f32[world, capacity + 1, 16].  capacity: slots per chunk, the same on every rank (>= every count of
        every rank); n_visible: this rank's visible count (both known to the host after the one size read of the frame)."""
        rows_per_band = uniform_band_rows(f.height // TILE, self.world)
        f.capacity, f.n_visible = int(capacity), int(n_visible)
        send, f.pos = hip_ops.route_scatter(f.attrs, f.num_keys, f.counters, f.width, f.height, rows_per_band, self.world,
                                            capacity, f.counts, f.route_ws)
        return send

    # ------------------------------------------------------------------ phase B: bin, sort, blend the received records
    def blend(self, f: _Frame, received: torch.Tensor, need_state: bool = True, gather_in_place: bool = True):
        """received f32[world, capacity + 1, 16]: chunk s = what rank s sent to this band.  -> (image, depth, count):
        this rank's tile rows rendered (allocated so that the all-gather of the other bands runs in place)."""
        cfg = self.config
        width, height = f.width, f.height
        layout = self._layout(height, True)
        f.records, f.band_layout = received, layout
        records = received.view(-1, hip_ops.ATTR_STRIDE)
        counters, ntiles, nkeys, bsums, bsums_full = hip_ops.count_keys(received, width, height, layout,
                                                                        cfg.depth_to_sort_key_scale, ws=self._scratch)
        n_keys, n_slots, max_depth_key, _ = hip_ops.scan_block_sums(bsums, counters, bsums_full)   # (one size read)
        num_bins = layout.num_bins(width, height)
        kdb, depth_bits, tile_bits = hip_ops.key_layout(cfg.near_plane, cfg.far_plane, cfg.depth_to_sort_key_scale,
                                                        num_bins, max_depth_key)
        keys, payload, slot_offsets = hip_ops.make_keys(
            records, nkeys, bsums, n_keys, width, height, cfg.depth_to_sort_key_scale, layout, kdb,
            ntiles if need_state else None, bsums_full if need_state else None, ws=self._scratch)
        keys, payload = hip_ops.sort_pairs(keys, payload, depth_bits, tile_bits, kdb, in_place=False, ws=self._scratch)
        start, end = hip_ops.tile_ranges(keys, num_bins, kdb)
        del keys
        work = None
        if need_state and self.ordered_dispatch:
            work = torch.empty(hip_ops.num_owned_tiles(width, height, layout), dtype=torch.int32, device=records.device)
        emit = bool(need_state and self.backward_on_walked_lists and layout.filter != 0 and layout.bin_shift <= 2 and
                    (max(payload.shape[0], 1) << (2 * layout.bin_shift)) < 2 ** 31)
        rgb_only = bool(cfg.rgb_only)
        blended = hip_ops.blend_forward(
            start, end, payload, records, width, height, layout, rgb_only=rgb_only, need_state=need_state,
            gathered_rows=padded_image_rows(height, self.world) if gather_in_place else 0, ordered=self.ordered_dispatch,
            tile_work=work, ws=self._scratch, emit_walked_lists=emit)
        if emit:
            start, payload, blended = blended[5], blended[6], blended[:5]
        image, depth, acc_alpha, last_eff, count = blended
        if rgb_only:
            depth = torch.zeros((height, width), dtype=torch.float32, device=records.device)
            count = torch.zeros((height, width), dtype=torch.int32, device=records.device)
        f.band_num_overlap_tiles, f.band_num_keys, f.slot_offsets = ntiles, nkeys, slot_offsets
        f.n_slots, f.n_keys = int(n_slots), int(n_keys)
        f.list_start, f.payload, f.acc_alpha, f.last_eff, f.tile_work = start, payload, acc_alpha, last_eff, work
        f.layout_bwd = hip_ops.walked_layout(layout) if emit else layout
        f.outputs = (image, depth, count)
        # next frame's list layout: the single-GPU operator's rule on this band's key count scaled to the whole image
        owned = max(len(layout.owned_rows(height)), 1)
        k_frame = n_keys * (height // TILE) / owned
        m = max(int(records.shape[0]), 1)
        used = layout.bin_shift
        if n_keys > 0:
            if used == 0:
                self._auto_bin_shift = 2 if k_frame >= 64 * m else (1 if k_frame >= 2_000_000 else 0)
            elif used == 1:
                self._auto_bin_shift = 2 if k_frame >= 16 * m else (0 if k_frame < 700_000 else 1)
            else:
                self._auto_bin_shift = 1 if k_frame < 3 * m else used
        f.stats.update(records_received=int(records.shape[0]), keys=f.n_keys, slots=f.n_slots, bin_shift=used)
        return f.outputs

    # ------------------------------------------------------------------ phase C: the band's pixels, backward
    def backward_band(self, f: _Frame, grad_image: torch.Tensor) -> torch.Tensor:
        """-> f32[world, capacity + 1, 12]: chunk s = the accumulator rows of the records rank s sent, to be returned."""
        records = f.records.view(-1, hip_ops.ATTR_STRIDE)
        partials, flags, magnitude = hip_ops.blend_backward_partials(
            f.list_start, f.payload, records, grad_image, f.acc_alpha, f.last_eff, f.slot_offsets, f.n_slots, f.width,
            f.height, f.layout_bwd, tile_work=f.tile_work, ws=self._scratch)
        acc = hip_ops.reduce_partials(f.slot_offsets, f.band_num_overlap_tiles, flags, partials, f.band_num_keys, records,
                                      f.width, f.height)
        f.stats["magnitude_image"] = magnitude
        return acc.view(self.world, f.capacity + 1, hip_ops.ACC_STRIDE)

    # ------------------------------------------------------------------ phase D: the rank's own rows, backward
    def backward_points(self, f: _Frame, returned: torch.Tensor):
        """returned f32[world, capacity + 1, 12]: chunk b = the rows band b produced for this rank's records.
        -> (grad_point_cloud [N_g,3], grad_point_cloud_features [N_g,56]); calls the hook with this rank's fields."""
        cfg = self.config
        m = f.n_visible
        acc = hip_ops.gather_returned_rows(returned, f.pos, m, f.capacity)
        ids, attrs = f.ids[:m], f.attrs[:m]
        hook = self.hook
        out = hip_ops.point_backward(
            f.xyz, f.features, f.obj, f.intrinsics, f.q_cp, f.t_cp, f.t_pc, ids, acc, attrs, f.color_max_sh_band,
            cfg.grad_q_factor, cfg.grad_s_factor, cfg.grad_alpha_factor, cfg.grad_color_factor,
            cfg.grad_high_order_color_factor, want_visible=hook is not None, visible_mask=f.visible_mask,
            num_owned_tiles=f.num_keys[:m], want_visible_features=hook is not None and self.hook_feature_gradients,
            want_hook_fields=hook is not None, width=f.width, height=f.height)
        grad_xyz, grad_feat, gx_vis, gf_vis = out[:4]
        if hook is not None:   # RAS:1127-1142, with THIS RANK's rows: ids index the rank's block of the point cloud
            hook(_Op.BackwardValidPointHookInput(
                point_id_in_camera_list=ids, grad_point_in_camera=gx_vis, grad_pointfeatures_in_camera=gf_vis,
                magnitude_grad_viewspace_on_image=f.stats.get("magnitude_image"),   # this rank's band, zeros elsewhere
                num_overlap_tiles=f.num_overlap_tiles[:m], **out[4]))
        return grad_xyz, grad_feat


def _chunk_capacity(max_count: int) -> int:
    """Slots per chunk for a largest count of ``max_count``: a multiple of 64 with one spare block (16-B aligned rows)."""
    return max(64, -(-int(max_count) // 64) * 64)


# ====================================================================== (a) torch.distributed: one process per GPU
def _all_to_all_chunks(send: torch.Tensor, group) -> torch.Tensor:
    """send [world, ...] -> received [world, ...] with received[s] = rank s's send[this rank] (equal splits)."""
    recv = torch.empty_like(send)
    try:
        dist.all_to_all_single(recv.view(-1), send.view(-1), group=group)
    except RuntimeError:
        if dist.get_backend(group) == "nccl":
            raise
        # back ends without a device all-to-all (gloo on the one-GPU tests): stage through the host
        host_send, host_recv = send.cpu(), torch.empty(send.shape, dtype=send.dtype)
        dist.all_to_all_single(host_recv.view(-1), host_send.view(-1), group=group)
        recv.copy_(host_recv)
    return recv


class OwnerShardedRasterisation(torch.nn.Module):
    """The operator's surface (``forward(GaussianPointCloudRasterisationInput) -> (image, depth, count)``) for an
    owner-sharded scene under ``torch.distributed``: the input tensors are THIS RANK's block of the point cloud, the
    outputs are the full frame on every rank, the gradients (and the hook's fields) are this rank's rows."""

    def __init__(self, config, backward_valid_point_hook=None, group: Optional[dist.ProcessGroup] = None):
        super().__init__()
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.core = OwnerShardedRasteriser(config, self.rank, self.world, backward_valid_point_hook)
        self.config = config
        self.last_frame_stats = {}
        outer = self

        class _fn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, point_cloud, point_cloud_features, input_data, need_state):
                core = outer.core
                with _lib.stream_scope(point_cloud.device):
                    f = core.project(input_data, need_state)
                    # ONE size exchange per frame: every rank's per-band counts and visible count (the chunk capacity
                    # must be the same everywhere)
                    mine = torch.cat([f.counts.to(torch.int64), f.counters[:1].to(torch.int64)])
                    sizes = torch.empty((outer.world, outer.world + 1), dtype=torch.int64, device=mine.device)
                    dist.all_gather_into_tensor(sizes.view(-1), mine, group=outer.group)
                    host = sizes.cpu()
                    capacity = _chunk_capacity(int(host[:, :outer.world].max()))
                    send = core.pack(f, capacity, int(host[outer.rank, outer.world]))
                    received = _all_to_all_chunks(send, outer.group)
                    image, depth, count = core.blend(f, received, need_state)
                    outs = [image] if core.config.rgb_only else [image, depth, count]
                    from .distributed import all_gather_tile_rows
                    all_gather_tile_rows(outs, outer.rank, outer.world, outer.group)
                outer.last_frame_stats = dict(f.stats, capacity=capacity, records_sent=int(host[outer.rank, :outer.world].sum()),
                                              bytes_sent_forward=int(send.numel() * 4))
                ctx.frame = f if need_state else None
                ctx.mark_non_differentiable(count)
                ctx.set_materialize_grads(False)
                return image, depth, count

            @staticmethod
            def backward(ctx, grad_image, grad_depth, grad_count):
                f = ctx.frame
                if f is None or not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
                    return None, None, None, None
                core = outer.core
                with _lib.stream_scope(f.xyz.device):
                    if grad_image is None:
                        grad_image = torch.zeros((f.height, f.width, 3), dtype=torch.float32, device=f.xyz.device)
                    rows = core.backward_band(f, grad_image.contiguous())
                    returned = _all_to_all_chunks(rows, outer.group)
                    grad_xyz, grad_feat = core.backward_points(f, returned)
                return grad_xyz, grad_feat, None, None

        self._fn = _fn

    def forward(self, input_data):
        need_state = torch.is_grad_enabled() and (input_data.point_cloud.requires_grad or
                                                  input_data.point_cloud_features.requires_grad)
        return self._fn.apply(input_data.point_cloud, input_data.point_cloud_features, input_data, need_state)


def owned_point_rows(n_points: int, rank: int, world: int) -> range:
    """The contiguous block of point-cloud rows rank ``rank`` owns (equal blocks of ceil(N / world))."""
    block = -(-n_points // world)
    return range(min(rank * block, n_points), min((rank + 1) * block, n_points))


# ====================================================================== (b) all ranks in one process, in lockstep
def simulate_frame(cores: Sequence[OwnerShardedRasteriser], inputs: Sequence, grad_image: Optional[torch.Tensor] = None,
                   timings: Optional[dict] = None):
    """Plays one frame of ``len(cores)`` ranks on ONE device: every rank's four phases exactly as under
    torch.distributed, the exchanges as device copies between the ranks' buffers.  -> (image, depth, count, grads):
    the assembled frame and, when ``grad_image`` is given, per rank (grad_point_cloud, grad_point_cloud_features).
    timings (optional dict): filled with HIP-event times per rank and phase in ms -- what a rank's GPU does in a frame,
    exchanges excluded."""
    world = len(cores)
    dev = inputs[0].point_cloud.device

    def timed(name, rank, fn):
        if timings is None:
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        timings.setdefault("_events", []).append((name, rank, a, b))
        return out

    frames = [timed("project", g, lambda g=g: cores[g].project(inputs[g], grad_image is not None)) for g in range(world)]
    host = torch.stack([torch.cat([f.counts.to(torch.int64), f.counters[:1].to(torch.int64)]) for f in frames]).cpu()
    capacity = _chunk_capacity(int(host[:, :world].max()))
    sends = [timed("pack", g, lambda g=g: cores[g].pack(frames[g], capacity, int(host[g, world]))) for g in range(world)]
    received = [torch.stack([sends[s][g] for s in range(world)]) for g in range(world)]   # the all-to-all
    outs = [timed("blend", g, lambda g=g: cores[g].blend(frames[g], received[g], grad_image is not None,
                                                          gather_in_place=False)) for g in range(world)]
    height = frames[0].height
    image, depth, count = [torch.zeros_like(t) for t in outs[0]]
    for g in range(world):   # the all-gather of the bands
        rows = cores[g].band_rows(height)
        sl = slice(rows.start * TILE, rows.stop * TILE)
        for dst, src in zip((image, depth, count), outs[g]):
            dst[sl] = src[sl]
    grads = None
    if grad_image is not None:
        back = [timed("backward_band", g, lambda g=g: cores[g].backward_band(frames[g], grad_image)) for g in range(world)]
        returned = [torch.stack([back[b][g] for b in range(world)]) for g in range(world)]   # the all-to-all back
        grads = [timed("backward_points", g, lambda g=g: cores[g].backward_points(frames[g], returned[g]))
                 for g in range(world)]
    if timings is not None:
        torch.cuda.synchronize(dev)
        for name, rank, a, b in timings.pop("_events"):
            timings.setdefault(rank, {})[name] = timings.get(rank, {}).get(name, 0.0) + a.elapsed_time(b)
        timings["capacity"] = capacity
        timings["records_sent"] = host[:, :world].sum(dim=1).tolist()
        timings["visible"] = host[:, world].tolist()
        timings["frames"] = [dict(f.stats, magnitude_image=None) for f in frames]
    return image, depth, count, grads
