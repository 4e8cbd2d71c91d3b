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

import ctypes
from typing import Callable, Optional, Sequence

import torch
import torch.distributed as dist

from . import _lib, hip_ops
from .distributed import band_boundaries, padded_image_rows, uniform_band_rows
from .frame_path import Slab
from .GaussianPointCloudRasterisation import GaussianPointCloudRasterisation as _Op

TILE = 16


class _Frame:
    """What one rank keeps between the phases of a frame: the owner-side and the band-side ``GsFrame`` with the slabs
    their pointers refer to, the tensors that travel, and the host's copies of the sizes."""

    def __init__(self):
        self.width = self.height = 0
        self.owner = self.owner_slab = None          # projection of the rank's own rows, routing, per-point backward
        self.band = self.band_slab = None            # lists and blend of the received records, per-pixel backward
        self.keep = []                               # tensors the structs point into
        self.counts = self.counters = None           # device: records per band i32[world], the owner-side counters
        self.capacity, self.n_visible = 0, -1
        self.records = None                          # f32[world, capacity + 1, 16] as received
        self.band_layout = self.layout_bwd = None
        self.walked = False
        self.n_slots = self.n_keys = 0
        self.outputs = ()
        self.stats = {}
        self.bounds_array = None                     # ctypes int32[world + 1]: weighted band boundaries (host)


class OwnerShardedRasteriser:
    """One rank's device work of an owner-sharded frame, phase by phase (see the module docstring); every phase is ONE
    foreign call (``gs_frame_forward`` / ``gs_frame_backward`` with the phase's stages).  Options as the single-GPU
    operator's: ``bin_shift`` (None = chosen per frame from the previous frame's sizes), ``exact_tile_cull``,
    ``ordered_dispatch``, ``backward_on_walked_lists``, ``hook_feature_gradients``, ``speculative_sizes`` (the band's list
    stages are launched with the key capacity and depth range of the previous frame and redone when they do not fit)."""

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
        self.speculative_sizes = True
        self.split_small_grid_backward = True   # the band's backward pass gives a tile up to four workgroups (list splitting)
        self.split_small_grid_forward = True    # ... and so does its forward pass when the band walks per-tile lists
        # per-tile-row weights that place the bands' boundaries (distributed.band_boundaries: band g ends where the running
        # weight reaches (g + 1) / world of the total) -- THE SAME list on every rank.  None: equal bands.  A trained
        # scene crowds its Gaussians into some rows: with equal bands the slowest of eight ranks took twice the fastest
        # (``balanced_row_weights`` makes the list from the ranks' own walk lengths)
        self.row_weights: Optional[Sequence[float]] = None
        self.speculation_stats = {"frames": 0, "redone": 0}
        self._size_guesses = {}
        self._scratch = hip_ops.Workspaces()
        self._readback = None

    # ------------------------------------------------------------------ geometry of the bands
    def weights_for(self, height: int) -> Optional[list]:
        """``row_weights`` for an image of ``height`` pixels: the list itself, or -- measured on a frame of another height
        (the 4x / 2x down-sampled iterations of a training run, TRN:139-148) -- the same density over the image height
        re-binned to this frame's tile rows (plain arithmetic on the list: the same on every rank)."""
        w, th = self.row_weights, height // TILE
        if w is None or len(w) == th:
            return None if w is None else list(w)
        n, out = len(w), []
        for r in range(th):
            a, b, acc = r * n / th, (r + 1) * n / th, 0.0
            k = int(a)
            while k < n and k < b:
                acc += float(w[k]) * (min(b, k + 1) - max(a, k))
                k += 1
            out.append(acc)
        return out

    def band_bounds(self, height: int) -> list:
        """world + 1 tile rows: band g = rows [bounds[g], bounds[g + 1])."""
        return band_boundaries(height // TILE, self.world, self.weights_for(height))

    def band_rows(self, height: int) -> range:
        bounds = self.band_bounds(height)
        return range(bounds[self.rank], bounds[self.rank + 1])

    def row_work(self, fr: "_Frame") -> Optional[torch.Tensor]:
        """f32[tile rows of the image] on the device: the list positions this rank's backward pass walks, summed per tile
        row of its band (zeros elsewhere) -- after ``blend`` of a frame with state and ordered dispatch, else None.  Summed
        over the ranks it is the weight list ``balanced_row_weights`` turns into boundaries."""
        b = fr.band
        if b is None or not b.tile_work or not b.need_state:
            return None
        rows = self.band_rows(fr.height)
        cols = fr.width // TILE
        out = torch.zeros(fr.height // TILE, dtype=torch.float32, device=fr.records.device)
        if len(rows):
            work = fr.band_slab.tensor("tile_work", torch.int32, (len(rows) * cols,))
            out[rows.start:rows.stop] = work.view(len(rows), cols).sum(dim=1).to(torch.float32)
        return out

    def _layout(self, height: int, sharded: bool) -> hip_ops.ListLayout:
        shift = self._auto_bin_shift if self.bin_shift is None else self.bin_shift
        if not sharded:
            return hip_ops.ListLayout(bin_shift=shift, exact_cull=self.exact_tile_cull)
        rows = self.band_rows(height)
        return hip_ops.ListLayout(bin_shift=shift, exact_cull=self.exact_tile_cull, row_begin=rows.start, row_step=1,
                                  row_end=rows.stop)

    def _ws(self, name: str, nbytes: int, device) -> int:
        return self._scratch.get(name, max(int(nbytes), 16), torch.uint8, device).data_ptr()

    def _call(self, which: str, frame_struct, stages: int, device) -> None:
        fn = getattr(_lib.load(), which)
        _lib.check(fn(ctypes.addressof(frame_struct), stages, _lib.current_stream(device)), which)

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
        fr = _Frame()
        fr.width, fr.height = width, height
        xyz = input_data.point_cloud.detach().contiguous()
        features = feats.detach()
        dev = xyz.device
        obj = input_data.point_object_id.to(torch.int32).contiguous()
        invalid = input_data.point_invalid_mask.to(torch.int8).contiguous()
        intrinsics = cam.camera_intrinsics.to(device=dev, dtype=torch.float32).contiguous()
        q_pc = input_data.q_pointcloud_camera.to(torch.float32).reshape(-1, 4).contiguous()
        t_pc = input_data.t_pointcloud_camera.to(torch.float32).reshape(-1, 3).contiguous()
        if q_pc.shape[0] != t_pc.shape[0] or q_pc.shape[0] == 0:
            raise ValueError("q_pointcloud_camera / t_pointcloud_camera must be (K,4)/(K,3) with K >= 1")
        fr.keep = [xyz, features, obj, invalid, intrinsics, q_pc, t_pc]
        n, n_obj, world = xyz.shape[0], q_pc.shape[0], self.world
        layout = self._layout(height, False)
        slab = Slab()
        slab.add("q_cp", 16 * n_obj)
        slab.add("t_cp", 12 * n_obj)
        slab.add("counters", 4 * hip_ops.NUM_COUNTERS)
        slab.add("visible_mask", n)
        slab.add("ids", 4 * n)
        slab.add("attrs", 64 * n)
        slab.add("ntiles", 4 * n)
        slab.add("nkeys", 4 * n)
        slab.add("route_counts", 4 * world)
        slab.add("pos", 4 * world * max(n, 1))
        slab.allocate(dev)
        lib = _lib.load()
        f = _lib.GsFrame()
        f.n_points, f.n_objects, f.width, f.height = n, n_obj, width, height
        # full-image ownership: a record is completed (conic, colour) iff the Gaussian emits a key ANYWHERE on the image
        f.tile_row_begin, f.tile_row_step, f.tile_row_end = 0, 1, hip_ops._NO_ROW_LIMIT
        f.bin_shift, f.exact_tile_cull = layout.bin_shift, int(layout.exact_cull)
        f.always_store_rotation = int(bool(self.always_store_normalised_rotation))
        f.color_max_sh_band = int(input_data.color_max_sh_band)
        f.near_plane, f.far_plane, f.depth_scale = cfg.near_plane, cfg.far_plane, cfg.depth_to_sort_key_scale
        f.grad_q_factor, f.grad_s_factor, f.grad_alpha_factor = cfg.grad_q_factor, cfg.grad_s_factor, cfg.grad_alpha_factor
        f.grad_color_factor, f.grad_high_order_color_factor = cfg.grad_color_factor, cfg.grad_high_order_color_factor
        f.world, f.rows_per_band = world, uniform_band_rows(height // TILE, world)
        if self.row_weights is not None:   # (a HOST array: read by the two routing calls only, kept alive by the frame)
            fr.bounds_array = hip_ops._band_bounds_array(self.band_bounds(height), world)
            f.band_row_bounds = ctypes.addressof(fr.bounds_array)
        f.xyz, f.features, f.invalid_mask, f.object_id = xyz.data_ptr(), features.data_ptr(), invalid.data_ptr(), obj.data_ptr()
        f.intrinsics, f.q_pointcloud_camera, f.t_pointcloud_camera = intrinsics.data_ptr(), q_pc.data_ptr(), t_pc.data_ptr()
        f.q_camera_pointcloud, f.t_camera_pointcloud = slab.ptr("q_cp"), slab.ptr("t_cp")
        f.visible_mask, f.ids, f.counters = slab.ptr("visible_mask"), slab.ptr("ids"), slab.ptr("counters")
        f.attrs, f.num_overlap_tiles, f.num_keys = slab.ptr("attrs"), slab.ptr("ntiles"), slab.ptr("nkeys")
        nblk = (n + 255) // 256
        f.block_sums = self._ws("o_block_sums", 4 * nblk, dev)
        f.block_sums_full = self._ws("o_block_sums_full", 4 * nblk, dev)
        f.filter_workspace = self._ws("o_filter", lib.gs_filter_workspace_bytes(n), dev)
        f.route_workspace = self._ws("o_route", lib.gs_route_workspace_bytes(n, world), dev)
        f.route_counts, f.route_pos = slab.ptr("route_counts"), slab.ptr("pos")
        S = _lib.STAGES
        self._call("gs_frame_forward", f, S["GS_FWD_POSE_INVERSE"] | S["GS_FWD_FILTER_COMPACT"] | S["GS_FWD_PREPROCESS"] |
                   S["GS_FWD_ROUTE_COUNT"], dev)
        fr.owner, fr.owner_slab = f, slab
        fr.counts = slab.tensor("route_counts", torch.int32, (world,))
        fr.counters = slab.tensor("counters", torch.int32, (hip_ops.NUM_COUNTERS,))
        return fr

    # ------------------------------------------------------------------ phase A2: the send chunks
    def pack(self, fr: _Frame, capacity: int, n_visible: int) -> torch.Tensor:
        """-> send f32[world, capacity + 1, 16].  capacity: slots per chunk, the same on every rank (>= every count of
        every rank); n_visible: this rank's visible count (both known to the host after the one size read of the frame)."""
        f = fr.owner
        fr.capacity, fr.n_visible = int(capacity), int(n_visible)
        dev = fr.keep[0].device
        send = torch.empty((self.world, capacity + 1, hip_ops.ATTR_STRIDE), dtype=torch.float32, device=dev)
        f.chunk_capacity, f.route_send = int(capacity), send.data_ptr()
        self._call("gs_frame_forward", f, _lib.STAGES["GS_FWD_ROUTE_SCATTER"], dev)
        return send

    # ------------------------------------------------------------------ phase B: bin, sort, blend the received records
    def blend(self, fr: _Frame, received: torch.Tensor, need_state: bool = True, gather_in_place: bool = True):
        """received f32[world, capacity + 1, 16]: chunk s = what rank s sent to this band.  -> (image, depth, count):
        this rank's tile rows rendered (allocated so that the all-gather of the other bands runs in place)."""
        cfg = self.config
        width, height = fr.width, fr.height
        dev = received.device
        layout = self._layout(height, True)
        fr.records, fr.band_layout = received, layout
        n_rec = received.shape[0] * received.shape[1]
        num_bins = layout.num_bins(width, height)
        n_bins = (num_bins + 1) & ~1
        pixels = width * height
        tiles = (width // TILE) * (height // TILE)
        owned_tiles = hip_ops.num_owned_tiles(width, height, layout)
        rgb_only = bool(cfg.rgb_only)
        ordered = bool(self.ordered_dispatch)
        guess_key = (width, height, layout, cfg.near_plane, cfg.far_plane, cfg.depth_to_sort_key_scale)
        guess = self._size_guesses.get(guess_key) if self.speculative_sizes else None
        cap = int(guess[0]) if guess else 0
        shift2 = 2 * layout.bin_shift
        emit = bool(guess and need_state and self.backward_on_walked_lists and layout.filter != 0 and
                    hip_ops.can_emit_walked_lists(cap, layout.bin_shift))
        slab = Slab()
        slab.add("counters", 4 * hip_ops.NUM_COUNTERS)
        slab.add("ntiles", 4 * n_rec)
        slab.add("nkeys", 4 * n_rec)
        slab.add("ranges", 8 * n_bins)
        if need_state:
            slab.add("slot_offsets", 4 * n_rec)
            slab.add("acc_alpha", 4 * pixels)
            slab.add("last_eff", 4 * pixels)
            if ordered:
                slab.add("tile_work", 4 * owned_tiles)
        if emit:
            slab.add("walked_list", 4 * (max(cap, 1) << shift2))
            slab.add("walked_start", 4 * tiles)
        elif guess:
            slab.add("payload", 4 * max(cap, 1))
            slab.add("payload_alt", 4 * max(cap, 1))
        split = 0
        if guess and need_state and self.split_small_grid_backward:
            split = hip_ops.boundary_states_bytes(max(cap, 1) << (shift2 if emit else 0), width, height, layout, emit)
            if split:
                slab.add("boundary", split)
        slab.allocate(dev)
        lib = _lib.load()
        if gather_in_place and self.row_weights is None:   # (weighted bands are unequal blocks: gathered through a packed buffer)
            rows = padded_image_rows(height, self.world)
            image = torch.empty((rows, width, 3), dtype=torch.float32, device=dev)[:height]
            depth = None if rgb_only else torch.empty((rows, width), dtype=torch.float32, device=dev)[:height]
            count = None if rgb_only else torch.empty((rows, width), dtype=torch.int32, device=dev)[:height]
        else:
            image = torch.zeros((height, width, 3), dtype=torch.float32, device=dev)
            depth = None if rgb_only else torch.zeros((height, width), dtype=torch.float32, device=dev)
            count = None if rgb_only else torch.zeros((height, width), dtype=torch.int32, device=dev)
        if self._readback is None:
            self._readback = hip_ops.CounterReadback(dev)
            self._readback.event.record(torch.cuda.current_stream(dev))   # creates the event the library records by handle
        b = _lib.GsFrame()
        b.n_points, b.width, b.height = n_rec, width, height
        b.tile_row_begin, b.tile_row_step, b.tile_row_end = layout.row_begin, layout.row_step, layout.row_end
        b.bin_shift, b.exact_tile_cull = layout.bin_shift, int(layout.exact_cull)
        b.blend_flags = (hip_ops.BLEND_RGB_ONLY if rgb_only else 0) | (0 if need_state else hip_ops.BLEND_NO_STATE)
        b.need_state = int(bool(need_state))
        b.near_plane, b.far_plane, b.depth_scale = cfg.near_plane, cfg.far_plane, cfg.depth_to_sort_key_scale
        b.world, b.chunk_capacity, b.n_records = self.world, fr.capacity, n_rec
        b.records = received.data_ptr()
        b.counters = slab.ptr("counters")
        b.host_counters_pinned, b.size_event = self._readback.host.data_ptr(), self._readback.event.cuda_event
        b.num_overlap_tiles, b.num_keys, b.slot_offsets = slab.ptr("ntiles"), slab.ptr("nkeys"), slab.ptr("slot_offsets")
        nblk = (n_rec + 255) // 256
        b.block_sums = self._ws("b_block_sums", 4 * nblk, dev)
        b.block_sums_full = self._ws("b_block_sums_full", 4 * nblk, dev)
        b.bin_ranges, b.n_bins = slab.ptr("ranges"), n_bins
        b.image = image.data_ptr()
        b.depth = 0 if depth is None else depth.data_ptr()
        b.valid_count = 0 if count is None else count.data_ptr()
        b.acc_alpha, b.last_effective = slab.ptr("acc_alpha"), slab.ptr("last_eff")
        b.tile_order = self._ws("b_order_fwd", 4 * owned_tiles, dev) if ordered else 0
        b.tile_work = slab.ptr("tile_work")
        b.boundary_states = slab.ptr("boundary")
        if need_state and self.split_small_grid_backward:
            b.split_workspace = hip_ops.split_workspace(self._scratch, width, height, dev).data_ptr()
        if self.split_small_grid_forward:
            fsplit = hip_ops.forward_split_workspace(self._scratch, width, height, layout, dev)
            b.forward_split_workspace = 0 if fsplit is None else fsplit.data_ptr()
        S = _lib.STAGES
        stages = S["GS_FWD_COUNT_KEYS"] | S["GS_FWD_SCAN"] | S["GS_FWD_READ_SIZES"]
        if guess:   # the list stages and the blend, speculatively, behind the size read
            kdb, depth_bits, tile_bits = hip_ops.key_layout(cfg.near_plane, cfg.far_plane, cfg.depth_to_sort_key_scale,
                                                            num_bins, int(guess[1]))
            key_bytes = 4 if kdb > 0 else 8
            b.key_depth_bits, b.depth_bits, b.tile_bits, b.n_keys_capacity = kdb, depth_bits, tile_bits, cap
            b.keys = self._ws("b_keys", key_bytes * cap, dev)
            b.keys_alt = self._ws("b_keys_alt", key_bytes * cap, dev)
            if emit:
                b.payload, b.payload_alt = self._ws("b_payload", 4 * cap, dev), self._ws("b_payload_alt", 4 * cap, dev)
                b.walked_list, b.walked_start = slab.ptr("walked_list"), slab.ptr("walked_start")
            else:
                b.payload, b.payload_alt = slab.ptr("payload"), slab.ptr("payload_alt")
            b.sort_workspace = self._ws("b_sort", lib.gs_sort_workspace_bytes(cap), dev)
            stages |= S["GS_FWD_MAKE_KEYS"] | S["GS_FWD_SORT"] | S["GS_FWD_RANGES"] | S["GS_FWD_BLEND"]
        self._call("gs_frame_forward", b, stages, dev)
        host = self._readback.wait()
        n_keys, n_slots = host[hip_ops.COUNTER_NUM_KEYS], host[hip_ops.COUNTER_NUM_SLOTS]
        max_depth_key = host[hip_ops.COUNTER_MAX_DEPTH_KEY]
        if n_keys >= 0x7fffffff or n_slots >= 0x7fffffff:
            raise RuntimeError("more than 2^31-1 (tile, Gaussian) pairs: key offsets are int32 as in the reference")
        fits = bool(guess) and n_keys <= guess[0] and max_depth_key <= guess[1]
        self.speculation_stats["frames"] += 1
        self.speculation_stats["redone"] += 1 if (guess and not fits) else 0
        depth_bound = (1 << max(int(max_depth_key), 1).bit_length()) - 1
        if guess and depth_bound <= guess[1] <= 4 * depth_bound + 3:
            depth_bound = guess[1]
        if len(self._size_guesses) >= 16 and guess_key not in self._size_guesses:
            self._size_guesses.pop(next(iter(self._size_guesses)))
        self._size_guesses[guess_key] = (max(int(1.3 * n_keys) + 4096, int(0.99 * guess[0]) if guess else 0), depth_bound)
        fr.band, fr.band_slab, fr.walked = b, slab, emit
        if not fits:   # first frame, or the frame outgrew the speculative sizes: the list stages with exact sizes
            records = received.view(-1, hip_ops.ATTR_STRIDE)
            nkeys = slab.tensor("nkeys", torch.int32, (n_rec,))
            ntiles = slab.tensor("ntiles", torch.int32, (n_rec,))
            bsums = self._scratch.get("b_block_sums", max(4 * nblk, 16), torch.uint8, dev).view(torch.int32)[:nblk]
            bsums_full = self._scratch.get("b_block_sums_full", max(4 * nblk, 16), torch.uint8, dev).view(torch.int32)[:nblk]
            kdb, depth_bits, tile_bits = hip_ops.key_layout(cfg.near_plane, cfg.far_plane, cfg.depth_to_sort_key_scale,
                                                            num_bins, max_depth_key)
            keys, payload, slot_offsets = hip_ops.make_keys(
                records, nkeys, bsums, n_keys, width, height, cfg.depth_to_sort_key_scale, layout, kdb,
                ntiles if need_state else None, bsums_full if need_state else None, ws=self._scratch)
            keys, payload = hip_ops.sort_pairs(keys, payload, depth_bits, tile_bits, kdb, in_place=False, ws=self._scratch,
                                               bins_in_any_order=True)
            start, end = hip_ops.tile_ranges(keys, num_bins, kdb)
            del keys
            emit = bool(need_state and self.backward_on_walked_lists and layout.filter != 0 and
                        hip_ops.can_emit_walked_lists(payload.shape[0], layout.bin_shift))
            out = (image, depth,
                   slab.tensor("acc_alpha", torch.float32, (height, width)) if need_state else None,
                   slab.tensor("last_eff", torch.int32, (height, width)) if need_state else None, count)
            work = slab.tensor("tile_work", torch.int32, (owned_tiles,)) if (need_state and ordered) else None
            boundary = None
            if need_state and self.split_small_grid_backward:
                nbytes = hip_ops.boundary_states_bytes(max(payload.shape[0], 1) << (shift2 if emit else 0), width, height,
                                                       layout, emit)
                if nbytes:
                    boundary = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                    fr.keep.append(boundary)
            b.boundary_states = 0 if boundary is None else boundary.data_ptr()
            b.n_keys_capacity = int(payload.shape[0])   # (the list length the boundary buffer's layout is derived from)
            blended = hip_ops.blend_forward(start, end, payload, records, width, height, layout, out=out,
                                            rgb_only=rgb_only, need_state=need_state, ordered=ordered, tile_work=work,
                                            ws=self._scratch, emit_walked_lists=emit, boundary=boundary,
                                            split=self.split_small_grid_forward)
            if emit:
                start, payload = blended[5], blended[6]
            fr.keep += [start, payload, slot_offsets]
            b.list_start, b.list_payload = start.data_ptr(), payload.data_ptr()
            if slot_offsets is not None:
                b.slot_offsets = slot_offsets.data_ptr()
            fr.walked = emit
        elif emit:
            b.list_start, b.list_payload = b.walked_start, b.walked_list
        else:
            b.list_start = b.bin_ranges
            b.list_payload = b.payload_alt if b.sorted_in_alt else b.payload
        if rgb_only:
            depth = torch.zeros((height, width), dtype=torch.float32, device=dev)
            count = torch.zeros((height, width), dtype=torch.int32, device=dev)
        fr.n_slots, fr.n_keys = int(n_slots), int(n_keys)
        fr.layout_bwd = hip_ops.walked_layout(layout) if fr.walked else layout
        fr.outputs = (image, depth, count)
        # next frame's list layout: the single-GPU operator's rule -- its absolute thresholds on this band's key count scaled
        # to the whole image, its keys-per-Gaussian thresholds on the band's own keys and records (both are the band's: scaling
        # one of them made an ordinary G = 8 band of the headline frame, 2.1 keys per record, look like the stress distribution)
        owned = max(len(layout.owned_rows(height)), 1)
        k_frame = n_keys * (height // TILE) / owned
        m = max(n_rec, 1)
        used = layout.bin_shift
        if n_keys > 0:
            self._auto_bin_shift = hip_ops.next_bin_shift(used, k_frame, n_keys, m)
        fr.stats.update(records_received=n_rec, keys=fr.n_keys, slots=fr.n_slots, bin_shift=used,
                        speculative=bool(guess), fits=fits)
        return fr.outputs

    # ------------------------------------------------------------------ phase C: the band's pixels, backward
    def backward_band(self, fr: _Frame, grad_image: torch.Tensor) -> torch.Tensor:
        """-> f32[world, capacity + 1, 12]: chunk s = the accumulator rows of the records rank s sent, to be returned."""
        b = fr.band
        dev = grad_image.device
        width, height = fr.width, fr.height
        grad_image = grad_image.contiguous()
        if grad_image.dtype != torch.float32:
            raise TypeError("grad_rasterized_image must be float32")
        n_slots = max(fr.n_slots, 1)
        b.n_slots = fr.n_slots
        b.backward_bin_shift, b.backward_filter = fr.layout_bwd.bin_shift, fr.layout_bwd.filter
        b.grad_image = grad_image.data_ptr()
        b.partials = self._ws("b_partials", 48 * n_slots, dev)
        b.slot_flags = self._ws("b_slot_flags", (n_slots + 15) & ~15, dev)
        magnitude = torch.zeros((height, width, 2), dtype=torch.float32, device=dev)   # this rank's band, zeros elsewhere
        b.magnitude_image = magnitude.data_ptr()
        b.tile_order_backward = self._ws("b_order_bwd", 4 * hip_ops.num_owned_tiles(width, height, fr.band_layout), dev) \
            if b.tile_work else 0
        rows = torch.empty((self.world, fr.capacity + 1, hip_ops.ACC_STRIDE), dtype=torch.float32, device=dev)
        b.acc = rows.data_ptr()
        fr.keep.append(grad_image)
        S = _lib.STAGES
        self._call("gs_frame_backward", b, S["GS_BWD_BLEND"] | S["GS_BWD_REDUCE"], dev)
        fr.stats["magnitude_image"] = magnitude
        return rows

    # ------------------------------------------------------------------ phase D: the rank's own rows, backward
    def backward_points(self, fr: _Frame, returned: torch.Tensor):
        """returned f32[world, capacity + 1, 12]: chunk b = the rows band b produced for this rank's records.
        -> (grad_point_cloud [N_g,3], grad_point_cloud_features [N_g,56]); calls the hook with this rank's fields."""
        f, slab = fr.owner, fr.owner_slab
        m, n = fr.n_visible, f.n_points
        dev = returned.device
        hook = self.hook
        f.n_visible = m
        f.returned_rows = returned.data_ptr()
        f.acc = self._ws("o_acc", 48 * max(m, 1), dev)
        grad_xyz = torch.empty((n, 3), dtype=torch.float32, device=dev)
        grad_feat = torch.empty((n, hip_ops.FEATURE_DIM), dtype=torch.float32, device=dev)
        f.grad_xyz, f.grad_features = grad_xyz.data_ptr(), grad_feat.data_ptr()
        gx_vis = gf_vis = fields = None
        f.grad_xyz_visible = f.grad_features_visible = f.hook_compact = 0
        if hook is not None:
            gx_vis = torch.empty((m, 3), dtype=torch.float32, device=dev)
            compact = torch.empty(7 * m, dtype=torch.float32, device=dev)
            f.grad_xyz_visible, f.hook_compact = gx_vis.data_ptr(), compact.data_ptr()
            if self.hook_feature_gradients:
                gf_vis = torch.empty((m, hip_ops.FEATURE_DIM), dtype=torch.float32, device=dev)
                f.grad_features_visible = gf_vis.data_ptr()
            fields = dict(grad_viewspace=compact[0:2 * m].view(m, 2), magnitude_grad_viewspace=compact[2 * m:3 * m],
                          num_affected_pixels=compact[3 * m:4 * m].view(torch.int32), point_depth=compact[4 * m:5 * m],
                          point_uv_in_camera=compact[5 * m:7 * m].view(m, 2))
        S = _lib.STAGES
        self._call("gs_frame_backward", f, S["GS_BWD_GATHER_RETURNED"] | S["GS_BWD_POINTS"], dev)
        if hook is not None:   # RAS:1127-1142, with THIS RANK's rows: ids index the rank's block of the point cloud
            hook(_Op.BackwardValidPointHookInput(
                point_id_in_camera_list=slab.tensor("ids", torch.int32, (n,))[:m], grad_point_in_camera=gx_vis,
                grad_pointfeatures_in_camera=gf_vis,
                magnitude_grad_viewspace_on_image=fr.stats.get("magnitude_image"),   # this rank's band, zeros elsewhere
                num_overlap_tiles=slab.tensor("ntiles", torch.int32, (n,))[:m], **fields))
        return grad_xyz, grad_feat


def balanced_row_weights(row_work: Sequence[float], world: int, current: Optional[Sequence[float]] = None,
                         threshold: float = 1.15) -> Optional[list]:
    """The per-tile-row weights for the NEXT frames' band boundaries: ``row_work`` (the ranks' ``row_work`` summed: list
    positions walked per tile row) when the bands placed by ``current`` (None = equal bands) are out of balance by more
    than ``threshold`` (heaviest band / mean band), else ``current`` itself -- moving the boundaries restarts the bands'
    speculative sizes, so it is done only when it pays.  Every rank computes the same answer from the same numbers."""
    work = [float(w) for w in row_work]
    total = sum(work)
    if total <= 0.0 or world <= 1:
        return None if current is None else list(current)
    bounds = band_boundaries(len(work), world, current)
    heaviest = max(sum(work[bounds[g]:bounds[g + 1]]) for g in range(world))
    if heaviest * world <= threshold * total:
        return None if current is None else list(current)
    return work


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
        # every `rebalance_every`-th frame with state the ranks sum their per-tile-row walk lengths (one all-reduce of a
        # few hundred floats + one host read) and move the band boundaries if the heaviest band carries more than
        # `rebalance_threshold` times the mean (0 = equal bands for ever)
        self.rebalance_every = 0
        self.rebalance_threshold = 1.15
        self._frames_with_state = 0
        # the chunk capacity of the exchange is SPECULATED from the previous frame (15 % head-room, decaying 1 % per frame
        # as the key capacities do): the gathered sizes travel to pinned memory behind an event while the host keeps
        # launching pack, all-to-all and the band's stages, and are looked at where the host waits for the band's own sizes
        # anyway -- no host stall before the exchange.  A frame whose records do not fit (gs_route_scatter drops what does
        # not, the counts tell) repeats pack, exchange and blend with the exact capacity; every rank sees the same sizes
        # and takes the same decision.  False: read the sizes first (one blocking read per frame)
        self.speculative_capacity = True
        self._capacity_guess = 0
        self._sizes_host = None      # pinned int32[world, world + 1]
        self._sizes_event = None
        self.capacity_stats = {"frames": 0, "redone": 0}
        outer = self

        class _fn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, point_cloud, point_cloud_features, input_data, need_state):
                core = outer.core
                with _lib.stream_scope(point_cloud.device):
                    f = core.project(input_data, need_state)
                    # ONE size exchange per frame: every rank's per-band counts and visible count (the chunk capacity
                    # must be the same everywhere)
                    mine = torch.cat([f.counts, f.counters[:1]])   # int32: records for every band, visible count
                    sizes = torch.empty((outer.world, outer.world + 1), dtype=torch.int32, device=mine.device)
                    dist.all_gather_into_tensor(sizes.view(-1), mine, group=outer.group)
                    speculate = outer.speculative_capacity and outer._capacity_guess > 0
                    if speculate:
                        if outer._sizes_host is None:
                            outer._sizes_host = torch.empty((outer.world, outer.world + 1), dtype=torch.int32).pin_memory()
                            outer._sizes_event = torch.cuda.Event()
                        outer._sizes_host.copy_(sizes, non_blocking=True)
                        outer._sizes_event.record(torch.cuda.current_stream(mine.device))
                        capacity = outer._capacity_guess
                        # (what blend() learns from a frame -- size guesses, the automatic bin shift, its counters -- is set
                        # aside: a frame whose exchange turns out truncated is blended again below, and must not leave the
                        # truncated frame's sizes behind nor be counted twice)
                        learnt = (dict(core._size_guesses), core._auto_bin_shift, dict(core.speculation_stats))
                        send = core.pack(f, capacity, -1)
                        received = _all_to_all_chunks(send, outer.group)
                        image, depth, count = core.blend(f, received, need_state)   # (ends waiting for the band's sizes)
                        outer._sizes_event.synchronize()
                        host = outer._sizes_host.clone()
                        f.n_visible = int(host[outer.rank, outer.world])   # (known from here on, whatever follows)
                    else:
                        host = sizes.cpu()
                    needed = int(host[:, :outer.world].max())
                    outer.capacity_stats["frames"] += 1
                    if not speculate or needed > capacity:   # first frame, or the records outgrew the speculated chunks
                        outer.capacity_stats["redone"] += 1 if speculate else 0
                        if speculate:
                            core._size_guesses, core._auto_bin_shift = learnt[0], learnt[1]
                            core.speculation_stats.update(learnt[2])
                        capacity = _chunk_capacity(needed)
                        send = core.pack(f, capacity, int(host[outer.rank, outer.world]))
                        received = _all_to_all_chunks(send, outer.group)
                        image, depth, count = core.blend(f, received, need_state)
                    f.n_visible = int(host[outer.rank, outer.world])
                    outer._capacity_guess = max(_chunk_capacity(int(1.15 * needed) + 64),
                                                _chunk_capacity(int(0.99 * outer._capacity_guess)))
                    outs = [image] if core.config.rgb_only else [image, depth, count]
                    from .distributed import all_gather_tile_rows
                    all_gather_tile_rows(outs, outer.rank, outer.world, outer.group,
                                         row_weights=core.weights_for(image.shape[0]))
                    if need_state and outer.rebalance_every > 0:
                        outer._frames_with_state += 1
                        if outer._frames_with_state % outer.rebalance_every == 0:
                            work = core.row_work(f)
                            if work is not None:   # (the same on every rank: options and need_state agree)
                                dist.all_reduce(work, op=dist.ReduceOp.SUM, group=outer.group)
                                core.row_weights = balanced_row_weights(work.tolist(), outer.world,
                                                                        core.weights_for(image.shape[0]),
                                                                        outer.rebalance_threshold)
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
                with _lib.stream_scope(f.keep[0].device):
                    if grad_image is None:
                        grad_image = torch.zeros((f.height, f.width, 3), dtype=torch.float32, device=f.keep[0].device)
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
                   timings: Optional[dict] = None, row_work: Optional[list] = None):
    """Plays one frame of ``len(cores)`` ranks on ONE device: every rank's four phases exactly as under
    torch.distributed, the exchanges as device copies between the ranks' buffers.  -> (image, depth, count, grads):
    the assembled frame and, when ``grad_image`` is given, per rank (grad_point_cloud, grad_point_cloud_features).
    timings (optional dict): filled with HIP-event times per rank and phase in ms -- what a rank's GPU does in a frame,
    exchanges excluded.  row_work (optional list): receives the ranks' ``row_work`` summed (a float list per tile row, or
    nothing when the frame kept no state) -- what ``balanced_row_weights`` turns into the next frames' boundaries."""
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
    host = torch.stack([torch.cat([f.counts, f.counters[:1]]) for f in frames]).cpu()
    capacity = _chunk_capacity(int(host[:, :world].max()))
    sends = [timed("pack", g, lambda g=g: cores[g].pack(frames[g], capacity, int(host[g, world]))) for g in range(world)]
    received = [torch.stack([sends[s][g] for s in range(world)]) for g in range(world)]   # the all-to-all
    outs = [timed("blend", g, lambda g=g: cores[g].blend(frames[g], received[g], grad_image is not None,
                                                          gather_in_place=False)) for g in range(world)]
    height = frames[0].height
    if row_work is not None:
        works = [cores[g].row_work(frames[g]) for g in range(world)]
        if all(w is not None for w in works):
            row_work[:] = torch.stack(works).sum(dim=0).tolist()
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
        timings["speculation"] = [dict(c.speculation_stats) for c in cores]
    return image, depth, count, grads
