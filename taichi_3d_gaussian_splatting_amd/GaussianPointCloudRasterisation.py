"""Drop-in mirror of the reference operator ``GaussianPointCloudRasterisation``.

Reference: ``taichi_3d_gaussian_splatting/GaussianPointCloudRasterisation.py`` (RAS):
``nn.Module`` + inner ``torch.autograd.Function`` RAS:775-1204, nested dataclasses
``GaussianPointCloudRasterisationConfig`` RAS:776-786, ``...Input`` RAS:788-804,
``BackwardValidPointHookInput`` RAS:806-817.  Same names, argument meaning, outputs
(image[H,W,3] f32, depth[H,W] f32, pixel_valid_point_count[H,W] i32), gradients (dense [N,3] and
[N,56], ``None`` for the other inputs), hook payload, assertion on the image size, and the same
side effect (in-place quaternion normalisation of visible rows, RAS:196-205,264).

Every stage runs in the hand-written HIP library behind the C ABI of ``include/gsplat_hip.h``
(no Taichi, no eager-PyTorch arithmetic, no CPU path).  Where the reference leaves outputs
uninitialised (no Gaussian on screen, RAS:967-980) this operator returns zeros.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Callable, Optional

import torch

from . import _lib, hip_ops
from .Camera import CameraInfo

BOUNDARY_TILES = 3   # RAS:26
TILE_WIDTH = 16      # RAS:27
TILE_HEIGHT = 16     # RAS:28


def find_tile_start_and_end(point_in_camera_sort_key: torch.Tensor, tile_points_start: torch.Tensor,
                            tile_points_end: torch.Tensor) -> None:
    """Same call shape as the reference kernel RAS:175-193 (int64 keys (tile << 32) + depth): fills the (pre-zeroed)
    ranges in place."""
    start, end = hip_ops.tile_ranges(point_in_camera_sort_key.contiguous(), tile_points_start.shape[0])
    tile_points_start.copy_(start)
    tile_points_end.copy_(end)


@dataclass
class GaussianPoint3DRow:
    """What ``load_point_cloud_row_into_gaussian_point_3d`` returns when no Taichi struct is available: the seven
    members of the reference's ``GaussianPoint3D`` (GP3:97-109), as tensors/arrays sliced out of the two inputs."""
    translation: object  # xyz
    cov_rotation: object  # quaternion x, y, z, w (feature columns 0-3)
    cov_scale: object  # log-scale (columns 4-6)
    alpha: object  # opacity logit (column 7)
    color_r: object  # 16 SH coefficients (columns 8-23)
    color_g: object  # columns 24-39
    color_b: object  # columns 40-55


def _build_row_loader():
    """RAS:208-236.  The reference's controller imports this helper next to the operator (ADC:4) and calls it inside
    its two Taichi kernels (ADC:10-42), so a drop-in module has to export it.  When Taichi and the host package's
    ``GaussianPoint3D`` struct are importable (this module registered as ``taichi_3d_gaussian_splatting.
    GaussianPointCloudRasterisation``) it is a ``ti.func`` producing that struct; otherwise
    a plain function producing ``GaussianPoint3DRow``.  Column layout: RAS:214-226."""
    columns = dict(cov_rotation=(0, 4), cov_scale=(4, 7), color_r=(8, 24), color_g=(24, 40), color_b=(40, 56))
    try:
        import importlib
        import taichi as ti
        host = "taichi_3d_gaussian_splatting"   # the package this module is injected into
        point_struct = importlib.import_module(host + ".GaussianPoint3D").GaussianPoint3D
        vec16f = importlib.import_module(host + ".SphericalHarmonics").vec16f
    except Exception:  # no Taichi / no host package: the plain-Python form
        def load_point_cloud_row_into_gaussian_point_3d(pointcloud, pointcloud_features, point_id):
            row = pointcloud_features[point_id]
            parts = {name: row[a:b] for name, (a, b) in columns.items()}
            return GaussianPoint3DRow(translation=pointcloud[point_id], alpha=row[7], **parts)
        return load_point_cloud_row_into_gaussian_point_3d

    @ti.func
    def load_point_cloud_row_into_gaussian_point_3d(
            pointcloud: ti.types.ndarray(ti.f32, ndim=2),  # (N, 3)
            pointcloud_features: ti.types.ndarray(ti.f32, ndim=2),  # (N, 56)
            point_id: ti.i32):
        return point_struct(
            translation=ti.math.vec3([pointcloud[point_id, c] for c in ti.static(range(3))]),
            cov_rotation=ti.math.vec4([pointcloud_features[point_id, c] for c in ti.static(range(0, 4))]),
            cov_scale=ti.math.vec3([pointcloud_features[point_id, c] for c in ti.static(range(4, 7))]),
            alpha=pointcloud_features[point_id, 7],
            color_r=vec16f([pointcloud_features[point_id, c] for c in ti.static(range(8, 24))]),
            color_g=vec16f([pointcloud_features[point_id, c] for c in ti.static(range(24, 40))]),
            color_b=vec16f([pointcloud_features[point_id, c] for c in ti.static(range(40, 56))]))
    return load_point_cloud_row_into_gaussian_point_3d


def __getattr__(name):
    # PEP 562: built on first use -- by then the importing package (the reference's controller, ADC:4) is known and its
    # Taichi struct can be imported; building it at module import would import the host package half-initialised.
    if name == "load_point_cloud_row_into_gaussian_point_3d":
        fn = _build_row_loader()
        globals()[name] = fn
        return fn
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


__all__ = ["BOUNDARY_TILES", "TILE_WIDTH", "TILE_HEIGHT", "CameraInfo", "GaussianPoint3DRow",
           "GaussianPointCloudRasterisation", "find_tile_start_and_end", "load_point_cloud_row_into_gaussian_point_3d"]


class GaussianPointCloudRasterisation(torch.nn.Module):
    @dataclass
    class GaussianPointCloudRasterisationConfig:
        near_plane: float = 0.8
        far_plane: float = 1000.
        depth_to_sort_key_scale: float = 100.
        rgb_only: bool = False
        # un-annotated => class attributes, not dataclass fields, exactly as RAS:782-786:
        # YAML/ctor values for these are ignored by the reference as well.
        grad_color_factor = 5.
        grad_high_order_color_factor = 1.
        grad_s_factor = 0.5
        grad_q_factor = 1.
        grad_alpha_factor = 20.

    @dataclass
    class GaussianPointCloudRasterisationInput:
        point_cloud: torch.Tensor  # Nx3
        point_cloud_features: torch.Tensor  # Nx56
        point_object_id: torch.Tensor  # N, int32, in [0, K-1]
        point_invalid_mask: torch.Tensor  # N, int8
        camera_info: CameraInfo
        q_pointcloud_camera: torch.Tensor  # Kx4 (x,y,z,w), camera -> pointcloud
        t_pointcloud_camera: torch.Tensor  # Kx3
        color_max_sh_band: int = 2

    @dataclass
    class BackwardValidPointHookInput:
        point_id_in_camera_list: torch.Tensor  # M
        grad_point_in_camera: torch.Tensor  # Mx3
        grad_pointfeatures_in_camera: Optional[torch.Tensor]  # Mx56 (None when hook_feature_gradients is off)
        grad_viewspace: torch.Tensor  # Mx2
        magnitude_grad_viewspace: torch.Tensor  # M
        magnitude_grad_viewspace_on_image: torch.Tensor  # HxWx2
        num_overlap_tiles: torch.Tensor  # M
        num_affected_pixels: torch.Tensor  # M
        point_depth: torch.Tensor  # M
        point_uv_in_camera: torch.Tensor  # Mx2

    def __init__(
        self,
        config: "GaussianPointCloudRasterisation.GaussianPointCloudRasterisationConfig",
        backward_valid_point_hook: Optional[Callable[
            ["GaussianPointCloudRasterisation.BackwardValidPointHookInput"], None]] = None,
    ):
        super().__init__()
        self.config = config
        # image-space sharding (multi-GPU): this instance renders tile rows begin, begin+step, ... below end; either set
        # directly or derived per frame from `shard` = (rank, world, "bands" | "interleaved") by
        # distributed.shard_rasteriser_across_tile_rows (`shard_row_weights`: optional per-tile-row weights, the same on
        # every rank, that balance the bands)
        self.tile_row_begin = 0
        self.tile_row_step = 1
        self.tile_row_end = hip_ops._NO_ROW_LIMIT
        self.shard = None
        self.shard_row_weights = None
        # drop (bin | tile, Gaussian) pairs that cannot reach alpha >= 1/255 anywhere in the bin | tile (output-identical)
        self.exact_tile_cull = True
        # sort keys per bin of (1 << bin_shift)^2 tiles: 0 = per tile as the reference (fastest on small frames), 1 = 32 x 32
        # pixels, 2 = 64 x 64 pixels (the blend kernels recover each tile's list from its bin's list, in order; the more
        # tiles a Gaussian covers the larger the bin that pays), None = chosen per frame from the previous frame's sizes
        self.bin_shift: Optional[int] = None
        self._auto_bin_shift = 0
        self._auto_bin_shift_by_size = {}
        # launch the list stages from device-side counts with the previous frame's capacities instead of waiting for
        # this frame's sizes (see _forward); False = wait for the sizes first (two dependent halves, as round 1)
        self.speculative_sizes = True
        # tiles are handed to the hardware longest list / longest backward walk first instead of in image order (same
        # results; the launches lose their tail of half-empty CUs)
        self.ordered_dispatch = True
        # sum the backward's (Gaussian, tile) slot records inside the per-point kernel instead of a launch of their own
        # (same bits).  Measured slower at the headline size (0.186 vs 0.069 + 0.101 ms: the gather wants more waves in
        # flight than the per-point kernel can hold), so off by default
        self.fused_slot_reduction = False
        # binned layouts: the backward pass walks the per-tile lists the forward pass wrote out while filtering its bin's
        # list, instead of filtering the bin's list a second time (same entries in the same order: the same bits on grids
        # above 3,840 tiles; below, the per-tile lists let the backward take the four-waves-per-tile form, whose slot sums
        # add the same per-pixel terms in another order)
        self.backward_on_walked_lists = True
        # per-pass entry points: the blend kernels in the forms that suit the frame's walk lengths.  Evenly loaded frames: two
        # waves per tile, the backward's sums through LDS (throughput).  Frames where a few tiles walk many times the mean
        # (trained scenes) last as long as those tiles' chains: the forward pass with four waves per tile (-14 % of the
        # kernel; bit-identical outputs), the backward's sums in registers (-11 %).  The walk lengths are sampled every 16th
        # frame without a host wait (frame_path.walk_skew).  Same decisions either way, sums equal to rounding
        self.backward_form_by_walk_skew = os.environ.get("GS_BWD_FORM_BY_SKEW", "1") != "0"   # (the switch: A/B measurements)
        # the forward writes a normalised quaternion back only when the stored one differs (RAS:196-205: same memory
        # contents).  True = always write: what a training iteration pays -- the optimiser has just moved q -- for
        # benchmarks that time a static scene (bench.py)
        self.always_store_normalised_rotation = False
        # speculative frames go through ONE C call per pass (gs_frame_forward / gs_frame_backward: the library issues the
        # ~25 + ~5 launches back to back) instead of one foreign call per stage: same kernels, same arguments, same bits;
        # what changes is the host time per frame -- which bounds small frames.  False = stage by stage (hip_ops)
        self.frame_entry_points = True
        # grids that cannot fill the chip (at most 3840 rendered tiles: down-sampled training frames, small views, a rank's
        # band): the forward pass leaves per-pixel boundary states every 128 list entries and the backward pass gives a
        # tile up to four workgroups (include/gsplat_hip.h "List splitting").  Same slot records up to rounding
        self.split_small_grid_backward = True
        # ... and so does the forward pass (probe / blend / combine: gs_blend_forward_split)
        self.split_small_grid_forward = True
        # per-pass entry points only: the colours of the visible Gaussians (RAS:280-282,302-310 -- a streaming read of the
        # 192 B of SH coefficients each) evaluated on a second stream BESIDE key generation, sort and ranges, which do not
        # read them; the blend waits for them.  Same device code as inside gs_preprocess: the same bits (tested).  MEASURED
        # SLOWER (headline 1.083 -> 1.098 ms, trained scene 0.808 -> 0.819, cfg 1 0.242 -> 0.252: the fork / join through
        # two events costs ~10 us on this runtime and the list stages slow down by what the colours save), so off
        self.colours_beside_list_stages = False
        self._aux_streams = {}   # device -> (second stream, fork event, join event)
        self._scratch = hip_ops.Workspaces()   # buffers that do not outlive a call, kept between frames
        self._size_guesses, self._readbacks = {}, {}   # (image size, list layout, planes) -> (key capacity, depth bound)
        self.speculation_stats = {"frames": 0, "redone": 0}
        outer = self

        class _module_function(torch.autograd.Function):

            @staticmethod
            def forward(ctx, pointcloud, *rest):
                if not pointcloud.is_cuda:
                    raise RuntimeError("GaussianPointCloudRasterisation needs tensors on a HIP device (no CPU path)")
                with _lib.stream_scope(pointcloud.device):   # one stream look-up for the ~25 launches below
                    return _module_function._forward(ctx, pointcloud, *rest)

            @staticmethod
            def backward(ctx, *grads):
                if not ctx.saved_tensors:   # forward ran without backward state (nothing differentiable was asked for)
                    return (None,) * 9
                with _lib.stream_scope(ctx.saved_tensors[0].device):
                    state = getattr(ctx, "frame_state", None)
                    if state is None:
                        return _module_function._backward(ctx, *grads)
                    if not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):  # RAS:1028
                        return (None,) * 9
                    from . import frame_path
                    grad_image = grads[0]
                    if grad_image is None:  # only depth was used downstream: its gradient is ignored
                        grad_image = torch.zeros((state.height, state.width, 3), dtype=torch.float32,
                                                 device=ctx.saved_tensors[0].device)
                    grad_xyz, grad_feat = frame_path.backward(
                        outer, state, grad_image, backward_valid_point_hook,
                        GaussianPointCloudRasterisation.BackwardValidPointHookInput,
                        backward_valid_point_hook is not None and outer.hook_feature_gradients)
                    return grad_xyz, grad_feat, None, None, None, None, None, None, None

            @staticmethod
            def _forward(ctx, pointcloud, pointcloud_features, point_invalid_mask, point_object_id,
                         q_pointcloud_camera, t_pointcloud_camera, camera_info, color_max_sh_band, need_state):
                cfg = outer.config
                width, height = camera_info.camera_width, camera_info.camera_height
                layout = outer.list_layout(height, width)
                if not pointcloud_features.is_contiguous():
                    raise ValueError("point_cloud_features must be contiguous (it is normalised in place)")
                if pointcloud_features.dtype != torch.float32 or pointcloud_features.shape[1] != 56:
                    raise TypeError("point_cloud_features must be float32 [N,56]")
                xyz = pointcloud.contiguous()
                invalid = point_invalid_mask.to(torch.int8).contiguous()
                obj = point_object_id.to(torch.int32).contiguous()
                intrinsics = camera_info.camera_intrinsics.to(device=xyz.device, dtype=torch.float32).contiguous()
                q_pc = q_pointcloud_camera.to(torch.float32).contiguous()
                t_pc = t_pointcloud_camera.to(torch.float32).contiguous()

                rgb_only = bool(cfg.rgb_only)
                # multi-GPU with the default (un-weighted) bands: outputs are allocated so that the all-gather of the
                # other ranks' rows runs in place (distributed.all_gather_tile_rows)
                gathered_rows = 0
                if outer.image_gather is not None and outer.shard is not None and outer.shard[2] == "bands" and \
                        outer.shard_row_weights is None:
                    from .distributed import padded_image_rows
                    gathered_rows = padded_image_rows(height, outer.shard[1])
                guess_key = (width, height, layout, cfg.near_plane, cfg.far_plane, cfg.depth_to_sort_key_scale)
                # one guess per (image size, layout, planes): data sets that mix resolutions keep speculating
                guess = outer._size_guesses.get(guess_key) if outer.speculative_sizes else None
                if guess is not None and outer.frame_entry_points and not outer.fused_slot_reduction and \
                        q_pc.dim() == 2 and t_pc.dim() == 2 and q_pc.shape[0] == t_pc.shape[0] >= 1 and \
                        q_pc.shape[1] == 4 and t_pc.shape[1] == 3:
                    # the whole pass through ONE foreign call (frame_path.py); a frame that does not fit its speculative
                    # capacities falls through to the stage-by-stage path below and is redone with exact sizes
                    from . import frame_path
                    image, depth, count, state, host = frame_path.forward(
                        outer, xyz, pointcloud_features, invalid, obj, intrinsics, q_pc, t_pc, camera_info,
                        color_max_sh_band, need_state, layout, guess, gathered_rows, outer._counter_readback(xyz.device))
                    if outer._sizes_arrived(host, layout, width, height, guess, guess_key):
                        if rgb_only:
                            depth = torch.zeros((height, width), dtype=torch.float32, device=xyz.device)
                            count = torch.zeros((height, width), dtype=torch.int32, device=xyz.device)
                        if outer.image_gather is not None:
                            outer.image_gather([image] if rgb_only else [image, depth, count])
                        ctx.mark_non_differentiable(count)
                        if need_state:
                            # (a split backward pass reads the forward's image: saved as an output, in-place changes are caught)
                            ctx.save_for_backward(xyz, pointcloud_features, state.slab.buf, obj, intrinsics, t_pc,
                                                  image if state.split else None)
                            ctx.frame_state = state
                            ctx.camera_info = camera_info
                            ctx.set_materialize_grads(False)
                        return image, depth, count
                    # The frame did not fit: its per-point products (compaction, projection, counts, scans -- everything
                    # up to the size read) are good and are KEPT; only the list stages are redone below with exact sizes.
                    # (Running the projection again would normalise the stored quaternions a second time: q / |q| of an
                    # already normalised q can move by an ulp, and the frame would differ from a stage-by-stage one.)
                    n_pts, slab, ws_ = xyz.shape[0], state.slab, outer._scratch
                    visible_mask = slab.tensor("visible_mask", torch.int8, (n_pts,))
                    ids = slab.tensor("ids", torch.int32, (n_pts,))
                    attrs = slab.tensor("attrs", torch.float32, (n_pts, hip_ops.ATTR_STRIDE))
                    num_overlap_tiles = slab.tensor("ntiles", torch.int32, (n_pts,))
                    num_owned_tiles = slab.tensor("nkeys", torch.int32, (n_pts,))
                    q_cp = slab.tensor("q_cp", torch.float32, (q_pc.shape[0], 4))
                    t_cp = slab.tensor("t_cp", torch.float32, (q_pc.shape[0], 3))
                    nblk_ = (n_pts + 255) // 256
                    block_sums = ws_.get("f_block_sums", max(4 * nblk_, 16), torch.uint8, xyz.device).view(torch.int32)
                    block_sums_full = ws_.get("f_block_sums_full", max(4 * nblk_, 16), torch.uint8, xyz.device).view(torch.int32)
                    guess, frame_host = None, host
                else:
                    frame_host = None
                if frame_host is None:
                    # RAS:845  (q,t)_camera<-pointcloud
                    q_cp, t_cp = hip_ops.pose_inverse(q_pc, t_pc)
                    # RAS:848-870  frustum filter + ordered compaction; M stays on the device
                    visible_mask, ids, counters = hip_ops.filter_compact(
                        xyz, invalid, obj, intrinsics, q_cp, t_cp, cfg.near_plane, cfg.far_plane, width, height,
                        sync=False, ws=outer._scratch)
                    # RAS:887-911  per-point projection + tile counts (launched for the capacity N)
                    attrs, num_overlap_tiles, num_owned_tiles, block_sums, block_sums_full = hip_ops.preprocess(
                        xyz, pointcloud_features, obj, intrinsics, q_cp, t_cp, ids, width, height, layout,
                        cfg.depth_to_sort_key_scale, counters, n_visible_on_device=True,
                        always_store_rotation=outer.always_store_normalised_rotation, ws=outer._scratch)
                    # RAS:913-922  scans.  The reference blocks twice on sizes (RAS:870,916); here the one read-back of
                    # M, K, the slot count and the depth range travels to pinned memory while the host keeps launching:
                    # key generation, sort, ranges and the blend run SPECULATIVELY from the device-side counts with the
                    # capacities learnt from the previous frame, and are redone with exact sizes in the rare frame that
                    # does not fit (first frame, jump in the number of keys or in the depth range).
                    hip_ops.scan_block_sums_async(block_sums, counters, block_sums_full)
                    readback = outer._counter_readback(xyz.device)
                    readback.start(counters)
                num_bins = layout.num_bins(width, height)

                def lists_and_blend(attrs_, nkeys_, bsums_, bsums_full_, ntiles_, n_keys_, max_depth_key_, counters_):
                    # RAS:927-945 keys (one per (bin, Gaussian)), RAS:947-950 stable sort, RAS:952-964 list ranges,
                    # RAS:967-997 blend.  rgb_only (RAS:464-469,478-484): depth and count are not computed -- the
                    # reference returns uninitialised memory for them, this operator zeros.  The state the backward
                    # pass reads (acc_alpha, last_effective) is produced whenever a gradient can be asked for -- also
                    # with rgb_only, where the reference's backward would read garbage -- and skipped otherwise.
                    kdb, depth_bits, tile_bits = hip_ops.key_layout(
                        cfg.near_plane, cfg.far_plane, cfg.depth_to_sort_key_scale, num_bins, max_depth_key_)
                    n_dev = None if counters_ is None else counters_[hip_ops.COUNTER_NUM_KEYS:hip_ops.COUNTER_NUM_KEYS + 1]
                    keys, payload_, slot_offsets_ = hip_ops.make_keys(
                        attrs_, nkeys_, bsums_, n_keys_, width, height, cfg.depth_to_sort_key_scale, layout, kdb,
                        ntiles_ if need_state else None, bsums_full_ if need_state else None, counters=counters_,
                        ws=outer._scratch)
                    keys, payload_ = hip_ops.sort_pairs(keys, payload_, depth_bits, tile_bits, kdb, in_place=False,
                                                        n_keys_device=n_dev, ws=outer._scratch, bins_in_any_order=True)
                    start_, end_ = hip_ops.tile_ranges(keys, num_bins, kdb, n_keys_device=n_dev)
                    del keys
                    work_ = None
                    if need_state and outer.ordered_dispatch:
                        work_ = torch.empty(hip_ops.num_owned_tiles(width, height, layout), dtype=torch.int32,
                                            device=attrs_.device)
                    # binned layouts: the forward pass writes out every tile's own list as far as it walks it, and the
                    # backward pass runs on those plain per-tile lists (no second filtering of the bin's entries)
                    emit = bool(need_state and outer.backward_on_walked_lists and layout.filter != 0 and
                                hip_ops.can_emit_walked_lists(payload_.shape[0], layout.bin_shift))
                    boundary_ = None
                    if need_state and outer.split_small_grid_backward:
                        nbytes = hip_ops.boundary_states_bytes(payload_.shape[0] << (2 * layout.bin_shift if emit else 0),
                                                               width, height, layout, emit)
                        if nbytes:
                            boundary_ = torch.empty(nbytes, dtype=torch.uint8, device=attrs_.device)
                    blended = hip_ops.blend_forward(start_, end_, payload_, attrs_, width, height, layout,
                                                    rgb_only=rgb_only, need_state=need_state,
                                                    gathered_rows=gathered_rows, ordered=outer.ordered_dispatch,
                                                    tile_work=work_, ws=outer._scratch, emit_walked_lists=emit,
                                                    boundary=boundary_, split=outer.split_small_grid_forward)
                    if emit:   # what the backward pass walks: (list starts, list) of the emitted per-tile lists
                        start_, payload_, blended = blended[5], blended[6], blended[:5]
                    return payload_, slot_offsets_, start_, blended, work_, emit, boundary_

                result = None
                if guess is not None:
                    result = lists_and_blend(attrs, num_owned_tiles, block_sums, block_sums_full, num_overlap_tiles,
                                             guess[0], guess[1], counters)
                host = frame_host if frame_host is not None else readback.wait()
                m, n_keys, n_slots = (host[hip_ops.COUNTER_NUM_VISIBLE], host[hip_ops.COUNTER_NUM_KEYS],
                                      host[hip_ops.COUNTER_NUM_SLOTS])
                max_depth_key = host[hip_ops.COUNTER_MAX_DEPTH_KEY]
                nb = (m + 255) // 256
                # (a frame that came back from gs_frame_forward too large has been accounted for already)
                fits = False if frame_host is not None else outer._sizes_arrived(host, layout, width, height, guess, guess_key)
                ids, attrs, num_overlap_tiles, num_owned_tiles = ids[:m], attrs[:m], num_overlap_tiles[:m], \
                    num_owned_tiles[:m]
                if not fits:
                    result = lists_and_blend(attrs, num_owned_tiles, block_sums[:nb], block_sums_full[:nb],
                                             num_overlap_tiles, n_keys, max_depth_key, None)
                payload, slot_offsets, tile_start, (image, depth, acc_alpha, last_eff, count), tile_work, walked, \
                    boundary = result
                if slot_offsets is not None:
                    slot_offsets = slot_offsets[:m]
                if rgb_only:
                    depth = torch.zeros((height, width), dtype=torch.float32, device=xyz.device)
                    count = torch.zeros((height, width), dtype=torch.int32, device=xyz.device)
                if outer.image_gather is not None:  # multi-GPU: all-gather the tile rows of the other ranks
                    outer.image_gather([image] if rgb_only else [image, depth, count])
                if not need_state:   # nothing to save: no backward pass will run
                    ctx.mark_non_differentiable(count)
                    return image, depth, count

                ctx.save_for_backward(xyz, pointcloud_features, payload, ids, tile_start, acc_alpha,
                                      last_eff, num_overlap_tiles, obj, q_cp, t_cp, t_pc, attrs, intrinsics,
                                      slot_offsets, visible_mask, num_owned_tiles, boundary,
                                      image if boundary is not None else None)
                ctx.tile_work = tile_work
                ctx.layout_bwd = hip_ops.walked_layout(layout) if walked else layout
                ctx.n_slots = n_slots
                ctx.camera_info = camera_info
                ctx.color_max_sh_band = color_max_sh_band
                ctx.layout = layout
                ctx.mark_non_differentiable(count)
                ctx.set_materialize_grads(False)  # no zero-filled dL/ddepth (it is ignored, RAS:1027)
                return image, depth, count

            @staticmethod
            def _backward(ctx, grad_rasterized_image, grad_rasterized_depth, grad_pixel_valid_point_count):
                grad_pointcloud = grad_pointcloud_features = None
                if grad_rasterized_image is None:  # only depth was used downstream: its gradient is ignored
                    grad_rasterized_image = torch.zeros(
                        (ctx.camera_info.camera_height, ctx.camera_info.camera_width, 3), dtype=torch.float32,
                        device=ctx.saved_tensors[0].device)
                if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:  # RAS:1028
                    (xyz, features, payload, ids, tile_start, acc_alpha, last_eff, num_overlap_tiles,
                     obj, q_cp, t_cp, t_pc, attrs, intrinsics, slot_offsets, visible_mask,
                     num_owned_tiles, boundary, image) = ctx.saved_tensors
                    cfg = outer.config
                    camera_info = ctx.camera_info
                    width, height = camera_info.camera_width, camera_info.camera_height
                    hook = backward_valid_point_hook
                    # RAS:531-705  per-pixel pass: one 48-B record per (Gaussian, tile) slot, no atomics
                    partials, slot_flags, magnitude_image = hip_ops.blend_backward_partials(
                        tile_start, payload, attrs, grad_rasterized_image, acc_alpha, last_eff, slot_offsets,
                        ctx.n_slots, width, height, ctx.layout_bwd, tile_work=ctx.tile_work, ws=outer._scratch,
                        image=image, boundary=boundary)
                    acc = slots = None
                    if outer.grad_accumulator_reduce is None and outer.fused_slot_reduction:
                        # the slot sums are formed inside the per-point kernel and stay in registers
                        slots = (slot_offsets, num_overlap_tiles, slot_flags, partials)
                    else:   # the per-Gaussian sums go through memory (multi-GPU: they are summed over the ranks)
                        acc = hip_ops.reduce_partials(slot_offsets, num_overlap_tiles, slot_flags, partials,
                                                      num_owned_tiles if ctx.layout.sharded else None, attrs, width, height,
                                                      ws=outer._scratch)
                        if outer.grad_accumulator_reduce is not None:
                            acc = outer.grad_accumulator_reduce(acc, num_owned_tiles)
                    # RAS:707-772 + 1102-1125  per-point pass, band clearing and factors fused
                    out = hip_ops.point_backward(
                        xyz, features, obj, intrinsics, q_cp, t_cp, t_pc, ids, acc, attrs, ctx.color_max_sh_band,
                        cfg.grad_q_factor, cfg.grad_s_factor, cfg.grad_alpha_factor, cfg.grad_color_factor,
                        cfg.grad_high_order_color_factor, want_visible=hook is not None,
                        visible_mask=visible_mask, num_owned_tiles=num_owned_tiles,
                        want_visible_features=hook is not None and outer.hook_feature_gradients,
                        want_hook_fields=hook is not None, slots=slots, width=width, height=height)
                    grad_pointcloud, grad_pointcloud_features, gx_vis, gf_vis = out[:4]
                    if hook is not None:  # RAS:1127-1142; the column fields come compact out of the same kernel
                        hook(GaussianPointCloudRasterisation.BackwardValidPointHookInput(
                            point_id_in_camera_list=ids,
                            grad_point_in_camera=gx_vis,
                            grad_pointfeatures_in_camera=gf_vis,
                            magnitude_grad_viewspace_on_image=magnitude_image,
                            num_overlap_tiles=num_overlap_tiles,
                            **out[4]))
                return grad_pointcloud, grad_pointcloud_features, None, None, None, None, None, None, None

        self._module_function = _module_function
        # multi-GPU hooks installed by distributed.shard_rasteriser_across_tile_rows (None on 1 GPU)
        # the hook's M-compact copy of the feature gradients (RAS:1132) costs a 224 B x M write per backward; a consumer
        # that does not read it on every iteration (the trainer: only when the controller densifies) may switch it
        # off, ``grad_pointfeatures_in_camera`` is then None
        self.hook_feature_gradients: bool = True
        self.grad_accumulator_reduce: Optional[Callable[[torch.Tensor, torch.Tensor], torch.Tensor]] = None
        self.image_gather: Optional[Callable[[list], None]] = None

    def list_layout(self, height: Optional[int] = None, width: Optional[int] = None) -> "hip_ops.ListLayout":
        """The list layout the next forward pass will use (for an image of ``height`` pixels when sharded; with
        ``width`` too, the automatic bin size is the one learnt for that image size, else the last frame's)."""
        begin, step, end = self.tile_row_begin, self.tile_row_step, self.tile_row_end
        if self.shard is not None:
            from .distributed import owned_tile_rows
            if height is None:
                raise ValueError("a sharded layout depends on the image height")
            rank, world, mode = self.shard
            rows = owned_tile_rows(height // TILE_HEIGHT, rank, world, mode, self.shard_row_weights)
            begin, step, end = rows.start, rows.step, rows.stop
        auto = self._auto_bin_shift if width is None else self._auto_bin_shift_by_size.get((width, height),
                                                                                              self._auto_bin_shift)
        return hip_ops.ListLayout(bin_shift=auto if self.bin_shift is None else self.bin_shift,
                                  exact_cull=self.exact_tile_cull, row_begin=begin, row_step=step, row_end=end)

    def _sizes_arrived(self, host, layout, width, height, guess, guess_key) -> bool:
        """The frame's sizes have reached the host: picks the next frame's list layout and capacities; -> whether the
        speculative launches of this frame (``guess`` = (key capacity, depth bound) or None) held."""
        outer = self
        m, n_keys, n_slots = (host[hip_ops.COUNTER_NUM_VISIBLE], host[hip_ops.COUNTER_NUM_KEYS],
                              host[hip_ops.COUNTER_NUM_SLOTS])
        max_depth_key = host[hip_ops.COUNTER_MAX_DEPTH_KEY]
        if n_keys >= 0x7fffffff or n_slots >= 0x7fffffff:
            raise RuntimeError("more than 2^31-1 (tile, Gaussian) pairs: key offsets are int32 as in the reference")
        # next frame's list layout, from this frame's key count K (after the exact cull, in the layout this frame
        # used; scaled to the whole image when sharded) and M, with hysteresis:
        #   per-tile keys -> 2x2-tile bins once K >= 2e6: key generation + three radix passes then cost more than
        #     the blend kernels pay for filtering twice as many list entries (-4 % at the headline size); back
        #     below 0.7e6 bin keys (small frames: the filter costs more than the sort saves);
        #   -> 4x4-tile bins once a Gaussian emits >= 64 tile keys / >= 16 bin keys on average (lists dominated
        #     by pairs that are never blended: the reference's stress distribution, 8.3 -> 0.64 ms); back below 3.
        # Counting emitted keys rather than tile-box areas keeps needle-shaped Gaussians (huge boxes that the cull
        # empties) from pushing an ordinary frame into the coarse bins.
        # Sharded runs: every rank decides from its OWN key count (scaled to the whole image), so ranks may pick
        # different layouts for the same frame.  That is correct by construction -- image, depth and count are
        # bit-identical across layouts (tests: test_list_layouts_are_output_identical, the sharded x binned tests), and
        # the gradients every rank ends up with after the accumulator exchange are identical ACROSS RANKS.  They are not
        # bit-identical to an un-sharded run or between layouts on small grids: a rank's band usually has <= 3,840
        # tiles, where per-tile / walked lists select the four-waves-per-tile backward, whose slot sums add the same
        # per-pixel terms in another order.  Costs at most some load imbalance; set `bin_shift` to pin one layout.
        owned = len(layout.owned_rows(height))
        k_frame = n_keys * ((height // TILE_HEIGHT) / owned if owned else 1.0)
        used = layout.bin_shift
        # (the replicated cloud's M is the whole frame's: it goes with the key count scaled to the whole frame)
        choice = hip_ops.next_bin_shift(used, k_frame, k_frame, m)
        outer._auto_bin_shift = choice
        if len(outer._auto_bin_shift_by_size) >= 64 and (width, height) not in outer._auto_bin_shift_by_size:
            outer._auto_bin_shift_by_size.pop(next(iter(outer._auto_bin_shift_by_size)))   # bounded
        outer._auto_bin_shift_by_size[(width, height)] = outer._auto_bin_shift   # cameras of several sizes
        fits = guess is not None and n_keys <= guess[0] and max_depth_key <= guess[1]
        outer.speculation_stats["frames"] += 1
        outer.speculation_stats["redone"] += 0 if (fits or guess is None) else 1
        # capacities for the next frame: 30 % head-room over this frame, decaying slowly (1 % per frame) from the
        # high-water mark -- training alternates between views whose key counts differ; the depth range as the
        # largest value with the same number of bits, never less than the previous bound while it still fits
        depth_bound = (1 << max(int(max_depth_key), 1).bit_length()) - 1
        if guess is not None and depth_bound <= guess[1] <= 4 * depth_bound + 3:
            depth_bound = guess[1]
        if len(outer._size_guesses) >= 16 and guess_key not in outer._size_guesses:
            outer._size_guesses.pop(next(iter(outer._size_guesses)))   # bounded: forget the oldest configuration
        outer._size_guesses[guess_key] = (max(int(1.3 * n_keys) + 4096, int(0.99 * guess[0]) if guess else 0),
                                          depth_bound)
        return fits

    def release_workspaces(self) -> None:
        """Gives back the scratch buffers this operator keeps between frames (sort ping-pong, slot records, ...: they only
        grow, to the high-water mark of training); they are re-allocated on demand.  Call it around evaluation or after a
        densification shrank the frame -- not while a frame is in flight on another stream (the buffers assume the one
        stream the operator is called on)."""
        self._scratch.release()

    def _counter_readback(self, device):
        rb = self._readbacks.get(device)
        if rb is None:
            rb = self._readbacks[device] = hip_ops.CounterReadback(device)
            rb.event.record(torch.cuda.current_stream(device))   # creates the HIP event: gs_frame_forward records it by handle
        return rb

    def forward(self, input_data: "GaussianPointCloudRasterisation.GaussianPointCloudRasterisationInput"):
        camera_info = input_data.camera_info
        assert camera_info.camera_width % TILE_WIDTH == 0    # RAS:1193
        assert camera_info.camera_height % TILE_HEIGHT == 0  # RAS:1194
        # will anything be back-propagated through this call?  (Function.forward itself always runs in no-grad mode, so
        # the question is answered here.)  If not, the forward skips the state only the backward pass reads.
        need_state = torch.is_grad_enabled() and (input_data.point_cloud.requires_grad or
                                                  input_data.point_cloud_features.requires_grad)
        return self._module_function.apply(
            input_data.point_cloud,
            input_data.point_cloud_features,
            input_data.point_invalid_mask,
            input_data.point_object_id,
            input_data.q_pointcloud_camera,
            input_data.t_pointcloud_camera,
            camera_info,
            input_data.color_max_sh_band,
            need_state,
        )
