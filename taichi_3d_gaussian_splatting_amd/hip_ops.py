"""Stage-level Python wrappers over the C ABI (include/gsplat_hip.h).

Each function allocates its outputs with torch (the caller owns every buffer, the library
allocates nothing), passes raw device pointers + the current HIP stream, and returns torch
tensors.  One function per reference stage; the stage it replaces is cited in the C header.
PyTorch is plumbing here (device memory + streams); all arithmetic runs in the HIP kernels.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import call, current_stream, ptr

TILE_WIDTH = 16
TILE_HEIGHT = 16
ATTR_STRIDE = 16
ACC_STRIDE = 12
FEATURE_DIM = 56
COUNTER_NUM_VISIBLE = 0
COUNTER_NUM_KEYS = 1
COUNTER_NUM_SLOTS = 2
COUNTER_MAX_DEPTH_KEY = 3
NUM_COUNTERS = 8
_PRE_BLOCK = 256  # points per workgroup of gs_preprocess / gs_make_keys
FILTER_BOX = 1     # include/gsplat_hip.h GS_FILTER_BOX
FILTER_CULL = 2    # GS_FILTER_CULL
_NO_ROW_LIMIT = 1 << 30


@dataclass(frozen=True)
class ListLayout:
    """How the sorted lists of one frame are organised (the same object goes to preprocess, make_keys, tile_ranges and
    both blend passes -- they must agree):
      bin_shift   : sort keys are emitted per bin of (1 << bin_shift)^2 tiles.  0 (default) = one key per tile as in
                    the reference (RAS:131-172): the blend kernels stage their tile's list as it is.  > 0 (2 = 64 x 64
                    pixels): 3-16x fewer keys to generate and sort; the blend kernels then walk their bin's list and
                    keep, in order, the entries of their own tile (box + exact contribution test) -- pays when a
                    Gaussian covers many tiles (the reference's stress test: 5760 tiles per Gaussian), costs 10-25 %
                    on scenes of small Gaussians where the sort is a small part of the frame
      exact_cull  : drop (bin | tile, Gaussian) pairs that cannot reach alpha >= 1/255 (output-identical)
      row_begin / row_step / row_end : the tile rows {row_begin + k*row_step} < row_end this GPU renders."""
    bin_shift: int = 0
    exact_cull: bool = True
    row_begin: int = 0
    row_step: int = 1
    row_end: int = _NO_ROW_LIMIT

    @property
    def filter(self) -> int:
        """Staging filter of the blend kernels: none for per-tile lists (the key generator emitted exactly the tile's
        entries), tile box (+ exact contribution test) for bin lists."""
        if self.bin_shift == 0:
            return 0
        return FILTER_BOX | (FILTER_CULL if self.exact_cull else 0)

    @property
    def sharded(self) -> bool:
        return self.row_begin != 0 or self.row_step != 1 or self.row_end < _NO_ROW_LIMIT

    def num_bins(self, width: int, height: int) -> int:
        side = TILE_WIDTH << self.bin_shift
        return ((width + side - 1) // side) * ((height + side - 1) // side)

    def owned_rows(self, height: int) -> range:
        return range(self.row_begin, min(height // TILE_HEIGHT, self.row_end), self.row_step)


PER_TILE_LISTS = ListLayout(bin_shift=0, exact_cull=False)   # the reference's lists


class Workspaces:
    """Scratch buffers of one operator instance that are dead when the call that uses them returns (sort ping-pong keys,
    histograms, block sums, slot records, ...): kept between frames instead of ~10 ``torch.empty`` calls per frame (host
    time that small frames cannot hide behind the GPU).  Buffers only grow.  Nothing that is returned to the caller or
    saved for the backward pass lives here, so frames may interleave (two forwards, then two backwards) on one stream."""

    def __init__(self):
        self._bufs = {}

    def release(self) -> None:
        """Drops every buffer (they come back on demand): for callers that want the high-water marks of training back
        around evaluation or after a densification that shrank the frame.  Only call it when no launch that uses the
        buffers is pending on another stream: the buffers assume ONE stream (calls on the operator's stream are ordered
        by it; a buffer handed to a second stream while the first still reads it would race)."""
        self._bufs.clear()

    def get(self, name: str, shape, dtype: torch.dtype, device, zeroed: bool = False) -> torch.Tensor:
        """zeroed: the buffer is zero-filled when it is (re)allocated -- for workspaces whose users leave them zero."""
        numel = 1
        for d in (shape if isinstance(shape, (tuple, list)) else (shape,)):
            numel *= int(d)
        t = self._bufs.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype or t.device != device:
            grow = numel if t is None else max(numel, (t.numel() * 5) // 4)
            t = (torch.zeros if zeroed else torch.empty)(max(grow, 1), dtype=dtype, device=device)
            self._bufs[name] = t
        return t[:numel].view(shape)


WALKED_LIST_MAX_BYTES = 512 << 20   # above this a binned frame keeps the staged (re-filtering) backward


def next_bin_shift(used: int, keys_whole_image: float, keys: float, gaussians: int) -> int:
    """The automatic list layout of the NEXT frame from this frame's key count, with hysteresis (one rule for the operator
    and for a band of the owner mode).  `keys_whole_image`: the emitted (bin, Gaussian) keys in the layout `used`, scaled to
    the whole image when only some tile rows were rendered -- the absolute thresholds: per-tile keys -> 2 x 2-tile bins from
    2e6 keys on (key generation + radix passes then cost more than the blend kernels pay for filtering), back below 0.7e6
    bin keys.  `keys` / `gaussians`: keys and the Gaussians that emitted them, BOTH of the same region (the frame, or the
    band) -- the keys-per-Gaussian thresholds: 4 x 4-tile bins once a Gaussian emits >= 64 tile keys / >= 16 bin keys on
    average (lists dominated by pairs that are never blended: the reference's stress distribution), back below 3."""
    if gaussians <= 0:
        return used       # nothing on screen: no information
    if used == 0:
        return 2 if keys >= 64 * gaussians else (1 if keys_whole_image >= 2_000_000 else 0)
    if used == 1:
        return 2 if keys >= 16 * gaussians else (0 if keys_whole_image < 700_000 else 1)
    return 1 if keys < 3 * gaussians else used


def can_emit_walked_lists(n_keys: int, bin_shift: int) -> bool:
    """Whether a binned forward pass may write out its per-tile lists for the backward pass: the buffer holds
    (n_keys << 2 bin_shift) int32 -- the key capacity amplified by the tiles per bin -- and lives until the backward pass;
    it must index with int32 and stay within WALKED_LIST_MAX_BYTES (a dense frame at 4 x 4-tile bins would otherwise pin
    gigabytes per pending frame where the staged backward needs none)."""
    entries = max(int(n_keys), 1) << (2 * int(bin_shift))
    return 0 < bin_shift <= 2 and entries < 2 ** 31 and 4 * entries <= WALKED_LIST_MAX_BYTES


def _scratch(ws: Optional["Workspaces"], name: str, shape, dtype, device) -> torch.Tensor:
    return torch.empty(shape, dtype=dtype, device=device) if ws is None else ws.get(name, shape, dtype, device)


def _require_device(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on {t.device}: the rasteriser runs on an AMD GPU (HIP) only; there is no CPU path")


def _f32(t: torch.Tensor, name: str, contiguous: bool = True) -> torch.Tensor:
    _require_device(t, name)
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous() if contiguous else t


def read_counters(counters: torch.Tensor) -> Tuple[int, ...]:
    host = (ctypes.c_int32 * NUM_COUNTERS)()
    call("gs_read_counters", ptr(counters), ctypes.cast(host, ctypes.c_void_p), NUM_COUNTERS,
         current_stream(counters.device))
    return tuple(host)


class CounterReadback:
    """The frame's sizes (M, K, slot count, depth range) on their way to the host WITHOUT stalling the launch queue.

    Two transports into the same 4 * NUM_COUNTERS = 32 bytes of pinned host memory (read as int32[8] or as uint64[4]):
    * ``start`` / ``wait()``: the counters are copied (or stored by the scan kernel) as int32[NUM_COUNTERS] and an event
      recorded behind them tells the host when;
    * ``next_stamp`` / ``wait(stamp)`` (gs_frame_forward with GsFrame.size_stamp): each size arrives as one 64-bit word
      {stamp, value} written through to host memory by the scan kernel, and the host reads the words until they carry the
      frame's stamp -- no event, whose signal packet holds the next kernel back by ~6 us per frame.  Needs memory that is
      coherent while kernels run (gs_host_alloc_coherent); GS_SIZE_STAMPS=0 or a failed allocation leave the event form."""

    def __init__(self, device):
        self.device = device
        self.event = torch.cuda.Event()
        self._raw, self._stamp = None, 0
        # (GS_HOST_MIRROR=0 -- the development knob of gs_frame_forward that sends the sizes through a copy launch -- excludes
        # the stamped words, which the scan kernel itself stores: the event form is used then)
        if os.environ.get("GS_SIZE_STAMPS", "1") != "0" and os.environ.get("GS_HOST_MIRROR", "1") != "0":
            p = ctypes.c_void_p()
            if _lib.load().gs_host_alloc_coherent(4 * NUM_COUNTERS, ctypes.byref(p)) == 0 and p.value:
                self._raw = p.value
                self._buf = (ctypes.c_int32 * NUM_COUNTERS).from_address(p.value)
                self.host = torch.frombuffer(self._buf, dtype=torch.int32)
        if self._raw is None:
            self.host = torch.empty(NUM_COUNTERS, dtype=torch.int32).pin_memory()

    def __del__(self):
        raw, self._raw = getattr(self, "_raw", None), None
        if raw is not None:
            try:
                _lib.load().gs_host_free(ctypes.c_void_p(raw))
            except Exception:   # interpreter shutdown
                pass

    def start(self, counters: torch.Tensor) -> None:
        call("gs_read_counters_async", ptr(counters), self.host.data_ptr(), NUM_COUNTERS, current_stream(self.device))
        self.event.record(torch.cuda.current_stream(self.device))

    def next_stamp(self) -> int:
        """A stamp for the next frame's sizes (0: stamps are not available, use the event)."""
        if self._raw is None:
            return 0
        self._stamp = self._stamp % 0x7FFFFFFF + 1
        return self._stamp

    def wait(self, stamp: int = 0) -> Tuple[int, ...]:
        if not stamp:
            self.event.synchronize()
            return tuple(self.host.tolist())
        sizes = (ctypes.c_int32 * 4)()
        lib = _lib.load()
        rc = lib.gs_wait_stamped_sizes(ctypes.c_void_p(self._raw), stamp, 2_000_000, sizes)
        if rc == 1:   # not there after 2 s: let the stream finish (a fault surfaces here), then they must be
            torch.cuda.current_stream(self.device).synchronize()
            rc = lib.gs_wait_stamped_sizes(ctypes.c_void_p(self._raw), stamp, 100_000, sizes)
        if rc != 0:
            raise RuntimeError("the frame's sizes never reached the host (stamped words; GS_SIZE_STAMPS=0 selects the event form)")
        out = [0] * NUM_COUNTERS
        out[COUNTER_NUM_VISIBLE], out[COUNTER_NUM_KEYS] = sizes[0], sizes[1]
        out[COUNTER_NUM_SLOTS], out[COUNTER_MAX_DEPTH_KEY] = sizes[2], sizes[3]
        return tuple(out)


def scan_block_sums_async(block_sums: torch.Tensor, counters: torch.Tensor, block_sums_full: torch.Tensor) -> None:
    """The two in-place scans of ``scan_block_sums`` without its size read-back."""
    call("gs_scan_block_sums2", ptr(block_sums), ptr(block_sums_full), block_sums.shape[0], ptr(counters),
         current_stream(counters.device))


def pose_inverse(q_pointcloud_camera: torch.Tensor, t_pointcloud_camera: torch.Tensor):
    q = _f32(q_pointcloud_camera, "q_pointcloud_camera").reshape(-1, 4)
    t = _f32(t_pointcloud_camera, "t_pointcloud_camera").reshape(-1, 3)
    if q.shape[0] != t.shape[0] or q.shape[0] == 0:
        raise ValueError("q_pointcloud_camera / t_pointcloud_camera must be (K,4)/(K,3) with K >= 1")
    q_inv, t_inv = torch.empty_like(q), torch.empty_like(t)
    call("gs_pose_inverse", ptr(q), ptr(t), ptr(q_inv), ptr(t_inv), q.shape[0], current_stream(q.device))
    return q_inv, t_inv


def filter_compact(xyz, invalid_mask, object_id, intrinsics, q_cp, t_cp, near_plane, far_plane, width, height,
                   counters: Optional[torch.Tensor] = None, sync: bool = True, ws: Optional[Workspaces] = None):
    """-> (mask int8[N], ids, counters).  sync=True: blocks on the size read-back (RAS:870) and returns
    ids int32[M]; sync=False: returns the int32[N] buffer whose first M entries are valid, M staying on the
    device in counters[COUNTER_NUM_VISIBLE] (pass it on with preprocess(..., n_visible_on_device=True))."""
    xyz = _f32(xyz, "point_cloud")
    n = xyz.shape[0]
    dev = xyz.device
    mask = torch.empty(n, dtype=torch.int8, device=dev)
    ids = torch.empty(n, dtype=torch.int32, device=dev)
    if counters is None:
        counters = torch.empty(NUM_COUNTERS, dtype=torch.int32, device=dev)   # zeroed by gs_filter_compact
    scratch = _scratch(ws, "filter", _lib.load().gs_filter_workspace_bytes(n), torch.uint8, dev)
    call("gs_filter_compact", ptr(xyz), ptr(invalid_mask), ptr(object_id), ptr(intrinsics), ptr(q_cp), ptr(t_cp),
         n, float(near_plane), float(far_plane), int(width), int(height), ptr(mask), ptr(ids), ptr(counters),
         ptr(scratch), current_stream(dev))
    if not sync:
        return mask, ids, counters
    m = read_counters(counters)[COUNTER_NUM_VISIBLE]
    return mask, ids[:m], counters


def preprocess(xyz, features, object_id, intrinsics, q_cp, t_cp, ids, width, height, layout: ListLayout = ListLayout(),
               depth_to_sort_key_scale=100.0, counters=None, n_visible_on_device=False,
               always_store_rotation: bool = False, ws: Optional[Workspaces] = None, colours: bool = True):
    """-> (attrs f32[M,16], num_overlap_tiles i32[M], num_keys i32[M], block_sums, block_sums_full).
    colours=False: gs_preprocess_geometry -- floats 8..10 of the records are left to ``view_colours``.
    Normalises features[ids, 0:4] IN PLACE (RAS:196-205).  num_overlap_tiles is the reference's box count
    (hook output; its scan gives the backward slots); num_keys is the number of sort keys emitted (bins reached in
    owned tile rows, after the exact cull); the two block_sums are int32[ceil(M/256)] partial sums of them."""
    m = ids.shape[0]
    dev = xyz.device
    attrs = torch.empty((m, ATTR_STRIDE), dtype=torch.float32, device=dev)
    ntiles = torch.empty(m, dtype=torch.int32, device=dev)
    nkeys = torch.empty(m, dtype=torch.int32, device=dev)
    nblk = (m + _PRE_BLOCK - 1) // _PRE_BLOCK
    block_sums = _scratch(ws, "block_sums", nblk, torch.int32, dev)
    block_sums_full = _scratch(ws, "block_sums_full", nblk, torch.int32, dev)
    call("gs_preprocess" if colours else "gs_preprocess_geometry", ptr(xyz), ptr(features), ptr(object_id),
         ptr(intrinsics), ptr(q_cp), ptr(t_cp), ptr(ids), m, int(bool(n_visible_on_device)), int(width), int(height), layout.row_begin, layout.row_step, layout.row_end,
         layout.bin_shift, int(layout.exact_cull), int(bool(always_store_rotation)), float(depth_to_sort_key_scale),
         ptr(counters), ptr(attrs),
         ptr(ntiles), ptr(nkeys), ptr(block_sums), ptr(block_sums_full), current_stream(dev))
    return attrs, ntiles, nkeys, block_sums, block_sums_full


def view_colours(xyz, features, object_id, q_cp, t_cp, ids, num_keys, attrs, counters=None, n_visible_on_device=False,
                 stream: Optional[int] = None) -> None:
    """The colour half of ``preprocess`` (RAS:280-282,302-310): writes floats 8..10 of the records of the Gaussians with
    num_keys > 0, in place.  stream: raw handle of the stream to launch on (default: torch's current stream)."""
    dev = xyz.device
    call("gs_view_colours", ptr(xyz), ptr(features), ptr(object_id), ptr(q_cp), ptr(t_cp), ptr(ids), ids.shape[0],
         int(bool(n_visible_on_device)), ptr(counters), ptr(num_keys), ptr(attrs),
         current_stream(dev) if stream is None else stream)


def scan_block_sums(block_sums: torch.Tensor, counters: torch.Tensor,
                    block_sums_full: Optional[torch.Tensor] = None):
    """In-place exclusive scans; returns K (and the number of backward slots when block_sums_full is given).
    Blocks on ONE size read-back (RAS:916)."""
    stream = current_stream(counters.device)
    if block_sums_full is not None:
        call("gs_scan_block_sums2", ptr(block_sums), ptr(block_sums_full), block_sums.shape[0], ptr(counters), stream)
    else:
        call("gs_scan_block_sums", ptr(block_sums), block_sums.shape[0], ptr(counters), COUNTER_NUM_KEYS, stream)
    host = read_counters(counters)
    k, n_slots = host[COUNTER_NUM_KEYS], host[COUNTER_NUM_SLOTS]
    if k >= 0x7fffffff or n_slots >= 0x7fffffff:
        raise RuntimeError("more than 2^31-1 (tile, Gaussian) pairs: key offsets are int32 as in the reference")
    if block_sums_full is None:
        return k
    return k, n_slots, host[COUNTER_MAX_DEPTH_KEY], host[COUNTER_NUM_VISIBLE]


def make_keys(attrs, num_keys, block_offsets, n_keys, width, height, depth_to_sort_key_scale,
              layout: ListLayout = ListLayout(), key_depth_bits=0, num_overlap_tiles=None, block_offsets_full=None,
              counters=None, ws: Optional[Workspaces] = None):
    """-> (keys, payload, slot_offsets).  key_depth_bits == 0: int64 keys in the reference layout
    (bin << 32) + depth; key_depth_bits > 0: 32-bit keys (bin << key_depth_bits) | depth, stored in an int32
    tensor.  slot_offsets i32[M] = exclusive scan of num_overlap_tiles (base of every Gaussian's backward
    slots) when num_overlap_tiles + its scanned block sums are given, else None (inference).
    counters given: the sizes are still on the device -- attrs/num_keys are capacity-sized, n_keys is the CAPACITY of
    the key arrays (see include/gsplat_hip.h)."""
    dev = attrs.device
    m = attrs.shape[0]
    keys = _scratch(ws, "keys", n_keys, torch.int64 if key_depth_bits == 0 else torch.int32, dev)
    payload = torch.empty(n_keys, dtype=torch.int32, device=dev)
    slot_offsets = torch.empty(m, dtype=torch.int32, device=dev) if num_overlap_tiles is not None else None
    if m > 0:
        call("gs_make_keys", ptr(attrs), ptr(num_keys), ptr(block_offsets), m, ptr(counters), int(n_keys),
             int(width), int(height), layout.row_begin, layout.row_step, layout.row_end, layout.bin_shift, int(layout.exact_cull),
             int(key_depth_bits), float(depth_to_sort_key_scale), ptr(keys), ptr(payload), ptr(num_overlap_tiles),
             ptr(block_offsets_full), ptr(slot_offsets), current_stream(dev))
    return keys, payload, slot_offsets


def sort_key_bits(near_plane: float, far_plane: float, depth_to_sort_key_scale: float, num_tiles: int):
    """Bit ranges that can differ between keys: quantised depth [0,depth_bits), tile [32,32+tile_bits)."""
    tile_bits = max(int(num_tiles) - 1, 0).bit_length()
    max_dq = far_plane * depth_to_sort_key_scale
    if near_plane >= 0 and depth_to_sort_key_scale >= 0 and 0 <= max_dq < 2 ** 31 - 1:
        return max(int(max_dq), 1).bit_length(), tile_bits
    return 64, tile_bits  # negative depths borrow from the tile field: sort the whole signed key


def key_layout(near_plane: float, far_plane: float, depth_to_sort_key_scale: float, num_tiles: int,
               max_depth_key: Optional[int] = None):
    """-> (key_depth_bits, depth_bits, tile_bits).  key_depth_bits > 0 selects the compressed 32-bit key
    (possible when the quantised depth is provably in [0, 2^depth_bits) and tile+depth fit 32 bits).
    max_depth_key: the largest quantised depth on screen as measured by gs_preprocess; when given, the depth
    field is sized to the bits actually in use (fewer radix passes than the far_plane*scale bound)."""
    depth_bits, tile_bits = sort_key_bits(near_plane, far_plane, depth_to_sort_key_scale, num_tiles)
    if max_depth_key is not None and depth_bits < 64 and max_depth_key >= 0:
        depth_bits = min(depth_bits, max(int(max_depth_key), 1).bit_length())
    if depth_bits < 64 and depth_bits + tile_bits <= 32:
        return depth_bits, depth_bits, tile_bits
    return 0, depth_bits, tile_bits


def sort_pairs(keys: torch.Tensor, payload: torch.Tensor, depth_bits: int, tile_bits: int,
               key_depth_bits: int = 0, in_place: bool = True, n_keys_device: Optional[torch.Tensor] = None,
               ws: Optional[Workspaces] = None, bins_in_any_order: bool = False, ranges: Optional[torch.Tensor] = None):
    """Stable sort of (keys, payload).  in_place=True: the inputs hold the result.  in_place=False: returns the
    (keys, payload) tensors that hold the result (the inputs or the ping-pong buffers: no copy back after an
    odd number of passes); the other pair is scratch.  bins_in_any_order: what a frame needs -- every bin's pairs
    contiguous and stably sorted by depth, the bins themselves in the sort's own order (include/gsplat_hip.h).
    ranges: int32[2, number of bins] -- zero-filled here and, when the sort can (MSD-first path, buckets of whole bins),
    filled with the bins' [start, end) ranges by the sort itself; with `ranges` the return value is
    (keys, payload, ranges_were_written) and the caller runs ``tile_ranges`` only if they were not."""
    n = keys.shape[0]   # capacity when n_keys_device (an int32 device scalar holding the actual count) is given
    if ranges is not None and (in_place or ranges.dtype != torch.int32 or ranges.dim() != 2 or ranges.shape[0] != 2):
        raise ValueError("ranges: int32[2, bins], with in_place=False")
    if n <= 1:
        if ranges is not None:
            return keys, payload, False
        return None if in_place else (keys, payload)
    if keys.dtype != (torch.int64 if key_depth_bits == 0 else torch.int32):
        raise TypeError("key dtype does not match the key layout")
    dev = keys.device
    if ranges is not None:
        ranges.zero_()
    keys_alt = _scratch(ws, "keys_alt", n, keys.dtype, dev)
    payload_alt = torch.empty_like(payload)   # (either payload buffer may end up holding the result: not scratch)
    scratch = _scratch(ws, "sort", _lib.load().gs_sort_workspace_bytes(n), torch.uint8, dev)
    status = _lib.load().gs_sort_pairs_and_zero(ptr(keys), ptr(payload), ptr(keys_alt), ptr(payload_alt), n,
                                                ptr(n_keys_device), int(key_depth_bits), int(depth_bits), int(tile_bits),
                                                0 if in_place else 1, int(bool(bins_in_any_order)), ptr(scratch), None, 0,
                                                None if ranges is None else ranges[0].data_ptr(),
                                                None if ranges is None else ranges[1].data_ptr(),
                                                0 if ranges is None else ranges.shape[1], current_stream(dev))
    if status < 0:
        _lib.check(status, "gs_sort_pairs")
    if not in_place:
        out = (keys_alt, payload_alt) if status & 1 else (keys, payload)
        return out + (bool(status & 2),) if ranges is not None else out
    return None


def tile_ranges(keys_sorted: torch.Tensor, num_tiles: int, key_depth_bits: int = 0,
                n_keys_device: Optional[torch.Tensor] = None):
    """Per-bin [start, end) ranges of the sorted keys (num_tiles = number of bins; tiles with bin_shift 0)."""
    if keys_sorted.dtype != (torch.int64 if key_depth_bits == 0 else torch.int32):
        raise TypeError("key dtype does not match the key layout")
    dev = keys_sorted.device
    both = torch.empty((2, num_tiles), dtype=torch.int32, device=dev)  # one buffer -> one fill in the library
    start, end = both[0], both[1]
    call("gs_tile_ranges", ptr(keys_sorted), keys_sorted.shape[0], ptr(n_keys_device), int(key_depth_bits), ptr(start), ptr(end),
         int(num_tiles), current_stream(dev))
    return start, end


BLEND_RGB_ONLY = 1      # include/gsplat_hip.h GS_BLEND_RGB_ONLY: no depth / per-pixel count (RAS:464-469,478-484)
BLEND_NO_STATE = 2      # GS_BLEND_NO_STATE: no acc_alpha / last_effective (nothing will be back-propagated)
BLEND_ARMS = {None: 0, "two_waves": 4, "four_waves": 8, "one_wave": 16,   # GS_BLEND_TWO_WAVES / _FOUR_WAVES / _ONE_WAVE (None: by tile count)
              "two_waves_skewed": 4 | 64}                                 # ... | GS_BLEND_SKEWED_WALKS (backward: sums in registers)


SPLIT_GRID_TILES = 1024          # csrc/gs_blend.hip backward_split_for: grids up to this many tiles get a split backward
FORWARD_SPLIT_GRID_TILES = 320   # include/gsplat_hip.h GS_FORWARD_SPLIT_TILES: ... a split forward
MAX_BOUNDARY_BYTES = 256 << 20   # above this the backward pass is not split (boundary states cost 4 KB per 128 list entries)


def boundary_states_bytes(list_length: int, width: int, height: int, layout: ListLayout, walked: bool) -> int:
    """Bytes of the buffer in which the forward pass leaves its boundary states for a SPLIT backward pass
    (include/gsplat_hip.h "List splitting"), or 0 when the backward pass of this frame will not be split: both passes
    walk the layout's own per-tile lists (bin_shift 0, no filter) on a grid of at most 1024 tiles.  list_length = the
    payload's length (capacity), the same value the two blend calls are given."""
    if walked or layout.bin_shift != 0 or layout.filter != 0:
        return 0
    if num_owned_tiles(width, height, layout) > SPLIT_GRID_TILES:
        return 0
    nbytes = int(_lib.load().gs_blend_boundary_bytes(int(list_length), int(width), int(height)))
    return nbytes if nbytes <= MAX_BOUNDARY_BYTES else 0


def split_workspace(ws: Workspaces, width: int, height: int, device) -> torch.Tensor:
    """The split backward's tile counters + per-segment images: zero when first used, left zero by the library."""
    return ws.get("split_backward", int(_lib.load().gs_blend_split_workspace_bytes(int(width), int(height))), torch.uint8,
                  device, zeroed=True)


BLEND_SPLIT_FORWARD = 32   # GS_BLEND_SPLIT_FORWARD
BLEND_SKEWED_WALKS = 64    # GS_BLEND_SKEWED_WALKS


def forward_split_bytes(width: int, height: int, layout: ListLayout, force: bool = False) -> int:
    """Bytes of scratch with which gs_blend_forward_split gives a tile several workgroups (include/gsplat_hip.h "List
    splitting": per-tile lists taken as they are on a grid of at most 320 rendered tiles), or 0: the forward is not split."""
    if layout.bin_shift != 0 or layout.filter != 0:
        return 0
    if not force and num_owned_tiles(width, height, layout) > FORWARD_SPLIT_GRID_TILES:
        return 0
    return int(_lib.load().gs_blend_forward_split_workspace_bytes(int(width), int(height)))


def forward_split_workspace(ws: Optional[Workspaces], width: int, height: int, layout: ListLayout, device,
                            force: bool = False) -> Optional[torch.Tensor]:
    nbytes = forward_split_bytes(width, height, layout, force)
    return _scratch(ws, "split_forward", nbytes, torch.uint8, device) if nbytes else None


def blend_forward(bin_start, bin_end, payload, attrs, width, height, layout: ListLayout = ListLayout(),
                  out=None, rgb_only=False, need_state=True, debug_hits=False, gathered_rows: int = 0,
                  ordered: bool = False, tile_work: Optional[torch.Tensor] = None, arm: Optional[str] = None,
                  ws: Optional[Workspaces] = None, emit_walked_lists: bool = False,
                  boundary: Optional[torch.Tensor] = None, split: bool = False):
    """-> (image, depth, acc_alpha, last_effective, count).  rgb_only: depth and count are not computed (returned
    as None); need_state=False: acc_alpha / last_effective are not computed (None) -- the inference path.
    debug_hits=True appends a uint32-as-int32 [H,W,2] tensor {blended count, hash of blended payloads} per pixel.
    ordered: tiles are dispatched longest list first (same results, shorter tail of the launch); tile_work (int32[owned
    tiles], needs the state): receives the walk lengths the backward pass will see (blend_backward_partials).
    emit_walked_lists (binned layouts with state): appends (walked_start i32[tiles], walked_list i32[K << 2 bin_shift]) --
    every tile's own list as far as it was walked; last_effective then refers to positions in walked_list and the
    backward pass is run on (walked_start, walked_list) with ``walked_layout(layout)``.
    split: small grids with per-tile lists give a tile several workgroups (gs_blend_forward_split): every decision as the
    un-split pass takes it, values equal to rounding; split="force": whatever the grid size (tests, measurements)."""
    dev = bin_start.device
    flags = (BLEND_RGB_ONLY if rgb_only else 0) | (0 if need_state else BLEND_NO_STATE) | BLEND_ARMS[arm]
    if out is None:
        alloc = torch.zeros if layout.sharded else torch.empty  # un-owned tiles are left untouched
        f32 = lambda *shape: alloc(shape, dtype=torch.float32, device=dev)   # noqa: E731
        i32 = lambda *shape: alloc(shape, dtype=torch.int32, device=dev)     # noqa: E731
        if gathered_rows >= height:
            # the caller all-gathers image / depth / count over the ranks of a tile-row sharding: allocated with
            # `gathered_rows` rows (distributed.padded_image_rows) so that the gather runs in place, returned as their
            # first `height` rows, and not zero-filled -- every row is either rendered here or received
            g32 = lambda dt, *shape: torch.empty((gathered_rows,) + shape[1:], dtype=dt, device=dev)[:height]   # noqa: E731
            out = (g32(torch.float32, height, width, 3), None if rgb_only else g32(torch.float32, height, width),
                   torch.empty((height, width), dtype=torch.float32, device=dev) if need_state else None,
                   torch.empty((height, width), dtype=torch.int32, device=dev) if need_state else None,
                   None if rgb_only else g32(torch.int32, height, width))
        else:
            out = (f32(height, width, 3), None if rgb_only else f32(height, width),
                   f32(height, width) if need_state else None, i32(height, width) if need_state else None,
                   None if rgb_only else i32(height, width))
    image, depth, acc_alpha, last_eff, count = out
    dbg = torch.zeros((height, width, 2), dtype=torch.int32, device=dev) if debug_hits else None
    order = _scratch(ws, "order_fwd", num_owned_tiles(width, height, layout), torch.int32, dev) if ordered else None
    walked_list = walked_start = None
    if emit_walked_lists:
        if not (need_state and layout.filter != 0 and (payload.shape[0] << (2 * layout.bin_shift)) < 2 ** 31):
            raise ValueError("walked lists: binned layout with state, and K << 2 bin_shift must fit int32")
        walked_list = torch.empty(max(payload.shape[0], 1) << (2 * layout.bin_shift), dtype=torch.int32, device=dev)
        walked_start = torch.empty((width // TILE_WIDTH) * (height // TILE_HEIGHT), dtype=torch.int32, device=dev)
    # boundary (uint8 buffer of boundary_states_bytes(...), optional): the forward leaves its boundary states there
    split_ws = forward_split_workspace(ws, width, height, layout, dev, force=split == "force") if split else None
    if split == "force" and split_ws is not None:
        flags |= BLEND_SPLIT_FORWARD
    call("gs_blend_forward_split", ptr(bin_start), ptr(bin_end), ptr(payload), ptr(attrs), int(width), int(height),
         layout.row_begin, layout.row_step, layout.row_end, layout.bin_shift, layout.filter, ptr(image), ptr(depth),
         ptr(acc_alpha), ptr(last_eff), ptr(count), flags, ptr(dbg), ptr(order), ptr(tile_work), ptr(walked_list),
         ptr(walked_start), ptr(boundary), int(payload.shape[0]), ptr(split_ws), current_stream(dev))
    if emit_walked_lists:
        out = out + (walked_start, walked_list)
    return out + (dbg,) if debug_hits else out


def walked_layout(layout: ListLayout) -> ListLayout:
    """The layout under which the backward pass walks the per-tile lists a binned forward pass emitted: plain per-tile
    lists, same tile-row ownership."""
    return ListLayout(bin_shift=0, exact_cull=False, row_begin=layout.row_begin, row_step=layout.row_step,
                      row_end=layout.row_end)


def num_owned_tiles(width: int, height: int, layout: ListLayout) -> int:
    return (width // TILE_WIDTH) * len(layout.owned_rows(height))


def blend_backward_partials(bin_start, payload, attrs, grad_image, acc_alpha, last_eff, slot_offsets, n_slots,
                            width, height, layout: ListLayout = ListLayout(), debug_hits=False, tile_order=None,
                            tile_work=None, arm: Optional[str] = None, ws: Optional[Workspaces] = None,
                            image: Optional[torch.Tensor] = None, boundary: Optional[torch.Tensor] = None):
    """Per-pixel backward pass -> (partials f32[S,12], slot_flags u8[S], magnitude image f32[H,W,2]): one partial
    record per (Gaussian, tile) slot, plain stores, no atomics.  debug_hits=True appends the per-pixel
    {count, hash} record of the pairs the backward treated as blended (see blend_forward)."""
    dev = attrs.device
    grad_image = _f32(grad_image, "grad_rasterized_image")
    partials = _scratch(ws, "partials", (max(int(n_slots), 1), ACC_STRIDE), torch.float32, dev)
    # the flag buffer is allocated padded to 16 bytes (gsplat_hip.h: one aligned fill); callers see the n_slots flags
    flags = _scratch(ws, "slot_flags", (max(int(n_slots), 1) + 15) & ~15, torch.uint8, dev)[:max(int(n_slots), 1)]
    alloc = torch.zeros if layout.sharded else torch.empty
    mag = alloc((height, width, 2), dtype=torch.float32, device=dev)
    dbg = torch.zeros((height, width, 2), dtype=torch.int32, device=dev) if debug_hits else None
    if tile_work is not None:   # the forward's walk lengths: the library sorts the tiles by them (longest first)
        tile_order = _scratch(ws, "order_bwd", tile_work.shape[0], torch.int32, dev)
    # image + boundary (the forward's output image and boundary states): the library may give a tile several workgroups
    split_ws = None
    if image is not None and boundary is not None:
        split_ws = split_workspace(ws, width, height, dev) if ws is not None else torch.zeros(
            int(_lib.load().gs_blend_split_workspace_bytes(int(width), int(height))), dtype=torch.uint8, device=dev)
    call("gs_blend_backward_split", ptr(bin_start), ptr(payload), ptr(attrs), ptr(grad_image), ptr(acc_alpha),
         ptr(last_eff), ptr(slot_offsets), int(n_slots), int(width), int(height), layout.row_begin, layout.row_step,
         layout.row_end, layout.bin_shift, layout.filter, ptr(partials), ptr(flags), ptr(mag), ptr(dbg), BLEND_ARMS[arm],
         ptr(tile_work), ptr(tile_order), ptr(image if split_ws is not None else None),
         ptr(boundary if split_ws is not None else None), int(payload.shape[0]), ptr(split_ws), current_stream(dev))
    return (partials, flags, mag, dbg) if debug_hits else (partials, flags, mag)


def reduce_partials(slot_offsets, num_overlap_tiles, flags, partials, num_keys=None, attrs=None, width=0, height=0,
                    ws: Optional[Workspaces] = None):
    """Per-Gaussian sum of its flagged slots, in slot order -> acc f32[M,12].  num_keys (optional): Gaussians with
    no sort key on this GPU are written as zeros without looking at their slots.  attrs + image size (optional): a
    Gaussian with many slots is only looked at where it can have been blended (same sums, fewer flags read)."""
    m = slot_offsets.shape[0]
    acc = _scratch(ws, "acc", (m, ACC_STRIDE), torch.float32, partials.device)
    call("gs_reduce_partials", ptr(slot_offsets), ptr(num_overlap_tiles), ptr(flags), ptr(partials), m, ptr(acc),
         ptr(num_keys), int(partials.shape[0]), ptr(attrs), int(width), int(height), current_stream(partials.device))
    return acc


def blend_backward(bin_start, payload, attrs, grad_image, acc_alpha, last_eff, slot_offsets, num_overlap_tiles,
                   n_slots, width, height, layout: ListLayout = ListLayout(), num_keys=None, tile_work=None):
    """-> (acc f32[M,12], magnitude_grad_viewspace_on_image f32[H,W,2]): blend_backward_partials + reduce_partials."""
    partials, flags, mag = blend_backward_partials(bin_start, payload, attrs, grad_image, acc_alpha, last_eff,
                                                   slot_offsets, n_slots, width, height, layout, tile_work=tile_work)
    return reduce_partials(slot_offsets, num_overlap_tiles, flags, partials, num_keys, attrs, width, height), mag


def compact_rows(acc: torch.Tensor, num_keys: torch.Tensor):
    """Multi-GPU: the accumulator rows this GPU produced (num_keys > 0), ascending -> (ids i32[M], rows f32[M,12],
    count i32[1] on the device); the first `count` entries are valid."""
    m, dev = acc.shape[0], acc.device
    ids = torch.empty(m, dtype=torch.int32, device=dev)
    rows = torch.empty((m, ACC_STRIDE), dtype=torch.float32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    ws = torch.empty(_lib.load().gs_compact_rows_workspace_bytes(m), dtype=torch.uint8, device=dev)
    call("gs_compact_rows", ptr(acc), ptr(num_keys), m, m, ptr(ids), ptr(rows), ptr(count), ptr(ws), current_stream(dev))
    return ids, rows, count


def merge_rows(lists: torch.Tensor, stride_words: int, capacity: int, counts: torch.Tensor, world: int, m: int):
    """Multi-GPU: `world` gathered row lists (int32 words: per rank `capacity` ids, then `capacity` 48-B rows) summed in
    rank order -> dense acc f32[M,12]."""
    acc = torch.empty((m, ACC_STRIDE), dtype=torch.float32, device=lists.device)
    call("gs_merge_rows", ptr(lists), int(stride_words), int(capacity), ptr(counts), int(world), int(m), ptr(acc),
         current_stream(lists.device))
    return acc


def point_backward(xyz, features, object_id, intrinsics, q_cp, t_cp, t_pc, ids, acc, attrs, color_max_sh_band,
                   grad_q_factor, grad_s_factor, grad_alpha_factor, grad_color_factor,
                   grad_high_order_color_factor, want_visible: bool, visible_mask=None, num_owned_tiles=None,
                   want_visible_features: Optional[bool] = None, want_hook_fields: bool = False, slots=None,
                   width: int = 0, height: int = 0):
    """attrs: the packed records of ``preprocess`` (the colour chain reads sigmoid(SH.Y) from them);
    num_owned_tiles (optional): records with 0 owned tiles are incomplete, their colour is re-evaluated on demand.
    acc=None + slots=(slot_offsets, num_overlap_tiles, slot_flags, partials) + image size: the fused form -- the
    accumulators are summed from the slot records of ``blend_backward_partials`` inside the kernel (same bits as
    ``reduce_partials``, no [M,12] array in memory)."""
    dev = xyz.device
    n, m = xyz.shape[0], ids.shape[0]
    grad_xyz = torch.empty((n, 3), dtype=torch.float32, device=dev)
    grad_feat = torch.empty((n, FEATURE_DIM), dtype=torch.float32, device=dev)
    if want_visible_features is None:
        want_visible_features = want_visible
    gx_vis = torch.empty((m, 3), dtype=torch.float32, device=dev) if want_visible else None
    gf_vis = torch.empty((m, FEATURE_DIM), dtype=torch.float32, device=dev) if want_visible_features else None
    hook = torch.empty(7 * m, dtype=torch.float32, device=dev) if want_hook_fields else None
    so, nt, fl, pa = slots if slots is not None else (None, None, None, None)
    if acc is None and slots is None:
        raise ValueError("point_backward needs either acc or the slot records")
    call("gs_point_backward", ptr(xyz), ptr(features), ptr(object_id), ptr(intrinsics), ptr(q_cp), ptr(t_cp),
         ptr(t_pc), ptr(ids), ptr(visible_mask), m, n, ptr(acc), ptr(attrs), ptr(num_owned_tiles), int(color_max_sh_band),
         float(grad_q_factor), float(grad_s_factor),
         float(grad_alpha_factor), float(grad_color_factor), float(grad_high_order_color_factor), ptr(grad_xyz),
         ptr(grad_feat), ptr(gx_vis), ptr(gf_vis), ptr(hook), ptr(so), ptr(nt), ptr(fl), ptr(pa), int(width), int(height),
         current_stream(dev))
    if want_hook_fields:   # views of the planes: viewspace [M,2], magnitude [M], pixels i32[M], depth [M], uv [M,2]
        fields = dict(grad_viewspace=hook[0:2 * m].view(m, 2), magnitude_grad_viewspace=hook[2 * m:3 * m],
                      num_affected_pixels=hook[3 * m:4 * m].view(torch.int32), point_depth=hook[4 * m:5 * m],
                      point_uv_in_camera=hook[5 * m:7 * m].view(m, 2))
        return grad_xyz, grad_feat, gx_vis, gf_vis, fields
    return grad_xyz, grad_feat, gx_vis, gf_vis


# ---------------------------------------------------------------- owner-sharded Gaussians: routed exchange (multi-GPU)
def _band_bounds_array(bounds, world: int):
    """ctypes int32[world + 1] of the bands' tile-row boundaries (None: equal bands, NULL)."""
    if bounds is None:
        return None
    if len(bounds) != world + 1:
        raise ValueError("band boundaries: world + 1 tile rows")
    return (ctypes.c_int32 * (world + 1))(*[int(b) for b in bounds])


def route_count(attrs, num_keys, counters, width, height, rows_per_band: int, world: int, bounds=None):
    """-> (counts i32[world] on the device: records this rank sends to every band, workspace for route_scatter).
    attrs / num_keys are capacity-sized, the visible count is read from ``counters`` on the device.
    bounds: world + 1 tile rows, band b = rows [bounds[b], bounds[b + 1]) (None: equal bands of rows_per_band rows)."""
    dev = attrs.device
    cap_m = attrs.shape[0]
    counts = torch.empty(world, dtype=torch.int32, device=dev)
    ws = torch.empty(_lib.load().gs_route_workspace_bytes(cap_m, world), dtype=torch.uint8, device=dev)
    arr = _band_bounds_array(bounds, world)
    call("gs_route_count", ptr(attrs), ptr(num_keys), cap_m, ptr(counters), int(width), int(height), int(rows_per_band),
         int(world), None if arr is None else ctypes.addressof(arr), ptr(counts), ptr(ws), current_stream(dev))
    return counts, ws


def route_scatter(attrs, num_keys, counters, width, height, rows_per_band: int, world: int, capacity: int, counts,
                  workspace, bounds=None):
    """-> (send f32[world, capacity + 1, 16]: chunk b = header slot + this rank's records for band b in visible-list
    order; pos i32[world, M_capacity]: slot of record i in chunk b or -1).  Must follow ``route_count`` on the same
    inputs (it reads the offsets left in ``workspace``)."""
    dev = attrs.device
    cap_m = attrs.shape[0]
    send = torch.empty((world, capacity + 1, ATTR_STRIDE), dtype=torch.float32, device=dev)
    pos = torch.empty((world, max(cap_m, 1)), dtype=torch.int32, device=dev)
    arr = _band_bounds_array(bounds, world)
    call("gs_route_scatter", ptr(attrs), ptr(num_keys), cap_m, ptr(counters), int(width), int(height), int(rows_per_band),
         int(world), None if arr is None else ctypes.addressof(arr), int(capacity), ptr(counts), ptr(send), ptr(pos),
         ptr(workspace), current_stream(dev))
    return send, pos


def count_keys(records: torch.Tensor, width, height, layout: ListLayout, depth_to_sort_key_scale,
               ws: Optional[Workspaces] = None):
    """Per-record counts of a RECEIVED buffer f32[world, chunk, 16] (see route_scatter), taken as an attrs array of
    world * chunk records -> (counters i32[8], num_overlap_tiles, num_keys, block_sums, block_sums_full)."""
    dev = records.device
    world, chunk = records.shape[0], records.shape[1]
    n = world * chunk
    counters = torch.zeros(NUM_COUNTERS, dtype=torch.int32, device=dev)
    ntiles = torch.empty(n, dtype=torch.int32, device=dev)
    nkeys = torch.empty(n, dtype=torch.int32, device=dev)
    nblk = (n + _PRE_BLOCK - 1) // _PRE_BLOCK
    block_sums = _scratch(ws, "block_sums", nblk, torch.int32, dev)
    block_sums_full = _scratch(ws, "block_sums_full", nblk, torch.int32, dev)
    call("gs_count_keys", ptr(records), n, chunk, int(width), int(height), layout.row_begin, layout.row_step,
         layout.row_end, layout.bin_shift, int(layout.exact_cull), float(depth_to_sort_key_scale), ptr(counters),
         ptr(ntiles), ptr(nkeys), ptr(block_sums), ptr(block_sums_full), current_stream(dev))
    return counters, ntiles, nkeys, block_sums, block_sums_full


def gather_returned_rows(returned: torch.Tensor, pos: torch.Tensor, n_visible: int, capacity: int) -> torch.Tensor:
    """Owner side of the backward exchange: returned f32[world, capacity + 1, 12] (chunk b = the accumulator rows band b
    produced for the records this rank sent it) -> acc f32[n_visible, 12], summed per record in band order."""
    world = returned.shape[0]
    acc = torch.empty((n_visible, ACC_STRIDE), dtype=torch.float32, device=returned.device)
    call("gs_gather_returned_rows", ptr(returned), ptr(pos), int(n_visible), pos.shape[1], world, int(capacity), ptr(acc),
         current_stream(returned.device))
    return acc


# ---------------------------------------------------------------- adaptive-controller kernels (row F2)
def ellipsoid_offsets(features: torch.Tensor) -> torch.Tensor:
    """Focal vector of each Gaussian's ellipsoid, f32[n,3] (ADC:10-25)."""
    features = _f32(features, "point_cloud_features")
    out = torch.empty((features.shape[0], 3), dtype=torch.float32, device=features.device)
    call("gs_ellipsoid_offsets", ptr(features), features.shape[0], ptr(out), current_stream(features.device))
    return out


def sample_from_points(xyz: torch.Tensor, features: torch.Tensor, uniforms: Optional[torch.Tensor] = None,
                       generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """One draw per Gaussian from N(xyz, R S S^T R^T), f32[n,3] (ADC:27-42).  uniforms f32[n,4] in (0,1]
    (drawn with torch when omitted)."""
    xyz, features = _f32(xyz, "point_cloud"), _f32(features, "point_cloud_features")
    n = xyz.shape[0]
    if uniforms is None:
        uniforms = 1.0 - torch.rand((n, 4), dtype=torch.float32, device=xyz.device, generator=generator)
    uniforms = _f32(uniforms, "uniforms")
    out = torch.empty((n, 3), dtype=torch.float32, device=xyz.device)
    call("gs_sample_from_points", ptr(xyz), ptr(features), ptr(uniforms), n, ptr(out), current_stream(xyz.device))
    return out


def controller_accumulate(ids, num_affected_pixels, magnitude, grad_xyz, num_in_camera, num_pixels, view_grad,
                          view_grad_avg, pos_grad, pos_grad_norm) -> None:
    """The controller's per-iteration statistics (ADC:130-146) in one pass, in place."""
    for t, dt, name in ((ids, torch.int32, "ids"), (num_affected_pixels, torch.int32, "num_affected_pixels"),
                        (num_in_camera, torch.int32, "num_in_camera"), (num_pixels, torch.int32, "num_pixels"),
                        (magnitude, torch.float32, "magnitude"), (grad_xyz, torch.float32, "grad_xyz"),
                        (view_grad, torch.float32, "view_grad"), (view_grad_avg, torch.float32, "view_grad_avg"),
                        (pos_grad, torch.float32, "pos_grad"), (pos_grad_norm, torch.float32, "pos_grad_norm")):
        _require_device(t, name)
        if t.dtype != dt or not t.is_contiguous():
            raise TypeError(f"{name} must be a contiguous {dt} tensor")
    call("gs_controller_accumulate", ptr(ids), ptr(num_affected_pixels), ptr(magnitude), ptr(grad_xyz), ids.shape[0],
         ptr(num_in_camera), ptr(num_pixels), ptr(view_grad), ptr(view_grad_avg), ptr(pos_grad), ptr(pos_grad_norm),
         current_stream(ids.device))


# ---------------------------------------------------------------- fused trainer loss (row F1)
def _image_layout(prediction: torch.Tensor):
    """(buffer, is_hwc, H, W) for a [3,H,W] prediction that is either contiguous or the ``permute(2,0,1)`` view
    of a contiguous [H,W,3] tensor (what the trainer hands over, TRN:170); anything else is copied."""
    if prediction.dim() != 3 or prediction.shape[0] != 3:
        raise ValueError("prediction must be [3,H,W]")
    _, h, w = prediction.shape
    if prediction.is_contiguous():
        return prediction, 0, h, w
    hwc = prediction.permute(1, 2, 0)
    if hwc.is_contiguous():
        return hwc, 1, h, w
    return prediction.contiguous(), 0, h, w


def loss_forward(prediction: torch.Tensor, target: torch.Tensor, lambda_value: float, clamp: bool,
                 need_grad: bool = True):
    """-> (losses f32[3] = {L, L1, 1-SSIM}, saved maps or None).  prediction/target [3,H,W] f32 on the device."""
    prediction, target = _f32(prediction, "prediction", contiguous=False), _f32(target, "target")
    buf, is_hwc, h, w = _image_layout(prediction)
    if tuple(target.shape) != (3, h, w):
        raise ValueError("target must be [3,H,W] like the prediction")
    dev = buf.device
    maps = torch.empty((9, h, w), dtype=torch.float32, device=dev) if need_grad else None
    ws = torch.empty(_lib.load().gs_loss_workspace_floats(h, w), dtype=torch.float32, device=dev)
    losses = torch.empty(3, dtype=torch.float32, device=dev)
    call("gs_loss_forward", ptr(buf), is_hwc, int(clamp), ptr(target), h, w, float(lambda_value), ptr(maps), ptr(ws),
         ptr(losses), current_stream(dev))
    return losses, maps


def loss_backward(prediction: torch.Tensor, target: torch.Tensor, maps: torch.Tensor, lambda_value: float,
                  clamp: bool, grad_total: Optional[torch.Tensor], grad_l1: Optional[torch.Tensor],
                  grad_dssim: Optional[torch.Tensor]) -> torch.Tensor:
    """Gradient w.r.t. ``prediction`` (same shape and memory layout)."""
    buf, is_hwc, h, w = _image_layout(prediction)
    dev = buf.device
    grads = [None if g is None else g.to(device=dev, dtype=torch.float32).contiguous()
             for g in (grad_total, grad_l1, grad_dssim)]
    out = torch.empty_like(buf)
    call("gs_loss_backward", ptr(buf), is_hwc, int(clamp), ptr(target), ptr(maps), h, w, float(lambda_value),
         ptr(grads[0]), ptr(grads[1]), ptr(grads[2]), ptr(out), current_stream(dev))
    return out.permute(2, 0, 1) if is_hwc else out


def scale_regulariser(features: torch.Tensor, point_invalid_mask: torch.Tensor, weight: float = 0.0,
                      grad_features: Optional[torch.Tensor] = None,
                      upstream: Optional[torch.Tensor] = None) -> torch.Tensor:
    """-> f32[2] = {mean ||exp(s)|| over live Gaussians, live count} (LOS:42-54).  With ``grad_features`` (a
    contiguous [N,56] tensor) the same pass adds ``weight * upstream * dR/ds`` into its columns 4..6, in place."""
    features = _f32(features, "point_cloud_features")
    _require_device(point_invalid_mask, "point_invalid_mask")
    if point_invalid_mask.dtype != torch.int8:
        raise TypeError("point_invalid_mask must be int8")
    if grad_features is not None:
        _f32(grad_features, "grad_features", False)
        if not grad_features.is_contiguous() or grad_features.shape != features.shape:
            raise ValueError("grad_features must be a contiguous [N,56] tensor")
    dev = features.device
    ws = torch.empty(_lib.load().gs_scale_regulariser_workspace_floats(), dtype=torch.float32, device=dev)
    out = torch.empty(2, dtype=torch.float32, device=dev)
    call("gs_scale_regulariser", ptr(features), ptr(point_invalid_mask.contiguous()), features.shape[0], float(weight),
         ptr(upstream), ptr(grad_features), ptr(ws), ptr(out), current_stream(dev))
    return out
