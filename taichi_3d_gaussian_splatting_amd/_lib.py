"""ctypes binding of the gfx950 rasteriser library (C ABI: include/gsplat_hip.h).

The shared library is built in-tree (``csrc/Makefile`` -> ``libgsplat_hip.so`` next to this
file).  There is NO fallback: if the library is missing or a call fails, a RuntimeError is
raised -- the product path never routes through PyTorch eager code or the CPU oracle.
"""
from __future__ import annotations

import ctypes
import threading
import os
import subprocess
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# GS_LIB_PATH: development knob (tuning sweeps load differently built variants of the library); default = in-tree build
LIB_PATH = os.environ.get("GS_LIB_PATH") or os.path.join(_HERE, "libgsplat_hip.so")
ABI_VERSION = 36
ABI_TUNING_OFFSET = 1000   # a tuning build of the library (measurement arms compiled in) reports ABI_VERSION + this

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_F = _c.c_float
_I64 = _c.c_int64

# name -> (restype, argtypes); mirrors include/gsplat_hip.h one to one
_SIGNATURES = {
    "gs_last_error": (_c.c_char_p, []),
    "gs_abi_version": (_I, []),
    "gs_pose_inverse": (_I, [_P, _P, _P, _P, _I, _P]),
    "gs_filter_workspace_bytes": (_c.c_size_t, [_I]),
    "gs_filter_compact": (_I, [_P, _P, _P, _P, _P, _P, _I, _F, _F, _I, _I, _P, _P, _P, _P, _P]),
    "gs_filter_compact_from_poses": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _F, _F, _I, _I, _P, _P, _P, _P, _P]),
    "gs_read_counters": (_I, [_P, _P, _I, _P]),
    "gs_preprocess": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P]),
    "gs_preprocess_geometry": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P,
                                    _P]),
    "gs_view_colours": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "gs_scan_block_sums": (_I, [_P, _I, _P, _I, _P]),
    "gs_scan_block_sums2": (_I, [_P, _P, _I, _P, _P]),
    "gs_scan_block_sums2_to_host": (_I, [_P, _P, _I, _P, _P, _P]),
    "gs_scan_block_sums2_stamped": (_I, [_P, _P, _I, _P, _P, _c.c_uint32, _P]),
    "gs_wait_stamped_sizes": (_I, [_P, _c.c_uint32, _I64, _P]),
    "gs_host_alloc_coherent": (_I, [_I64, _P]),
    "gs_host_free": (_I, [_P]),
    "gs_make_keys": (_I, [_P, _P, _P, _I, _P, _I64, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P]),
    "gs_sort_workspace_bytes": (_c.c_size_t, [_I64]),
    "gs_sort_pairs": (_I, [_P, _P, _P, _P, _I64, _P, _I, _I, _I, _I, _P, _P]),
    "gs_sort_pairs_and_zero": (_I, [_P, _P, _P, _P, _I64, _P, _I, _I, _I, _I, _I, _P, _P, _c.c_size_t, _P, _P, _I, _P]),
    "gs_tile_ranges": (_I, [_P, _I64, _P, _I, _P, _P, _I, _P]),
    "gs_tile_ranges_prezeroed": (_I, [_P, _I64, _P, _I, _P, _P, _I, _I, _P]),
    "gs_frame_struct_bytes": (_c.c_size_t, []),
    "gs_frame_layout": (_I, [_P, _I]),
    "gs_frame_forward": (_I, [_P, _c.c_uint32, _P]),
    "gs_frame_backward": (_I, [_P, _c.c_uint32, _P]),
    "gs_read_counters_async": (_I, [_P, _P, _I, _P]),
    "gs_blend_forward": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "gs_blend_backward": (_I, [_P, _P, _P, _P, _P, _P, _P, _I64, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P]),
    "gs_blend_boundary_bytes": (_c.c_size_t, [_I64, _I, _I]),
    "gs_blend_read_stats": (_I, [_P, _I, _P]),
    "gs_blend_split_workspace_bytes": (_c.c_size_t, [_I, _I]),
    "gs_blend_forward_with_boundaries": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P,
                                              _P, _P, _I64, _P]),
    "gs_blend_forward_split_workspace_bytes": (_c.c_size_t, [_I, _I]),
    "gs_blend_forward_split": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P,
                                    _P, _P, _I64, _P, _P]),
    "gs_blend_backward_split": (_I, [_P, _P, _P, _P, _P, _P, _P, _I64, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P,
                                     _P, _P, _I64, _P, _P]),
    "gs_reduce_partials": (_I, [_P, _P, _P, _P, _I, _P, _P, _I64, _P, _I, _I, _P]),
    "gs_compact_rows_workspace_bytes": (_c.c_size_t, [_I]),
    "gs_compact_rows": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "gs_merge_rows": (_I, [_P, _I64, _I, _P, _I, _I, _P, _P]),
    "gs_route_workspace_bytes": (_c.c_size_t, [_I, _I]),
    "gs_route_count": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "gs_route_scatter": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P]),
    "gs_count_keys": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P]),
    "gs_gather_returned_rows": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "gs_loss_workspace_floats": (_c.c_longlong, [_I, _I]),
    "gs_loss_forward": (_I, [_P, _I, _I, _P, _I, _I, _F, _P, _P, _P, _P]),
    "gs_loss_backward": (_I, [_P, _I, _I, _P, _P, _I, _I, _F, _P, _P, _P, _P, _P]),
    "gs_scale_regulariser_workspace_floats": (_c.c_longlong, []),
    "gs_scale_regulariser": (_I, [_P, _P, _I, _F, _P, _P, _P, _P, _P]),
    "gs_adam_step": (_I, [_P, _P, _P, _P, _c.c_longlong, _c.c_double, _c.c_double, _c.c_double, _c.c_double, _I, _P]),
    "gs_adam_step_features": (_I, [_P, _P, _P, _P, _c.c_longlong, _c.c_double, _c.c_double, _c.c_double, _c.c_double,
                                   _I, _P, _c.c_double, _P, _P]),
    "gs_adam_step_rows": (_I, [_P, _P, _P, _P, _c.c_longlong, _I, _c.c_double, _c.c_double, _c.c_double, _c.c_double, _I,
                           _P, _P, _c.c_double, _P, _P]),
    "gs_controller_accumulate": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P]),
    "gs_ellipsoid_offsets": (_I, [_P, _I, _P, _P]),
    "gs_sample_from_points": (_I, [_P, _P, _P, _I, _P, _P]),
    "gs_point_backward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _F, _F, _F, _F, _F,
                               _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib: Optional[ctypes.CDLL] = None
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "gsplat_hip.h")


def _frame_fields_from_header():
    """ctypes fields of ``GsFrame`` and the GS_FWD_* / GS_BWD_* stage bits, read from include/gsplat_hip.h itself: the
    struct has ~90 members and the header is the one definition (``load`` checks the mirror's size against the library)."""
    import re
    with open(HEADER_PATH) as fh:
        text = fh.read()
    body = re.search(r"typedef struct GsFrame \{(.*?)\} GsFrame;", text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    scalars = {"int32_t": _c.c_int32, "int64_t": _c.c_int64, "float": _c.c_float, "uint32_t": _c.c_uint32}
    fields = []
    for decl in body.split(";"):
        decl = decl.replace("const ", " ").strip()
        if not decl:
            continue
        base, rest = decl.split(None, 1)
        for name in rest.split(","):
            name = name.strip()
            if name.startswith("*"):
                fields.append((name.lstrip("* "), _c.c_void_p))
            else:
                fields.append((name, scalars[base]))
    stages = {m.group(1): 1 << int(m.group(2)) for m in re.finditer(r"#define (GS_(?:FWD|BWD)_\w+)\s+\(1u << (\d+)\)", text)}
    return fields, stages


_FRAME_FIELDS, STAGES = _frame_fields_from_header()


# members whose offsets the library reports (gs_frame_layout, same order as in the header's comment)
FRAME_SENTINELS = ("n_points", "blend_flags", "near_plane", "n_keys_capacity", "xyz", "q_camera_pointcloud", "attrs", "keys",
                   "bin_ranges", "n_bins", "image", "tile_order", "boundary_states", "route_counts", "list_start", "grad_image",
                   "acc", "grad_xyz", "aux_stream", "band_row_bounds")


class GsFrame(ctypes.Structure):
    """Mirror of ``GsFrame`` (include/gsplat_hip.h): pointers as integers (``tensor.data_ptr()``), 0 / None = NULL."""
    _fields_ = _FRAME_FIELDS


def build(verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 with the in-tree Makefile; returns the .so path."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    subprocess.run(cmd, check=True, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                "(or __graft_entry__.build()).  There is no fallback path.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export the symbol
            fn.restype = res
            fn.argtypes = args
        abi = lib.gs_abi_version()
        if abi == ABI_VERSION + ABI_TUNING_OFFSET and os.environ.get("GS_ALLOW_TUNING_LIB") == "1":
            pass   # a measuring tool asked for a tuning build (tools/blend_stats.py, tools/x_arms.sh): never the product path
        elif abi == ABI_VERSION + ABI_TUNING_OFFSET:
            raise RuntimeError(f"{LIB_PATH} is a TUNING build (measurement arms compiled in: its results may be wrong on "
                               "purpose); only the measuring tools load it, with GS_ALLOW_TUNING_LIB=1")
        elif abi != ABI_VERSION:
            raise RuntimeError(f"libgsplat_hip.so ABI {abi} != expected {ABI_VERSION}")
        if lib.gs_frame_struct_bytes() != ctypes.sizeof(GsFrame):
            raise RuntimeError(f"GsFrame: the library's struct has {lib.gs_frame_struct_bytes()} bytes, the Python mirror "
                               f"{ctypes.sizeof(GsFrame)} (include/gsplat_hip.h and the built library disagree)")
        # ... and where it keeps a score of members spread over the struct (the size alone passes two swapped pointers)
        n = lib.gs_frame_layout(None, 0)
        theirs = (ctypes.c_int32 * n)()
        lib.gs_frame_layout(theirs, n)
        ours = [getattr(GsFrame, name).offset for name in FRAME_SENTINELS]
        if n != len(FRAME_SENTINELS) or list(theirs) != ours:
            raise RuntimeError(f"GsFrame: member offsets differ between the library {list(theirs)} and the Python mirror {ours} "
                               "(include/gsplat_hip.h and the built library disagree)")
        _lib = lib
    return _lib


def ptr(t: Optional[torch.Tensor]):
    """Raw device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError("the C ABI takes contiguous buffers")
    return t.data_ptr()


_stream_cache = threading.local()   # per thread: forward runs on the caller's thread, backward on autograd's


def current_stream(device: torch.device) -> int:
    """Handle of torch's current stream on ``device``; inside a ``stream_scope`` the handle looked up at its entry."""
    cached = getattr(_stream_cache, "handle", None)
    if cached is not None and _stream_cache.device == device:
        return cached
    return torch.cuda.current_stream(device).cuda_stream


class stream_scope:
    """``with stream_scope(device):`` -- look the current stream up once for a whole sequence of launches (the operator's
    forward or backward: ~10 look-ups of ~4 us each).  The stream must not be switched inside the scope."""

    def __init__(self, device: torch.device):
        self.device = device

    def __enter__(self):
        self._outer = (getattr(_stream_cache, "handle", None), getattr(_stream_cache, "device", None))
        _stream_cache.handle = torch.cuda.current_stream(self.device).cuda_stream
        _stream_cache.device = self.device
        return self

    def __exit__(self, *exc):
        _stream_cache.handle, _stream_cache.device = self._outer
        return False


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().gs_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed with status {status}: {msg}")


def call(name: str, *args) -> None:
    check(getattr(load(), name)(*args), name)
