"""Image-space (tile-row) sharding of the rasteriser across the GPUs of one node.

The reference is single-GPU (no collective anywhere); this is the
multi-GPU path BASELINE.json's north_star defines: the point cloud is replicated, GPU ``g`` of ``G``
owns a contiguous band of 16-pixel tile rows (default; interleaved rows ``g, g+G, ...`` as the alternative), runs
binning/sort/blend only for its rows, and the full image is assembled on every rank by all-gathers (RCCL over xGMI;
``backend="nccl"`` is RCCL on ROCm): with the default bands each output tensor is gathered IN PLACE (the rasteriser
allocates it so that every band is an equal slice), otherwise the bands are packed into one buffer and gathered with
one collective.  In the backward pass every rank back-propagates its own tiles and the per-Gaussian accumulators
(48 B x M, not the 236 B x N dense gradients) are summed over the ranks before the per-point chain rule, so every rank ends
up with the full gradient of its replicated parameters: with bands by a SPARSE exchange -- a rank sends only the rows it
produced (~M/G + the Gaussians that straddle its band boundaries), all-gathered and added in rank order, bit-identical on
every rank (``exchange_accumulators_sparse``) --, with interleaved rows by one dense all-reduce.

One process per GPU (``torch.distributed``); works unchanged on ``gloo`` for the CPU tests of the
collective logic (tests/test_distributed_cpu.py).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.distributed as dist

TILE_HEIGHT = 16


def owned_tile_rows(num_tile_rows: int, rank: int, world: int, mode: str = "bands",
                    row_weights: Optional[Sequence[float]] = None) -> range:
    """Tile rows rendered by ``rank``.
    "bands" (default): one contiguous band per rank, so that a Gaussian is binned, sorted and blended by one rank
    (two if it straddles a boundary) and the per-rank list work scales with 1/world; the boundaries split
    ``row_weights`` (e.g. the previous frame's sort keys per tile row) evenly, or the rows when no weights are given.
    "interleaved": rows rank, rank + world, ... (balances any scene, but every rank meets every Gaussian)."""
    if mode == "interleaved":
        return range(rank, num_tile_rows, world)
    if mode != "bands":
        raise ValueError(mode)
    bounds = band_boundaries(num_tile_rows, world, row_weights)
    return range(bounds[rank], bounds[rank + 1])


def uniform_band_rows(num_tile_rows: int, world: int) -> int:
    """Tile rows per band of the un-weighted band split."""
    return -(-num_tile_rows // world)


def padded_image_rows(height: int, world: int) -> int:
    """Pixel rows an output tensor needs so that the un-weighted bands of all ``world`` ranks are equally long slices of
    it (the last ones reaching past the image): the rasteriser allocates its outputs with that many rows and returns the
    first ``height`` -- ``all_gather_tile_rows`` then gathers straight into them."""
    return uniform_band_rows(height // TILE_HEIGHT, world) * world * TILE_HEIGHT


def band_boundaries(num_tile_rows: int, world: int, row_weights: Optional[Sequence[float]] = None) -> list:
    """world + 1 non-decreasing row indices, 0 .. num_tile_rows: band g = rows [b[g], b[g+1]).  With weights, band g
    ends at the first row where the running weight reaches (g+1)/world of the total (deterministic: every rank
    computes the same boundaries from the same replicated weights)."""
    if row_weights is not None and len(row_weights) != num_tile_rows:
        raise ValueError(f"row_weights has {len(row_weights)} entries for {num_tile_rows} tile rows")
    if row_weights is None or float(sum(row_weights)) <= 0.0:
        # equal blocks of ceil(rows / world) rows, the last band(s) shorter: the longest band is as long as in any other
        # even split, and every band sits at a multiple of the block size -- what lets the all-gather run in place
        block = uniform_band_rows(num_tile_rows, world)
        return [min(g * block, num_tile_rows) for g in range(world + 1)]
    total = float(sum(row_weights))
    bounds, run, g = [0], 0.0, 1
    for r, w in enumerate(row_weights):
        run += float(w)
        while g < world and run >= total * g / world:
            bounds.append(r + 1)
            g += 1
    while len(bounds) < world + 1:
        bounds.append(num_tile_rows)
    bounds[-1] = num_tile_rows
    return bounds


def all_gather_tile_rows(tensors: Sequence[torch.Tensor], rank: int, world: int,
                         group: Optional[dist.ProcessGroup] = None, force: bool = False, mode: str = "bands",
                         row_weights: Optional[Sequence[float]] = None) -> None:
    """In place: every tensor is [H, W, ...] with only this rank's tile rows valid; after the call all
    rows are valid on every rank.  One collective per call: the per-rank blocks of all tensors are
    packed into one byte buffer (fewer, larger collectives)."""
    if world == 1 and not force:  # force: issue the collective anyway (single-GPU test of the RCCL call path)
        return
    height = tensors[0].shape[0]
    th = height // TILE_HEIGHT
    if mode == "bands" and row_weights is None and all(_padded_base(t, height, world) is not None for t in tensors):
        # un-weighted bands in outputs allocated with padded_image_rows(): band g IS the g-th equal slice of the padded
        # tensor, so each tensor is gathered where it lies -- no packing, no unpacking (the packed path below costs
        # 3 + 3 (G - 1) small copy kernels per frame).  RCCL/NCCL gather in place when the send buffer is the rank's own
        # slice of the receive buffer; other back ends (gloo on the CPU tests) get a copy of the slice.
        in_place = dist.get_backend(group) == "nccl"
        for t in tensors:
            slices = _padded_base(t, height, world).view(world, -1)
            dist.all_gather_into_tensor(slices.view(-1), slices[rank] if in_place else slices[rank].clone(), group=group)
        return
    rows = [owned_tile_rows(th, g, world, mode, row_weights) for g in range(world)]
    max_rows = max(len(r) for r in rows)
    mine = rows[rank]
    views, spans, total = [], [], 0
    for t in tensors:
        assert t.shape[0] == height and t.is_contiguous()
        v = t.view(th, -1)  # one row of tiles = 16 image rows, contiguous
        nbytes = max_rows * v.shape[1] * v.element_size()
        views.append(v)
        spans.append((total, nbytes))
        total += nbytes
    dev = tensors[0].device
    send = torch.empty(total, dtype=torch.uint8, device=dev)
    for v, (off, nbytes) in zip(views, spans):  # one (strided or contiguous) copy per tensor into the packed send buffer
        send[off: off + nbytes].view(v.dtype).view(max_rows, v.shape[1])[:len(mine)].copy_(
            v[mine.start:mine.stop:mine.step])
    recv = torch.empty(world * total, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv2d = recv.view(world, total)
    for v, (off, nbytes) in zip(views, spans):
        block = recv2d[:, off: off + nbytes].view(v.dtype).unflatten(1, (max_rows, v.shape[1]))   # [G, rows, len]
        for g, r in enumerate(rows):
            if g != rank and len(r) > 0:
                v[r.start:r.stop:r.step].copy_(block[g, :len(r)])


def _padded_base(t: torch.Tensor, height: int, world: int):
    """The padded allocation ``t`` is the first ``height`` rows of (see padded_image_rows), or None."""
    base = t._base
    if base is None or not t.is_contiguous() or not base.is_contiguous() or base.data_ptr() != t.data_ptr() or \
            base.dim() != t.dim() or base.shape[1:] != t.shape[1:] or base.shape[0] != padded_image_rows(height, world):
        return None
    return base


def _check_replicas_agree(m: int, device, group) -> None:
    """The accumulators are [M,12] with M = number of Gaussians in the frustum: the replicated point clouds must agree
    or the exchange below would add rows of different Gaussians (or hang on unequal sizes).  One 16-byte collective per
    backward pass -- small next to the accumulator exchange it protects (a replica that diverges between two sparse
    checks, e.g. a nondeterministic densification on one rank, would otherwise go unnoticed)."""
    sizes = torch.tensor([m, -m], dtype=torch.int64, device=device)
    dist.all_reduce(sizes, op=dist.ReduceOp.MAX, group=group)
    if int(sizes[0]) != m or int(sizes[1]) != -m:
        raise RuntimeError(f"rank {dist.get_rank(group)}: {m} Gaussians in the frustum, other ranks between "
                           f"{-int(sizes[1])} and {int(sizes[0])}: the replicated point clouds have diverged")


def all_reduce_accumulators(acc: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Sum the [M,12] backward accumulators over ranks with ONE dense collective (in place; returns acc).  Column 10
    holds an int32 pixel count in the float's bits (include/gsplat_hip.h): it is converted to a float value for the
    reduction (a pixel count is < 2^24, so the float sum is exact in any order) and back to integer bits afterwards."""
    _check_replicas_agree(acc.shape[0], acc.device, group)
    col = acc[:, 10]
    col.copy_(col.view(torch.int32))            # int32 bits -> float value, one converting copy over the same memory
    dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    col.view(torch.int32).copy_(col)            # the sum of integers below 2^24 is an integer: exact conversion back
    return acc


def exchange_accumulators_sparse(acc: torch.Tensor, num_keys: torch.Tensor, group: Optional[dist.ProcessGroup] = None,
                                 compact=None, merge=None, stats: Optional[dict] = None) -> torch.Tensor:
    """Sum of the [M,12] accumulators over ranks WITHOUT moving the rows a rank did not produce.  With bands a Gaussian
    is blended by one rank (two when it straddles a band boundary), so a rank's accumulator array is ~(1 - 1/G) zeros:
    each rank compacts the rows it produced (``num_keys > 0``) into an ascending (row id, 48-B record) list, the lists
    are all-gathered (one collective; the list lengths -- and the replica check -- travel in a 16-byte one before it)
    and every rank adds them in rank order: identical additions in identical order everywhere, so the replicated
    gradients are bit-identical across ranks and from run to run.  At G = 8 a rank sends ~M/8 + straddlers rows of 52 B
    instead of taking part in a 48 B x M all-reduce.  ``compact`` / ``merge``: the two device stages
    (``hip_ops.compact_rows`` / ``hip_ops.merge_rows``; the CPU tests of the collective logic inject stand-ins)."""
    if compact is None or merge is None:
        from . import hip_ops
        compact, merge = compact or hip_ops.compact_rows, merge or hip_ops.merge_rows
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    m, dev = acc.shape[0], acc.device
    ids, rows, count = compact(acc, num_keys)
    mine = torch.stack([count.reshape(()).to(torch.int64), torch.full((), m, dtype=torch.int64, device=dev)])
    sizes = torch.empty((world, 2), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes.view(-1), mine, group=group)
    host = sizes.tolist()                       # the one host synchronisation of the exchange (list lengths)
    if any(row[1] != m for row in host):
        raise RuntimeError(f"rank {rank}: {m} Gaussians in the frustum, other ranks {[row[1] for row in host]}: the "
                           f"replicated point clouds have diverged")
    counts = [row[0] for row in host]
    if sum(counts) > 1.5 * m:
        # most Gaussians were blended by several ranks (interleaved-like frames, the stress distribution, huge Gaussians):
        # the gathered lists would weigh world x M x 52 B on every rank, several times the dense array -- one all-reduce of
        # the 48 B x M array instead.  Every rank sees the same counts, so every rank takes this branch.
        if stats is not None:
            stats.update(rows_sent=counts[rank], rows_total=sum(counts), capacity=0, bytes_sent=48 * m, dense_bytes=48 * m,
                         dense_fallback=True)
        col = acc[:, 10]
        col.copy_(col.view(torch.int32))
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
        col.view(torch.int32).copy_(col)
        return acc
    cap = max(4, -(-max(counts) // 4) * 4)      # 16-B aligned rows behind the ids
    stride = 13 * cap
    send = torch.empty(stride, dtype=torch.int32, device=dev)
    n_mine = counts[rank]
    send[:n_mine].copy_(ids[:n_mine])
    send[cap:cap + 12 * n_mine].view(torch.float32).copy_(rows[:n_mine].reshape(-1))
    recv = torch.empty(world * stride, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    if stats is not None:
        stats.update(rows_sent=n_mine, rows_total=sum(counts), capacity=cap, bytes_sent=4 * stride,
                     dense_bytes=48 * m)
    return merge(recv, stride, cap, sizes[:, 0].to(torch.int32).contiguous(), world, m)


def shard_rasteriser_across_tile_rows(rasteriser, group: Optional[dist.ProcessGroup] = None, force: bool = False,
                                      mode: str = "bands", accumulator_exchange: Optional[str] = None):
    """Configure a ``GaussianPointCloudRasterisation`` instance for tile-row sharding over ``group`` (see
    ``owned_tile_rows`` for the two modes; ``rasteriser.shard_row_weights`` may be set to per-tile-row weights --
    identical on every rank -- to balance the bands).  accumulator_exchange: "sparse" (default with bands: every rank
    sends only the accumulator rows it produced, ``exchange_accumulators_sparse``) or "dense" (one all-reduce of the
    whole [M,12] array; default with interleaved rows, where every rank meets nearly every Gaussian)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    rasteriser.shard = (rank, world, mode)
    exchange = accumulator_exchange or ("sparse" if mode == "bands" else "dense")
    if exchange not in ("sparse", "dense"):
        raise ValueError(exchange)
    if world > 1 or force:
        rasteriser.image_gather = lambda tensors: all_gather_tile_rows(
            tensors, rank, world, group, force, mode, rasteriser.shard_row_weights)
        rasteriser.exchange_stats = {}
        if exchange == "sparse":
            rasteriser.grad_accumulator_reduce = lambda acc, num_keys: exchange_accumulators_sparse(
                acc, num_keys, group, stats=rasteriser.exchange_stats)
        else:
            rasteriser.grad_accumulator_reduce = lambda acc, num_keys: all_reduce_accumulators(acc, group)
    return rasteriser
