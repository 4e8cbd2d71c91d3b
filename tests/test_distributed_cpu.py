"""world_size-2 gloo tests (CPU) of the collective logic of the tile-row sharded path."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, height, width, mode, weights, padded=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from taichi_3d_gaussian_splatting_amd.distributed import (all_gather_tile_rows, all_reduce_accumulators,
                                                              owned_tile_rows)
    th = height // 16
    g = torch.Generator().manual_seed(7)
    full_image = torch.rand(height, width, 3, generator=g)
    full_depth = torch.rand(height, width, generator=g)
    full_count = torch.randint(0, 1000, (height, width), generator=g, dtype=torch.int32)
    # each rank holds only its own tile rows (others are zero, as hip_ops.blend_forward leaves them)
    rows = torch.arange(height) // 16
    own = torch.zeros(height, dtype=torch.bool)
    for r in owned_tile_rows(th, rank, world, mode, weights):
        own |= rows == r
    image = torch.where(own[:, None, None], full_image, torch.zeros_like(full_image)).contiguous()
    depth = torch.where(own[:, None], full_depth, torch.zeros_like(full_depth)).contiguous()
    count = torch.where(own[:, None], full_count, torch.zeros_like(full_count)).contiguous()
    if padded:   # outputs as the sharded rasteriser allocates them: first rows of a padded tensor, un-owned rows garbage
        from taichi_3d_gaussian_splatting_amd.distributed import _padded_base, padded_image_rows

        def as_padded(t):
            base = torch.full((padded_image_rows(height, world),) + t.shape[1:], 77, dtype=t.dtype)
            base[:height][own] = t[own]
            return base[:height]
        image, depth, count = as_padded(image), as_padded(depth), as_padded(count)
        assert all(_padded_base(t, height, world) is not None for t in (image, depth, count))
    all_gather_tile_rows([image, depth, count], rank, world, mode=mode, row_weights=weights)
    assert torch.equal(image, full_image) and torch.equal(depth, full_depth) and torch.equal(count, full_count)

    # gradient accumulators: float columns summed, column 10 summed as int32 bits
    m = 1000
    gens = [torch.Generator().manual_seed(100 + r) for r in range(world)]
    accs = [torch.rand(m, 12, generator=gg) for gg in gens]
    npix = [torch.randint(0, 5000, (m,), generator=gg, dtype=torch.int32) for gg in gens]
    acc = accs[rank].clone()
    acc[:, 10] = npix[rank].view(torch.float32)
    all_reduce_accumulators(acc)
    expect = sum(accs)
    assert torch.allclose(acc[:, :10], expect[:, :10]) and torch.allclose(acc[:, 11], expect[:, 11])
    assert torch.equal(acc[:, 10].contiguous().view(torch.int32), sum(npix))
    dist.destroy_process_group()


def _run(world, height, width, mode="bands", weights=None, padded=False):
    mp.spawn(_worker, args=(world, _free_port(), height, width, mode, weights, padded), nprocs=world, join=True)


def test_all_gather_in_place_for_padded_outputs_world2_and_3():
    """Un-weighted bands in outputs allocated with padded_image_rows(): every tensor is gathered where it lies."""
    _run(2, 17 * 16, 64, padded=True)    # 17 tile rows: bands of 9 + 8 rows, padded to 18
    _run(3, 16 * 16, 48, padded=True)    # 16 tile rows: 6 + 6 + 4, padded to 18


def test_all_gather_tile_rows_and_grad_reduce_world2():
    _run(2, 1072 // 4 // 16 * 16 + 16, 64)   # 17 tile rows: uneven bands (8 + 9)


def test_all_gather_tile_rows_world2_interleaved():
    _run(2, 1072 // 4 // 16 * 16 + 16, 64, mode="interleaved")   # rows 0,2,.. / 1,3,..


def test_all_gather_tile_rows_world3_uneven():
    _run(3, 80, 32)   # 5 tile rows over 3 ranks: 1 + 2 + 2


def test_all_gather_tile_rows_world3_weighted_bands_with_an_empty_band():
    _run(3, 80, 32, weights=[0.0, 0.0, 10.0, 0.5, 0.5])   # bands [0,3) [3,3) [3,5): rank 1 owns nothing


def _compact_stand_in(acc, num_keys):
    """What hip_ops.compact_rows does, in torch (CPU test double of the device stage)."""
    keep = torch.nonzero(num_keys > 0).flatten()
    m = acc.shape[0]
    ids = torch.full((m,), -7, dtype=torch.int32)
    rows = torch.full((m, 12), float("nan"))
    ids[:len(keep)] = keep.to(torch.int32)
    rows[:len(keep)] = acc[keep]
    return ids, rows, torch.tensor([len(keep)], dtype=torch.int32)


def _merge_stand_in(lists, stride, cap, counts, world, m):
    """What hip_ops.merge_rows does: the gathered lists added in rank order (column 10: int32 bits summed as integers)."""
    acc = torch.zeros(m, 12)
    npix = torch.zeros(m, dtype=torch.int32)
    for g in range(world):
        block = lists[g * stride:(g + 1) * stride]
        n = int(counts[g])
        ids = block[:n].long()
        rows = block[cap:cap + 12 * cap].view(torch.float32).view(cap, 12)[:n]
        acc[ids] += rows                      # ids are distinct within a list
        npix[ids] += rows[:, 10].contiguous().view(torch.int32)
    acc[:, 10] = npix.view(torch.float32)
    return acc


def _sparse_worker(rank, world, port, m):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from taichi_3d_gaussian_splatting_amd.distributed import exchange_accumulators_sparse
    # band-like ownership: rank g produced rows of a contiguous id range plus a few "straddlers" of its neighbours
    gens = [torch.Generator().manual_seed(300 + r) for r in range(world)]
    accs, keys = [], []
    for r in range(world):
        lo, hi = r * m // world, (r + 1) * m // world
        nk = torch.zeros(m, dtype=torch.int32)
        nk[max(lo - 5, 0):min(hi + 5, m)] = 1 + torch.randint(0, 9, (min(hi + 5, m) - max(lo - 5, 0),), generator=gens[r],
                                                              dtype=torch.int32)
        nk[torch.randint(0, m, (7,), generator=gens[r])] = 3          # a few large Gaussians seen by several ranks
        a = torch.rand(m, 12, generator=gens[r]) * (nk > 0)[:, None]
        a[:, 10] = (torch.randint(0, 5000, (m,), generator=gens[r], dtype=torch.int32) * (nk > 0)).view(torch.float32)
        a[:, 11] = 0.0
        accs.append(a); keys.append(nk)
    stats = {}
    got = exchange_accumulators_sparse(accs[rank].clone(), keys[rank], compact=_compact_stand_in, merge=_merge_stand_in,
                                       stats=stats)
    expect = torch.zeros(m, 12)
    for r in range(world):                      # rank order
        expect[:, :10] += accs[r][:, :10]
    assert torch.equal(got[:, :10], expect[:, :10])                   # same additions in the same order: bit-exact
    assert torch.equal(got[:, 10].contiguous().view(torch.int32),
                       sum(a[:, 10].contiguous().view(torch.int32) for a in accs))
    assert stats["rows_sent"] == int((keys[rank] > 0).sum()) and stats["rows_sent"] < 0.7 * m
    assert stats["bytes_sent"] < stats["dense_bytes"]
    # every rank ends with the same bits
    digest = got.view(torch.int32).long().sum().view(1)
    everyone = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(everyone, digest)
    assert all(torch.equal(everyone[0], e) for e in everyone)
    # diverged replicas are reported, not summed
    if world > 1:
        try:
            exchange_accumulators_sparse(accs[rank][:m - rank].clone(), keys[rank][:m - rank], compact=_compact_stand_in,
                                         merge=_merge_stand_in)
        except RuntimeError as e:
            assert "diverged" in str(e)
        else:
            raise AssertionError("replicas with different M went unnoticed")
    dist.destroy_process_group()


def test_sparse_accumulator_exchange_world2_and_3():
    """Collective logic of the sparse accumulator exchange (device stages replaced by torch stand-ins): the sum over ranks in
    rank order, bit-identical on every rank, fewer bytes than the dense all-reduce, diverged replicas detected."""
    for world in (2, 3):
        mp.spawn(_sparse_worker, args=(world, _free_port(), 1000), nprocs=world, join=True)


def _dense_fallback_worker(rank, world, port, m):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from taichi_3d_gaussian_splatting_amd.distributed import exchange_accumulators_sparse
    gens = [torch.Generator().manual_seed(700 + r) for r in range(world)]
    accs = []
    for r in range(world):   # every rank produced (nearly) every row: huge Gaussians / the stress distribution
        a = torch.rand(m, 12, generator=gens[r])
        a[:, 10] = torch.randint(0, 5000, (m,), generator=gens[r], dtype=torch.int32).view(torch.float32)
        a[:, 11] = 0.0
        accs.append(a)
    stats = {}
    got = exchange_accumulators_sparse(accs[rank].clone(), torch.ones(m, dtype=torch.int32), compact=_compact_stand_in,
                                       merge=_merge_stand_in, stats=stats)
    assert stats.get("dense_fallback") is True and stats["bytes_sent"] == 48 * m
    expect = sum(a[:, :10].double() for a in accs)
    assert torch.allclose(got[:, :10].double(), expect, rtol=1e-6, atol=1e-6)
    assert torch.equal(got[:, 10].contiguous().view(torch.int32), sum(a[:, 10].contiguous().view(torch.int32) for a in accs))
    dist.destroy_process_group()


def test_sparse_exchange_falls_back_to_the_dense_all_reduce():
    """When the ranks' lists together exceed 1.5 M rows (every rank met most Gaussians) the gathered lists would weigh
    several times the dense array: one all-reduce instead, decided from the gathered counts (the same on every rank)."""
    for world in (2, 3):
        mp.spawn(_dense_fallback_worker, args=(world, _free_port(), 600), nprocs=world, join=True)


def test_owned_rows_partition():
    from taichi_3d_gaussian_splatting_amd.distributed import band_boundaries, owned_tile_rows
    for th in (1, 5, 67):
        for world in (1, 2, 4, 8):
            for mode in ("bands", "interleaved"):
                got = sorted(r for g in range(world) for r in owned_tile_rows(th, g, world, mode))
                assert got == list(range(th))
    # K-balanced bands: boundaries follow the weights, stay a partition, are contiguous
    w = [1.0] * 10 + [9.0] * 10 + [1.0] * 47
    b = band_boundaries(67, 4, w)
    assert b[0] == 0 and b[-1] == 67 and b == sorted(b)
    loads = [sum(w[b[g]:b[g + 1]]) for g in range(4)]
    assert max(loads) <= 1.35 * sum(w) / 4
    assert band_boundaries(67, 8) == [0, 9, 18, 27, 36, 45, 54, 63, 67]   # equal blocks of ceil(67 / 8), last one shorter
    assert band_boundaries(16, 3) == [0, 6, 12, 16] and band_boundaries(2, 4) == [0, 1, 2, 2, 2]
    assert band_boundaries(5, 3, [0.0, 0.0, 10.0, 0.5, 0.5]) == [0, 3, 3, 5]   # the heavy row fills two shares
