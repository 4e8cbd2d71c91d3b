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
