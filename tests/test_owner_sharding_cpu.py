"""Host logic of the owner-sharded exchange on CPU (gloo, world size 2 and 3): the chunk all-to-all is a transposition
(received[s] = rank s's send[me]), its inverse returns every row to the slot it was sent from, the chunk capacity is agreed
from one size exchange, and the ranks' point blocks partition the cloud in rank order.  The device stages (routing, key
count, blend) run in tests/test_owner_sharding_gpu.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def _worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from taichi_3d_gaussian_splatting_amd.owner_sharding import _all_to_all_chunks, _chunk_capacity
        gen = torch.Generator().manual_seed(100 + rank)
        counts = torch.randint(0, 200, (world,), generator=gen)                 # records for every band
        mine = torch.cat([counts, torch.tensor([int(counts.sum()) + rank])])    # ... and the visible count
        sizes = torch.empty((world, world + 1), dtype=torch.int64)
        dist.all_gather_into_tensor(sizes.view(-1), mine.to(torch.int64))
        capacity = _chunk_capacity(int(sizes[:, :world].max()))
        assert capacity % 64 == 0 and capacity >= int(sizes[:, :world].max())
        # forward: slot j of chunk b carries (sender, band, j); the header carries the count
        send = torch.full((world, capacity + 1, 16), -1.0)
        for b in range(world):
            send[b, 0, 0] = float(counts[b])
            for j in range(int(counts[b])):
                send[b, 1 + j, :3] = torch.tensor([float(rank), float(b), float(j)])
        recv = _all_to_all_chunks(send, None)
        for s in range(world):
            n = int(sizes[s, rank])
            assert int(recv[s, 0, 0]) == n
            assert torch.equal(recv[s, 1:1 + n, 0], torch.full((n,), float(s)))
            assert torch.equal(recv[s, 1:1 + n, 1], torch.full((n,), float(rank)))
            assert torch.equal(recv[s, 1:1 + n, 2], torch.arange(n, dtype=torch.float32))
        # backward: the band answers slot by slot; the owner finds every answer where it sent the record
        rows = torch.zeros((world, capacity + 1, 12))
        rows[:, :, 0] = recv[:, :, 0] * 1000 + recv[:, :, 2]      # f(sender, slot)
        rows[:, :, 1] = float(rank)                               # the answering band
        back = _all_to_all_chunks(rows, None)
        for b in range(world):
            n = int(counts[b])
            assert torch.equal(back[b, 1:1 + n, 0], rank * 1000 + torch.arange(n, dtype=torch.float32))
            assert torch.equal(back[b, 1:1 + n, 1], torch.full((n,), float(b)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_chunk_exchange_is_a_transposition_and_its_inverse(world):
    mp.spawn(_worker, args=(world, _free_port()), nprocs=world, join=True)


def test_point_blocks_partition_the_cloud_in_rank_order():
    from taichi_3d_gaussian_splatting_amd.owner_sharding import owned_point_rows
    for n in (0, 1, 7, 1000, 1001, 999_999):
        for world in (1, 2, 3, 8):
            blocks = [owned_point_rows(n, r, world) for r in range(world)]
            assert blocks[0].start == 0 and blocks[-1].stop == n
            assert all(a.stop == b.start for a, b in zip(blocks, blocks[1:]))
            assert max(len(b) for b in blocks) - min(len(b) for b in blocks) <= -(-n // world)


def test_balanced_row_weights_move_the_boundaries_only_when_it_pays():
    """The boundaries of the owner-sharded bands follow the ranks' summed walk lengths per tile row -- only when the
    heaviest band carries more than the threshold times the mean (moving them restarts the bands' speculative sizes)."""
    from taichi_3d_gaussian_splatting_amd.distributed import band_boundaries
    from taichi_3d_gaussian_splatting_amd.owner_sharding import balanced_row_weights
    rows, world = 67, 8
    bell = [1000.0 * 2.718281828 ** (-((r - 33) / 12.0) ** 2) for r in range(rows)]     # a trained scene: crowded middle rows
    assert balanced_row_weights([1.0] * rows, world) is None                            # even work: equal bands stay
    assert balanced_row_weights([0.0] * rows, world) is None and balanced_row_weights(bell, 1) is None
    new = balanced_row_weights(bell, world)
    assert new == bell
    equal, moved = band_boundaries(rows, world), band_boundaries(rows, world, new)
    load = lambda b: max(sum(bell[b[g]:b[g + 1]]) for g in range(world)) * world / sum(bell)   # noqa: E731
    assert load(equal) > 1.9 and load(moved) < 1.25
    assert moved[0] == 0 and moved[-1] == rows and all(a <= b for a, b in zip(moved, moved[1:]))
    assert balanced_row_weights(bell, world, current=new) == new                        # balanced now: nothing moves
    drifted = [w * (1.0 + 0.02 * (r % 3)) for r, w in enumerate(bell)]
    assert balanced_row_weights(drifted, world, current=new) == new                     # ... nor for a few per cent


def test_row_weights_follow_a_change_of_resolution():
    """Weights measured on a frame of one height place the bands of a frame of another height (the down-sampled
    iterations of a training run): the density over the image is re-binned, its total and its shape are kept."""
    from taichi_3d_gaussian_splatting_amd import GaussianPointCloudRasterisation as Op
    from taichi_3d_gaussian_splatting_amd.owner_sharding import OwnerShardedRasteriser
    core = OwnerShardedRasteriser(Op.GaussianPointCloudRasterisationConfig(), 1, 4)
    assert core.weights_for(320) is None and core.band_bounds(320) == [0, 5, 10, 15, 20]
    core.row_weights = [0.0] * 10 + [4.0] * 10                       # measured at 20 tile rows: the lower half carries it all
    assert core.weights_for(320) == core.row_weights
    for height in (160, 640, 16 * 7, 16 * 33):
        w = core.weights_for(height)
        th = height // 16
        assert len(w) == th and abs(sum(w) - 40.0) < 1e-9
        assert sum(w[: th // 2]) <= 40.0 / th + 1e-9                 # (an odd row count shares one row between the halves)
        bounds = core.band_bounds(height)
        assert bounds[0] == 0 and bounds[-1] == th and bounds[1] >= th // 2
    assert core.band_rows(320) == range(13, 15)
