"""Owner-sharded Gaussians (owner_sharding.py): rank g owns a contiguous block of point-cloud rows, projects only those,
routes each record to the band(s) it touches, blends the records it receives, returns accumulator rows to the owners.

All ranks are played in ONE process on the one GPU (``simulate_frame``: every rank's device work exactly as under
torch.distributed, the two all-to-alls and the all-gather as device copies), and with real ranks over gloo
(``OwnerShardedRasterisation``).  The bar: image, depth and count of the assembled frame are BIT-IDENTICAL to the
un-sharded operator (the same records reach every tile in the same order); the ranks' gradient blocks, concatenated, are
the un-sharded gradients to summation order (a Gaussian that straddles bands has its per-band sums added in band order
where the un-sharded pass adds its slots in one sequence) -- and bit-identical at world size 1; the hook's integer
fields are exact."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHARD_GRAD_TOL = 2e-4   # as tests/test_fuzz_gpu.py (observed there <= 4.2e-5): the same terms, added per band first


def _inputs(s, rows=None, requires_grad=True):
    from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Op
    sl = slice(None) if rows is None else slice(rows.start, rows.stop)
    xyz = s.point_cloud[sl].clone().requires_grad_(requires_grad)
    feat = s.point_cloud_features[sl].clone().requires_grad_(requires_grad)
    return Op.GaussianPointCloudRasterisationInput(
        point_cloud=xyz, point_cloud_features=feat, point_object_id=s.point_object_id[sl],
        point_invalid_mask=s.point_invalid_mask[sl],
        camera_info=CameraInfo(camera_intrinsics=s.camera_intrinsics, camera_height=s.height, camera_width=s.width,
                               camera_id=0),
        q_pointcloud_camera=s.q_pointcloud_camera, t_pointcloud_camera=s.t_pointcloud_camera, color_max_sh_band=3)


def _config(s):
    from taichi_3d_gaussian_splatting_amd import GaussianPointCloudRasterisation as Op
    return Op.GaussianPointCloudRasterisationConfig(near_plane=s.near_plane, far_plane=s.far_plane,
                                                    depth_to_sort_key_scale=s.depth_to_sort_key_scale)


def _unsharded(s, g, bin_shift=None):
    from taichi_3d_gaussian_splatting_amd import GaussianPointCloudRasterisation as Op
    hooks = []
    op = Op(_config(s), backward_valid_point_hook=hooks.append)
    op.bin_shift = bin_shift
    # (bit-for-bit equality is between UN-SPLIT forward passes: a band's forward is not split, and the split form of a small
    #  frame equals the un-split one in every decision but only to rounding in its sums: tests/test_hip_parity.py)
    op.split_small_grid_forward = False
    inp = _inputs(s)
    image, depth, count = op(inp)
    image.backward(g)
    return image.detach(), depth.detach(), count, inp.point_cloud.grad, inp.point_cloud_features.grad, hooks[0], inp


def _sharded(s, g, world, bin_shift=None, frames=1, row_weights=None):
    from taichi_3d_gaussian_splatting_amd.owner_sharding import OwnerShardedRasteriser, owned_point_rows, simulate_frame
    n = s.point_cloud.shape[0]
    hooks = [[] for _ in range(world)]
    cores = [OwnerShardedRasteriser(_config(s), r, world, backward_valid_point_hook=hooks[r].append) for r in range(world)]
    for c in cores:
        c.bin_shift = bin_shift
        c.row_weights = row_weights
        c.split_small_grid_forward = False   # (bit-for-bit against the un-split baseline: see _unsharded)
    blocks = [owned_point_rows(n, r, world) for r in range(world)]
    for _ in range(frames):
        inputs = [_inputs(s, blocks[r], requires_grad=False) for r in range(world)]
        for h in hooks:
            h.clear()
        image, depth, count, grads = simulate_frame(cores, inputs, g)
    return image, depth, count, grads, [h[0] for h in hooks], blocks, inputs


def _compare(s, g, world, bin_shift=None, frames=1, exact_grads=False, row_weights=None):
    base = _unsharded(s, g, bin_shift)
    image, depth, count, grads, hooks, blocks, inputs = _sharded(s, g, world, bin_shift, frames, row_weights)
    assert torch.equal(base[0], image) and torch.equal(base[1], depth) and torch.equal(base[2], count)
    gx = torch.cat([gr[0] for gr in grads])
    gf = torch.cat([gr[1] for gr in grads])
    for a, b in ((base[3], gx), (base[4], gf)):
        if exact_grads:
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
        else:
            scale = float(a.abs().max())
            assert float((a - b).abs().max()) <= SHARD_GRAD_TOL * max(scale, 1e-30)
            assert torch.equal(a == 0, b == 0)                       # the same rows are visible, band clearing is exact
    # the in-place quaternion normalisation reached the owners' rows (RAS:196-205)
    assert torch.equal(torch.cat([i.point_cloud_features.detach() for i in inputs]), base[6].point_cloud_features.detach())
    # hook fields: every rank reports ITS rows; concatenated (ids shifted by the block start) they are the un-sharded ones
    hb = base[5]
    ids = torch.cat([h.point_id_in_camera_list + blocks[r].start for r, h in enumerate(hooks)])
    assert torch.equal(ids, hb.point_id_in_camera_list)
    assert torch.equal(torch.cat([h.num_overlap_tiles for h in hooks]), hb.num_overlap_tiles)
    assert torch.equal(torch.cat([h.num_affected_pixels for h in hooks]), hb.num_affected_pixels)   # integer sum: exact
    assert torch.equal(torch.cat([h.point_uv_in_camera for h in hooks]), hb.point_uv_in_camera)
    assert torch.equal(torch.cat([h.point_depth for h in hooks]), hb.point_depth)
    mag = torch.cat([h.magnitude_grad_viewspace for h in hooks])
    assert torch.allclose(mag, hb.magnitude_grad_viewspace, rtol=1e-4, atol=1e-12)
    # a rank's magnitude image is its band of the un-sharded one
    from taichi_3d_gaussian_splatting_amd.distributed import band_boundaries
    bounds = band_boundaries(s.height // 16, world, row_weights)
    for r, h in enumerate(hooks):
        sl = slice(bounds[r] * 16, bounds[r + 1] * 16)
        assert torch.equal(h.magnitude_grad_viewspace_on_image[sl], hb.magnitude_grad_viewspace_on_image[sl])
    return base, (image, depth, count, gx, gf)


@pytest.mark.parametrize("world,height,bin_shift", [(1, 144, None), (2, 144, None), (3, 80, 0), (4, 256, 1), (8, 272, None),
                                                    (5, 64, 0)])
def test_owner_sharded_frame_equals_unsharded(world, height, bin_shift):
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image, make_scene
    dev = torch.device("cuda", 0)
    s = make_scene(n=9000, height=height, width=160, s_min=0.01, s_max=0.08, seed=5, invalid_fraction=0.05).to(dev)
    g = make_grad_image(height, 160).to(dev)
    _compare(s, g, world, bin_shift, exact_grads=(world == 1))


@pytest.mark.parametrize("world,weights", [
    (4, [1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 30, 30, 1, 1, 1, 1]),      # the middle rows carry the work: short bands there
    (3, [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 5, 5, 5]),        # everything at the bottom: leading bands of no rows at all
    (8, [3, 1, 4, 1, 5, 9, 2, 6, 5, 3, 5, 8, 9, 7, 9, 3]),
])
def test_owner_sharded_frame_with_balanced_bands_equals_unsharded(world, weights):
    """Band boundaries placed by per-tile-row weights (`row_weights`: bands of unequal height, some of them empty) -- the
    assembled frame is still the un-sharded one bit for bit, gradients and hook fields at the usual bars."""
    from taichi_3d_gaussian_splatting_amd.distributed import band_boundaries
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image, make_scene
    dev = torch.device("cuda", 0)
    s = make_scene(n=9000, height=256, width=160, s_min=0.01, s_max=0.08, seed=6, invalid_fraction=0.05).to(dev)
    g = make_grad_image(256, 160).to(dev)
    bounds = band_boundaries(16, world, weights)
    assert bounds != band_boundaries(16, world)
    _compare(s, g, world, None, frames=2, row_weights=[float(w) for w in weights])


def test_owner_sharded_bands_follow_the_work():
    """The rebalancing loop of OwnerShardedRasterisation, played in one process: the ranks' walk lengths per tile row
    (`row_work`, summed) move the boundaries of a scene that crowds into the lower half of the image; the heaviest band's
    share drops, the frame stays the un-sharded one bit for bit, and a second look leaves the boundaries where they are."""
    from taichi_3d_gaussian_splatting_amd.distributed import band_boundaries
    from taichi_3d_gaussian_splatting_amd.owner_sharding import (OwnerShardedRasteriser, balanced_row_weights, owned_point_rows,
                                                                  simulate_frame)
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image, make_scene
    dev = torch.device("cuda", 0)
    s = make_scene(n=12000, height=320, width=160, s_min=0.01, s_max=0.06, seed=9).to(dev)
    s.point_cloud[:, 1] = s.point_cloud[:, 1].abs() * 0.9 + 0.05     # (y down: everything in the lower half of the frame)
    g = make_grad_image(320, 160).to(dev)
    base = _unsharded(s, g)
    world, rows = 4, 320 // 16
    cores = [OwnerShardedRasteriser(_config(s), r, world) for r in range(world)]
    for c in cores:
        c.bin_shift = 0   # (per-tile lists: a tile's walk length is then a property of the tile alone)
        c.split_small_grid_forward = False   # (bit-for-bit against the un-split baseline: see _unsharded)
    blocks = [owned_point_rows(s.point_cloud.shape[0], r, world) for r in range(world)]
    work = []
    for _ in range(2):
        image, depth, count, _ = simulate_frame(cores, [_inputs(s, b, requires_grad=False) for b in blocks], g, row_work=work)
    assert len(work) == rows and sum(work[: rows // 4]) == 0 and sum(work[rows // 2:]) > 0.8 * sum(work) > 0
    share = lambda b: max(sum(work[b[k]:b[k + 1]]) for k in range(world)) / sum(work)   # noqa: E731
    weights = balanced_row_weights(work, world)
    assert weights is not None and share(band_boundaries(rows, world, weights)) < 0.75 * share(band_boundaries(rows, world))
    for c in cores:
        c.row_weights = weights
    work2 = []
    for _ in range(2):
        image, depth, count, grads = simulate_frame(cores, [_inputs(s, b, requires_grad=False) for b in blocks], g,
                                                    row_work=work2)
    assert torch.equal(base[0], image) and torch.equal(base[1], depth) and torch.equal(base[2], count)
    gf = torch.cat([gr[1] for gr in grads])
    assert float((base[4] - gf).abs().max()) <= SHARD_GRAD_TOL * float(base[4].abs().max())
    assert work2 == work                                             # the walk lengths are the tiles', not the bands'
    assert balanced_row_weights(work2, world, current=weights) == weights


def test_owner_sharded_second_frame_and_large_gaussians():
    """Two frames through the same ranks (the automatic list layout moves), with Gaussians large enough to reach every
    band: a record travels to all ranks and its accumulator rows come back from all of them."""
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image, make_scene
    dev = torch.device("cuda", 0)
    s = make_scene(n=1500, height=192, width=192, s_min=0.05, s_max=0.6, seed=8).to(dev)
    g = make_grad_image(192, 192).to(dev)
    _compare(s, g, 4, None, frames=2)


@pytest.mark.parametrize("seed", range(int(os.environ.get("GS_OWNER_FUZZ_CASES", "10"))))
def test_owner_sharded_random_scenes(seed):
    """The random scenes of tests/test_fuzz_gpu.py (rotated cameras, several objects, invalid rows, needles, planes, tiny
    and non-square images) split over 2-6 owners."""
    from tests.test_fuzz_gpu import random_scene
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image
    dev = torch.device("cuda", 0)
    scene, band, needles, options = random_scene(1000 + seed)
    s = scene.to(dev)
    g = make_grad_image(s.height, s.width, seed=seed).to(dev)
    world = 2 + seed % 5
    _compare(s, g, world, options["bin_shift"])


def test_owner_sharded_is_reproducible_and_empty_ranks_are_fine():
    """More ranks than tile rows (empty bands), a rank whose block has nothing on screen; two runs give the same bits."""
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image, make_scene
    dev = torch.device("cuda", 0)
    s = make_scene(n=4000, height=48, width=96, s_min=0.01, s_max=0.1, seed=2).to(dev)
    s.point_invalid_mask[:1000] = 1                     # rank 0 of 4 owns only invalid rows
    g = make_grad_image(48, 96).to(dev)
    a = _sharded(s, g, 4)
    b = _sharded(s, g, 4)
    assert torch.equal(a[0], b[0])
    for ga, gb in zip(a[3], b[3]):
        assert torch.equal(ga[0].view(torch.int32), gb[0].view(torch.int32))
        assert torch.equal(ga[1].view(torch.int32), gb[1].view(torch.int32))
    _compare(s, g, 4)
    _compare(s, g, 7)                                   # 3 tile rows for 7 ranks


def _free_port():
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def _worker(rank, world, port, height):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from taichi_3d_gaussian_splatting_amd.owner_sharding import OwnerShardedRasterisation, owned_point_rows
        from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image, make_scene
        dev = torch.device("cuda", 0)
        s = make_scene(n=7000, height=height, width=160, s_min=0.01, s_max=0.08, seed=3).to(dev)
        g = make_grad_image(height, 160).to(dev)
        base = _unsharded(s, g)
        hooks = []
        op = OwnerShardedRasterisation(_config(s), backward_valid_point_hook=hooks.append)
        op.core.split_small_grid_forward = False   # (bit-for-bit against the un-split baseline: see _unsharded)
        rows = owned_point_rows(s.point_cloud.shape[0], rank, world)
        for _ in range(2):
            inp = _inputs(s, rows)
            image, depth, count = op(inp)
            image.backward(g)
        assert torch.equal(base[0], image.detach()) and torch.equal(base[1], depth.detach()) and torch.equal(base[2], count)
        sl = slice(rows.start, rows.stop)
        for a, b in ((base[3][sl], inp.point_cloud.grad), (base[4][sl], inp.point_cloud_features.grad)):
            assert float((a - b).abs().max()) <= SHARD_GRAD_TOL * float(base[4].abs().max())
        assert op.last_frame_stats["records_sent"] > 0
        # the second frame ran on the chunk capacity speculated from the first (no blocking size read before the exchange)
        assert op.capacity_stats == {"frames": 2, "redone": 0} and op._capacity_guess >= op.last_frame_stats["capacity"]
        # a frame whose records outgrow the speculated chunks repeats pack, exchange and blend with the exact capacity --
        # every rank takes the decision from the same gathered sizes -- and comes out the same
        op._capacity_guess = 64
        inp = _inputs(s, rows)
        image, depth, count = op(inp)
        image.backward(g)
        assert op.capacity_stats == {"frames": 3, "redone": 1} and op.last_frame_stats["capacity"] > 64
        assert torch.equal(base[0], image.detach()) and torch.equal(base[1], depth.detach()) and torch.equal(base[2], count)
        for a, b in ((base[3][sl], inp.point_cloud.grad), (base[4][sl], inp.point_cloud_features.grad)):
            assert float((a - b).abs().max()) <= SHARD_GRAD_TOL * float(base[4].abs().max())
        # under no_grad the forward alone (inference): the same image
        with torch.no_grad():
            image2 = op(_inputs(s, rows, requires_grad=False))[0]
        assert torch.equal(image2, base[0])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 144), (3, 80)])
def test_owner_sharded_operator_with_real_ranks(world, height):
    """OwnerShardedRasterisation under torch.distributed: two / three processes share the GPU (gloo transport)."""
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), height), nprocs=world, join=True)
