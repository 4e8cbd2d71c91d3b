"""GPU parity tests: every HIP stage (through the C ABI) against the CPU oracle on the same
seeded inputs, then the whole operator.  Integer/index work must be bit-exact; floating point is
held to the tolerance written next to each assertion (north_star: pixel L-inf <= 1e-4).

Each stage test feeds the ORACLE's intermediates to the stage under test, so one faulty kernel
cannot mask (or be masked by) another.
"""
import numpy as np
import pytest
import torch

from oracle import gs_oracle as O
from tests.helpers import (FRAGILE_MARGIN, PIXEL_TOL, close_fraction, dev, oracle_forward, pack_acc, pack_attrs,
                           rel_l2, report, small_scene)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from taichi_3d_gaussian_splatting_amd import hip_ops
    return hip_ops


@pytest.fixture(scope="module")
def scene():
    return small_scene(n=10_000, size=256, seed=0, sh_degree=3, invalid_fraction=0.05)


@pytest.fixture(scope="module")
def ofwd(scene):
    return oracle_forward(scene)


@pytest.fixture(scope="module")
def obwd(scene, ofwd):
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image
    g = make_grad_image(scene.height, scene.width).numpy()
    return g, O.backward(ofwd, g, color_max_sh_band=3)


def test_pose_inverse(ops):
    rng = np.random.default_rng(0)
    q = rng.normal(size=(7, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[3] *= 1.5  # the conjugate is returned un-normalised (UTL:426-432)
    t = rng.normal(size=(7, 3)).astype(np.float32)
    qi, ti = ops.pose_inverse(dev(q), dev(t))
    qo, to = O.inverse_se3_qt(q, t)
    assert np.array_equal(qi.cpu().numpy(), qo)
    assert np.allclose(ti.cpu().numpy(), to, rtol=0, atol=1e-6)


def test_filter_compact_bit_exact(ops, scene, ofwd):
    s = scene
    q_cp, t_cp = dev(ofwd["q_cp"]), dev(ofwd["t_cp"])
    mask, ids, _ = ops.filter_compact(dev(s.point_cloud), dev(s.point_invalid_mask), dev(s.point_object_id),
                                      dev(s.camera_intrinsics), q_cp, t_cp, s.near_plane, s.far_plane, s.width,
                                      s.height)
    assert np.array_equal(mask.cpu().numpy(), ofwd["mask"])
    assert np.array_equal(ids.cpu().numpy(), ofwd["ids"])  # ascending ids: order-preserving compaction


def test_preprocess(ops, scene, ofwd):
    s = scene
    feat = dev(s.point_cloud_features).clone()
    per_tile = ops.ListLayout(bin_shift=0, exact_cull=False)   # the reference's binning: one key per tile of the box
    attrs, ntiles, nowned, block_sums, block_sums_full = ops.preprocess(
        dev(s.point_cloud), feat, dev(s.point_object_id), dev(s.camera_intrinsics), dev(ofwd["q_cp"]),
        dev(ofwd["t_cp"]), dev(ofwd["ids"]), s.width, s.height, per_tile)
    a, ref = attrs.cpu().numpy(), pack_attrs(ofwd)
    assert a.shape[1] == 16
    # projection is evaluated in the oracle's operation order with contraction off: bit-exact
    assert np.array_equal(a[:, 0:3], ref[:, 0:3]), "uv / depth must be bit-exact"
    # rows 1-3 of the record are only written for Gaussians that emit at least one key (they are never
    # gathered otherwise); with the cull off that is every Gaussian whose tile box is not empty
    emits = ofwd["num_overlap_tiles"] > 0
    report("preprocess.emitting", fraction=float(emits.mean()))
    assert np.isinf(a[:, 3]).all() and (a[:, 3] > 0).all()     # cull off: the stored bound is +inf
    for name, sl, rtol in (("conic", slice(4, 7), 2e-5), ("radius", slice(7, 8), 1e-5), ("rgb", slice(8, 11), 2e-6),
                           ("opacity", slice(11, 12), 1e-6), ("amp_and_stop_weight", slice(12, 15), 2e-5),
                           ("rescale", slice(15, 16), 2e-5)):
        rows = emits
        frac = close_fraction(a[rows, sl], ref[rows, sl], rtol=rtol, atol=1e-7)
        report(f"preprocess.{name}", close=frac, max_abs=float(np.abs(a[rows, sl] - ref[rows, sl]).max()))
        assert frac == 1.0, name
    # Since the scale activation is the correctly rounded exponential on both sides (gs_exp_cr / R_EXP_SCALE), the chain
    # scales -> covariance -> radius / conic is IEEE arithmetic in one order: how many rows agree to the last bit
    report("preprocess.bits", radius_identical=float((a[emits, 7] == ref[emits, 7]).mean()),
           conic_identical=float((a[emits, 4:7] == ref[emits, 4:7]).all(axis=1).mean()))
    # in-place quaternion normalisation of visible rows only (RAS:196-205)
    f_hip, f_ref = feat.cpu().numpy(), ofwd["feat"]
    assert np.allclose(f_hip[:, :4], f_ref[:, :4], rtol=0, atol=1e-7)
    assert np.array_equal(f_hip[:, 4:], s.point_cloud_features.numpy()[:, 4:])
    invisible = np.setdiff1d(np.arange(f_hip.shape[0]), ofwd["ids"])
    assert np.array_equal(f_hip[invisible], s.point_cloud_features.numpy()[invisible])
    mism = int((ntiles.cpu().numpy() != ofwd["num_overlap_tiles"]).sum())
    report("preprocess.num_overlap_tiles", mismatches=mism, m=len(ofwd["ids"]))
    assert mism == 0   # integer work: identical (uv and the radius chain are evaluated in the oracle's operation order)
    assert np.array_equal(nowned.cpu().numpy(), ntiles.cpu().numpy())  # 1 GPU owns every row, one key per tile
    sums = np.add.reduceat(ntiles.cpu().numpy(), np.arange(0, len(ofwd["ids"]), 256))
    assert np.array_equal(block_sums.cpu().numpy(), sums)
    assert np.array_equal(block_sums_full.cpu().numpy(), sums)
    # bin layout (heavy scenes): keys per 64 x 64-pixel bin, exact cull on: the box count (hook output) is unchanged, the key
    # count is the number of bins of the box (at most), the cull bound is stored
    feat2 = dev(s.point_cloud_features).clone()
    attrs2, ntiles2, nkeys2, bs2, bsf2 = ops.preprocess(
        dev(s.point_cloud), feat2, dev(s.point_object_id), dev(s.camera_intrinsics), dev(ofwd["q_cp"]),
        dev(ofwd["t_cp"]), dev(ofwd["ids"]), s.width, s.height, ops.ListLayout(bin_shift=2))
    assert torch.equal(ntiles2, ntiles) and torch.equal(bsf2, block_sums_full)
    assert (nkeys2 <= ntiles2).all() and int(nkeys2.sum()) < int(ntiles2.sum())
    a2 = attrs2.cpu().numpy()
    assert np.array_equal(a2[:, 0:3], a[:, 0:3])
    ref_q = pack_attrs(ofwd, exact_cull=True)[:, 3]
    live = emits & (nkeys2.cpu().numpy() > 0)
    assert np.allclose(a2[live, 3], ref_q[live], rtol=1e-5, atol=1e-5)
    box_bins = O_box_bins(ofwd, s.width, s.height, 2)
    assert (nkeys2.cpu().numpy() <= box_bins).all()
    report("preprocess.bins", keys_per_tile_layout=int(ntiles.sum()), keys_bins_no_cull=int(box_bins.sum()),
           keys_bins_cull=int(nkeys2.sum()))


def O_tile_boxes(f, width, height):
    """Tile boxes [t0u, t1u) x [t0v, t1v) of RAS:81-103 restated on the oracle's uv / radii."""
    tw, th = width // 16, height // 16
    u, v = f["uv"][:, 0].astype(np.float32), f["uv"][:, 1].astype(np.float32)
    r = np.maximum(f["radii"].astype(np.float32), np.float32(1.0))
    t0u = np.minimum(np.floor(np.maximum(np.float32(0), u - r) / np.float32(16)).astype(np.int64), tw)
    t1u = np.minimum(np.maximum(np.floor((u + r) / np.float32(16)).astype(np.int64) + 1, t0u + 1), tw)
    t0v = np.minimum(np.floor(np.maximum(np.float32(0), v - r) / np.float32(16)).astype(np.int64), th)
    t1v = np.minimum(np.maximum(np.floor((v + r) / np.float32(16)).astype(np.int64) + 1, t0v + 1), th)
    return t0u, t1u, t0v, t1v


def O_box_bins(f, width, height, bin_shift):
    """Number of (1 << bin_shift)^2-tile bins under each Gaussian's tile box."""
    t0u, t1u, t0v, t1v = O_tile_boxes(f, width, height)
    nu = np.where(t1u > t0u, ((t1u - 1) >> bin_shift) - (t0u >> bin_shift) + 1, 0)
    nv = np.where(t1v > t0v, ((t1v - 1) >> bin_shift) - (t0v >> bin_shift) + 1, 0)
    return nu * nv


def test_keys_sort_ranges_bit_exact(ops, scene, ofwd):
    s = scene
    attrs = dev(pack_attrs(ofwd))
    nt = dev(ofwd["num_overlap_tiles"])
    m = nt.shape[0]
    block_sums = dev(np.add.reduceat(ofwd["num_overlap_tiles"], np.arange(0, m, 256)).astype(np.int32))
    counters = torch.zeros(ops.NUM_COUNTERS, dtype=torch.int32, device="cuda")
    k = ops.scan_block_sums(block_sums, counters)
    assert k == ofwd["keys"].shape[0]
    per_tile = ops.ListLayout(bin_shift=0, exact_cull=False)   # the reference's keys: one per tile of the box
    block_sums_full = block_sums.clone()
    keys, payload, slot_offsets = ops.make_keys(attrs, nt, block_sums, k, s.width, s.height,
                                                s.depth_to_sort_key_scale, per_tile, key_depth_bits=0,
                                                num_overlap_tiles=nt, block_offsets_full=block_sums_full)
    assert np.array_equal(slot_offsets.cpu().numpy(), ofwd["offsets"].astype(np.int32))  # RAS:913-922
    # unsorted keys: same generation order as RAS:161-172
    uk = np.empty(k, np.int64); up = np.empty(k, np.int32)
    import ctypes
    O._lib("f32").gs_oracle_make_keys(O._p(ofwd["uv"]), O._p(ofwd["xyz_cam"]), O._p(ofwd["radii"]),
                                      O._p(ofwd["offsets"]), ctypes.c_int(m), ctypes.c_int(s.width),
                                      ctypes.c_int(s.height), ctypes.c_float(s.depth_to_sort_key_scale),
                                      O._p(uk), O._p(up))
    assert np.array_equal(keys.cpu().numpy(), uk)
    assert np.array_equal(payload.cpu().numpy(), up)
    num_tiles = (s.width // 16) * (s.height // 16)
    db, tb = ops.sort_key_bits(s.near_plane, s.far_plane, s.depth_to_sort_key_scale, num_tiles)
    ops.sort_pairs(keys, payload, db, tb)
    assert np.array_equal(keys.cpu().numpy(), ofwd["keys"])
    assert np.array_equal(payload.cpu().numpy(), ofwd["payload"]), "stable tie order"
    start, end = ops.tile_ranges(keys, num_tiles)
    assert np.array_equal(start.cpu().numpy(), ofwd["tile_start"])
    assert np.array_equal(end.cpu().numpy(), ofwd["tile_end"])
    # compressed 32-bit key layout: same order, same payload permutation, same tile ranges
    kdb, db2, tb2 = ops.key_layout(s.near_plane, s.far_plane, s.depth_to_sort_key_scale, num_tiles)
    assert kdb == db and (db2, tb2) == (db, tb)
    keys32, payload32, none = ops.make_keys(attrs, nt, block_sums, k, s.width, s.height, s.depth_to_sort_key_scale,
                                            per_tile, key_depth_bits=kdb)
    assert none is None   # no slot offsets asked for (inference)
    expect32 = (((uk >> 32) << kdb) | (uk & 0xffffffff)).astype(np.uint32)
    assert np.array_equal(keys32.cpu().numpy().view(np.uint32), expect32)
    ops.sort_pairs(keys32, payload32, db, tb, kdb)
    assert np.array_equal(payload32.cpu().numpy(), ofwd["payload"])
    start32, end32 = ops.tile_ranges(keys32, num_tiles, kdb)
    assert np.array_equal(start32.cpu().numpy(), ofwd["tile_start"])
    assert np.array_equal(end32.cpu().numpy(), ofwd["tile_end"])


def _pipeline(ops, s, ofwd, layout, g=None, arm="two_waves"):
    """HIP stages after the frustum filter under a list layout -> forward outputs (+ debug hit records), sorted keys,
    and (with an upstream gradient g) the backward accumulators.  arm: the blend kernels' form (two waves per tile by
    default here, so that layouts are compared like for like; None = the library's choice by tile count)."""
    feat = dev(s.point_cloud_features).clone()  # fresh copy: preprocess normalises q in place
    a, nfull, nkeys, bsums, bsums_full = ops.preprocess(
        dev(s.point_cloud), feat, dev(s.point_object_id), dev(s.camera_intrinsics), dev(ofwd["q_cp"]),
        dev(ofwd["t_cp"]), dev(ofwd["ids"]), s.width, s.height, layout, s.depth_to_sort_key_scale)
    counters = torch.zeros(ops.NUM_COUNTERS, dtype=torch.int32, device="cuda")
    k, n_slots, _, _ = ops.scan_block_sums(bsums, counters, bsums_full)
    keys, payload, slot_offsets = ops.make_keys(a, nkeys, bsums, k, s.width, s.height, s.depth_to_sort_key_scale,
                                                layout, 0, nfull, bsums_full)
    assert np.array_equal(nfull.cpu().numpy(), ofwd["num_overlap_tiles"])  # hook output is always the box count
    assert np.array_equal(slot_offsets.cpu().numpy(), ofwd["offsets"].astype(np.int32))
    nb = layout.num_bins(s.width, s.height)
    db, tb = ops.sort_key_bits(s.near_plane, s.far_plane, s.depth_to_sort_key_scale, nb)
    ops.sort_pairs(keys, payload, db, tb)
    start, end = ops.tile_ranges(keys, nb)
    out = ops.blend_forward(start, end, payload, a, s.width, s.height, layout, debug_hits=True, arm=arm)
    res = dict(k=k, keys=keys.cpu().numpy(), payload=payload.cpu().numpy(), fwd=[t.cpu() for t in out])
    if g is not None:
        image, depth, acc_alpha, last_eff, count, _ = out
        partials, flags, mag, dbg = ops.blend_backward_partials(start, payload, a, g, acc_alpha, last_eff, slot_offsets,
                                                                n_slots, s.width, s.height, layout, debug_hits=True, arm=arm)
        res.update(acc=ops.reduce_partials(slot_offsets, nfull, flags, partials).cpu(), mag=mag.cpu(), bwd_dbg=dbg.cpu())
    return res


def test_list_layouts_are_output_identical(ops, scene, ofwd, obwd):
    """The reference sorts one key per (tile, Gaussian) of the tile box (RAS:131-172).  The default layout drops the
    pairs that cannot reach alpha >= 1/255; the bin layouts sort one key per (32 | 64 | 128-pixel bin, Gaussian) and let
    every tile recover its own list from its bin's list.  Every combination blends the SAME Gaussians into every pixel in the same order:
    image, depth, accumulated alpha, counts and the per-pixel {count, hash} of blended Gaussians are bit-identical,
    and so are the backward sums."""
    s = scene
    g = dev(obwd[0])
    layouts = {"tile": ops.ListLayout(bin_shift=0, exact_cull=False), "tile+cull": ops.ListLayout(bin_shift=0),
               "bin": ops.ListLayout(bin_shift=2, exact_cull=False), "bin+cull": ops.ListLayout(bin_shift=2),
               "bin8+cull": ops.ListLayout(bin_shift=3), "bin2+cull": ops.ListLayout(bin_shift=1)}
    outs = {name: _pipeline(ops, s, ofwd, lay, g) for name, lay in layouts.items()}
    ref = outs["tile"]
    assert np.array_equal(ref["keys"], ofwd["keys"]) and np.array_equal(ref["payload"], ofwd["payload"])
    report("list_layouts", **{name: o["k"] for name, o in outs.items()})
    assert outs["bin+cull"]["k"] < outs["bin"]["k"] < ref["k"] and outs["tile+cull"]["k"] < ref["k"]
    names = ["image", "depth", "acc_alpha", "last_eff", "count", "debug_hits"]
    for name, o in outs.items():
        for i in (0, 1, 2, 4, 5):  # last_eff is a list position and legitimately differs between layouts
            assert torch.equal(o["fwd"][i], ref["fwd"][i]), (name, names[i])
        assert torch.equal(o["bwd_dbg"], ref["fwd"][5]), name            # backward treats the same pairs as blended
        assert torch.equal(o["acc"].view(torch.int32), ref["acc"].view(torch.int32)), name
        assert torch.equal(o["mag"], ref["mag"]), name
    # the four-waves-per-tile form of the two blend kernels (small grids; per-tile lists taken as they are): every forward
    # output and the hit sets bit-identical, the backward's slot sums equal up to the order of the per-pixel terms
    for name in ("tile", "tile+cull"):
        four = _pipeline(ops, s, ofwd, layouts[name], g, arm="four_waves")
        for i in range(6):
            assert torch.equal(four["fwd"][i], outs[name]["fwd"][i]), (name, names[i])
        assert torch.equal(four["bwd_dbg"], ref["fwd"][5]) and torch.equal(four["mag"], ref["mag"]), name
        a4, a2 = four["acc"], outs[name]["acc"]
        assert torch.equal(a4[:, 10].contiguous().view(torch.int32), a2[:, 10].contiguous().view(torch.int32))
        scale = a2[:, :10].abs().amax(dim=0).clamp_min(1e-30)
        worst = float(((a4[:, :10] - a2[:, :10]).abs() / scale).max())
        report(f"four_waves.{name}", max_scaled_difference_of_slot_sums=worst)
        assert worst < 2e-6
    # every (bin, Gaussian) key the cull dropped stays below 1/255 on all pixels of the bin (float64 check)
    full, kept = outs["bin"], outs["bin+cull"]
    pair = lambda o: (o["keys"] >> 32) * (1 << 32) + o["payload"]  # noqa: E731  (bin, point) identifier
    culled = np.setdiff1d(pair(full), pair(kept))
    assert np.isin(pair(kept), pair(full)).all() and len(culled) > 0
    bins_u = (s.width // 16 + 3) // 4
    b, pt = culled >> 32, culled & 0xffffffff
    uv, conic, al = ofwd["uv"].astype(np.float64), ofwd["conic"].astype(np.float64), ofwd["alpha"].astype(np.float64)
    px = (b % bins_u)[:, None] * 64 + (np.arange(4096) % 64)[None, :] + 0.5
    py = (b // bins_u)[:, None] * 64 + (np.arange(4096) // 64)[None, :] + 0.5
    dx, dy = px - uv[pt, 0:1], py - uv[pt, 1:2]
    e = -0.5 * (dx * dx * conic[pt, 0:1] + dy * dy * conic[pt, 2:3]) - dx * dy * conic[pt, 1:2]
    alpha = np.exp(e) * conic[pt, 3:4] * al[pt, None]
    # only the bin's tiles inside the Gaussian's tile box count: the reference never blends outside the box
    box = O_tile_boxes(ofwd, s.width, s.height)
    in_box = ((px // 16 >= box[0][pt, None]) & (px // 16 < box[1][pt, None]) &
              (py // 16 >= box[2][pt, None]) & (py // 16 < box[3][pt, None]))
    amax = np.where(in_box, alpha, 0.0).max(axis=1)
    report("list_layouts.culled_bin_pairs", n=len(culled), max_alpha=float(amax.max()), threshold=1 / 255)
    assert amax.max() < 1.0 / 255.0


@pytest.mark.parametrize("n,depth_bits,tile_bits,compressed", [
    (1, 17, 13, False), (2, 17, 13, True), (257, 8, 3, True), (5000, 17, 13, False), (5000, 17, 13, True),
    (70_000, 9, 5, True), (300_001, 64, 13, False), (300_001, 19, 13, True),
    # above 512 x 4,096 keys the sort runs its large workgroups (16 rounds per wave), below the small ones (4)
    (2_097_151, 13, 11, True), (2_300_003, 13, 11, True), (2_200_001, 17, 13, False),
    # round 4 (single-sweep passes): more tiles than CUs -- the launch runs in several waves of workgroups -- in both
    # key widths, and one key short / one key over a tile boundary of every tile size (1024 x {1, 2, 4, 8, 11})
    (3_500_003, 11, 11, True), (2_400_001, 17, 13, False), (1024 * 11 * 7, 8, 8, True), (1024 * 8 * 31 + 1, 8, 8, True),
    (1024 * 4 * 100 - 1, 10, 6, True), (1024 * 2 * 200 + 1, 16, 16, True), (1024 * 255, 3, 3, True)])
def test_radix_sort_stable_vs_numpy(ops, n, depth_bits, tile_bits, compressed):
    rng = np.random.default_rng(n)
    if depth_bits == 64:  # negative depths: the full signed key is sorted
        dq = rng.integers(-50, 50, size=n).astype(np.int64)
    else:
        dq = rng.integers(0, 1 << min(depth_bits, 6), size=n).astype(np.int64)  # few values: many ties
        dq[rng.integers(0, n, size=max(1, n // 7))] = (1 << depth_bits) - 1     # and the top of the range
    tile = rng.integers(0, 1 << tile_bits, size=n).astype(np.int64)
    payload = np.arange(n, dtype=np.int32)
    if compressed:
        keys = (dq + (tile << depth_bits)).astype(np.uint32)
        k, p = dev(keys.view(np.int32)), dev(payload)
        ops.sort_pairs(k, p, depth_bits, tile_bits, depth_bits)
        got = k.cpu().numpy().view(np.uint32)
    else:
        keys = dq + (tile << 32)
        k, p = dev(keys), dev(payload)
        ops.sort_pairs(k, p, depth_bits, tile_bits)
        got = k.cpu().numpy()
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(got, keys[order])
    assert np.array_equal(p.cpu().numpy(), payload[order])


@pytest.mark.parametrize("n,capacity", [(100_000, 131_072), (700_001, 1_000_000), (2_900_000, 3_774_096), (0, 5000)])
def test_radix_sort_with_the_count_on_the_device(ops, n, capacity):
    """Speculative frames sort `*n_keys_device` pairs of a capacity-sized buffer (grids and tiles follow the capacity):
    the first n pairs come out stably sorted, nothing past them is read as a key."""
    rng = np.random.default_rng(capacity)
    keys = rng.integers(0, 1 << 22, size=capacity).astype(np.uint32)
    payload = np.arange(capacity, dtype=np.int32)
    k, p = dev(keys.view(np.int32)), dev(payload)
    n_dev = dev(np.array([n], dtype=np.int32))
    k, p = ops.sort_pairs(k, p, 11, 11, 11, in_place=False, n_keys_device=n_dev)
    order = np.argsort(keys[:n], kind="stable")
    assert np.array_equal(k.cpu().numpy().view(np.uint32)[:n], keys[:n][order])
    assert np.array_equal(p.cpu().numpy()[:n], payload[:n][order])


@pytest.mark.parametrize("n,depth_bits,tile_bits,skew", [
    (300_000, 9, 11, "one_bucket"),        # every key in ONE of the 256 buckets of the top digit: 18 chunks of the local sort
    (200_000, 13, 13, "two_buckets"),      # 26 bits: three bucket-local passes of 6 bits, chunked (odd number of passes)
    (50_000, 19, 13, "uniform"),           # 32 bits in use: three local passes of 8 bits
    (1_000_000, 9, 13, "centre_heavy"),    # a trained scene's shape: most keys in a few buckets, some above the LDS capacity
    (70_000, 5, 4, "uniform"),             # 9 bits: a one-bit local pass
    (3_000, 9, 11, "uniform"),             # buckets of a dozen keys
    (4_399_999, 11, 11, "uniform"),        # 1,024 buckets (ten-bit partitioning digit)
    (3_000_000, 11, 9, "centre_heavy"),    # 512 buckets, uneven
])
def test_radix_sort_msd_first_on_skewed_keys(ops, n, depth_bits, tile_bits, skew):
    """The MSD-first sort (one scatter pass on the top eight bits + bucket-local LSD passes in LDS) where its buckets are
    anything but even: buckets far above the LDS capacity are sorted chunk by chunk through the other buffer -- still
    bit-exact against the stable sort, payload included."""
    rng = np.random.default_rng(n + depth_bits)
    bits = depth_bits + tile_bits
    low = bits - 8   # the bucket of a key = its top eight bits in use
    if skew == "one_bucket":
        keys = (5 << low) + rng.integers(0, 1 << low, size=n)
    elif skew == "two_buckets":
        keys = np.where(rng.random(n) < 0.7, 0, 200 << low) + rng.integers(0, 1 << low, size=n)
    elif skew == "centre_heavy":
        tile = np.clip(rng.normal(0.5, 0.04, size=n) * (1 << tile_bits), 0, (1 << tile_bits) - 1).astype(np.int64)
        keys = (tile << depth_bits) + rng.integers(0, 1 << depth_bits, size=n)
    else:
        keys = rng.integers(0, 1 << bits, size=n)
    keys = keys.astype(np.uint64).astype(np.uint32)
    payload = rng.permutation(n).astype(np.int32)
    k, p = dev(keys.view(np.int32)), dev(payload)
    ops.sort_pairs(k, p, depth_bits, tile_bits, depth_bits)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k.cpu().numpy().view(np.uint32), keys[order])
    assert np.array_equal(p.cpu().numpy(), payload[order])


@pytest.mark.parametrize("n,depth_bits,tile_bits,shape", [
    (2_877_171, 9, 11, "centre_heavy"),    # the headline frame's size and key layout: 256 even buckets
    (5_800_000, 9, 11, "centre_heavy"),    # BASELINE config 4: 512 buckets
    (1_000_000, 9, 13, "centre_heavy"),    # a trained scene with per-tile keys
    (48_000, 9, 8, "uniform"),             # 256 tiles: a bucket is one bin
    (400_000, 11, 5, "uniform"),           # fewer bins than buckets: falls back to the top bits, ascending
    (123_457, 12, 20, "uniform"),          # 32 bits in use
    (9_500_000, 9, 13, "uniform"),         # too many pairs for the MSD-first path: LSD passes, ascending
])
def test_radix_sort_with_bins_in_any_order(ops, n, depth_bits, tile_bits, shape):
    """What the frame asks of its sort (gs_sort_pairs_and_zero, bins_in_any_order): every bin's pairs CONTIGUOUS and in
    stable ascending order within the bin; the bins themselves in whatever order the sort leaves them (the MSD-first sort
    partitions by the lowest bits of the bin field: even buckets whatever the density of the scene).  Checked against the
    stable sort: re-ordering the bins of the result stably must give exactly the fully sorted pairs."""
    rng = np.random.default_rng(n)
    if shape == "centre_heavy":
        tile = np.clip(rng.normal(0.5, 0.15, size=n) * (1 << tile_bits), 0, (1 << tile_bits) - 1).astype(np.int64)
    else:
        tile = rng.integers(0, 1 << tile_bits, size=n)
    keys = ((tile << depth_bits) + rng.integers(0, 1 << min(depth_bits, 7), size=n)).astype(np.uint64).astype(np.uint32)
    payload = rng.permutation(n).astype(np.int32)
    ranges = torch.full((2, 1 << tile_bits), -7, dtype=torch.int32, device="cuda")
    k, p, ranges_written = ops.sort_pairs(dev(keys.view(np.int32)), dev(payload), depth_bits, tile_bits, depth_bits,
                                          in_place=False, bins_in_any_order=True, ranges=ranges)
    got_k, got_p = k.cpu().numpy().view(np.uint32), p.cpu().numpy()
    bins = got_k >> depth_bits
    runs = 1 + int((bins[1:] != bins[:-1]).sum())
    assert runs == len(np.unique(tile)), "a bin's pairs are not contiguous"
    by_bin = np.argsort(bins, kind="stable")
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(got_k[by_bin], keys[order])
    assert np.array_equal(got_p[by_bin], payload[order])
    # the ranges of such a result are the ranges of the sorted one, bin by bin -- from gs_tile_ranges and, where the
    # MSD-first path ran on buckets of whole bins, from the sort itself (the same numbers)
    start, end = ops.tile_ranges(k, 1 << tile_bits, depth_bits)
    counts = np.bincount(tile, minlength=1 << tile_bits)
    assert np.array_equal((end - start).cpu().numpy(), counts)
    s_, e_ = start.cpu().numpy(), end.cpu().numpy()
    probe = rng.choice(np.nonzero(counts)[0], size=min(64, int((counts > 0).sum())), replace=False)
    for b in probe:
        assert (bins[s_[b]:e_[b]] == b).all()
    assert ranges_written == (tile_bits >= 8 and n < 9_000_000)
    if ranges_written:
        assert torch.equal(ranges[0], start) and torch.equal(ranges[1], end)
    else:
        assert not bool(ranges.any())   # zero-filled, left to gs_tile_ranges


def test_radix_sort_writes_the_ranges_in_ascending_order_too(ops):
    """Ascending output (bins_in_any_order off) with a partitioning digit inside the bin field: buckets are whole bins,
    the sort writes the ranges; with the digit reaching into the depth field it leaves them to gs_tile_ranges."""
    rng = np.random.default_rng(5)
    for n, depth_bits, tile_bits, written in ((500_000, 9, 12, True), (500_000, 12, 6, False), (1, 9, 12, False),
                                              (40_000, 9, 12, True)):
        keys = rng.integers(0, 1 << (depth_bits + tile_bits), size=n).astype(np.uint32)
        if n > 1000:
            keys[: n // 50] = (3 << depth_bits) + 5     # a bucket with one heavy bin, one bucket of a single key
            keys[n // 50] = ((1 << tile_bits) - 1) << depth_bits
        payload = np.arange(n, dtype=np.int32)
        ranges = torch.full((2, 1 << tile_bits), -7, dtype=torch.int32, device="cuda")
        k, p, got = ops.sort_pairs(dev(keys.view(np.int32)), dev(payload), depth_bits, tile_bits, depth_bits, in_place=False,
                                   ranges=ranges)
        order = np.argsort(keys, kind="stable")
        assert np.array_equal(k.cpu().numpy().view(np.uint32), keys[order]) and np.array_equal(p.cpu().numpy(), payload[order])
        assert got == written
        if got:
            start, end = ops.tile_ranges(k, 1 << tile_bits, depth_bits)
            assert torch.equal(ranges[0], start) and torch.equal(ranges[1], end)


def test_find_tile_start_and_end_known_answer(ops):
    # the reference's own known-answer vector, T_RAS:18-51, through the drop-in symbol
    from taichi_3d_gaussian_splatting_amd.GaussianPointCloudRasterisation import find_tile_start_and_end
    keys = torch.tensor([0x100000000, 0x100000001, 0x200000000, 0x200000001, 0x200000002, 0x300000000,
                         0x300000001], dtype=torch.int64, device="cuda")
    start = torch.zeros(4, dtype=torch.int32, device="cuda")
    end = torch.zeros(4, dtype=torch.int32, device="cuda")
    find_tile_start_and_end(keys, start, end)
    assert start.tolist() == [0, 0, 2, 5] and end.tolist() == [0, 2, 5, 7]


FRAGILE_PIXEL_BOUND = 5e-3   # a flipped 1/255 skip or T' < 1e-4 stop moves a pixel by at most ~alpha*T*c <= 4e-3
REGRESSION_PIXEL_TOL = 5e-6  # 10x the largest non-fragile L-inf observed on any workload (5.1e-7, round 2)


def _check_image(name, hip, ref, fragile):
    """North star: L-inf <= 1e-4 ON EVERY PIXEL (round 5: the pixels whose blend decisions lie within 5e-8 of a threshold --
    `fragile`, the oracle reports them -- are no longer admitted: the kernels take the reference's decision on them too,
    csrc/gs_common.h "threshold decisions"; they are still counted in the report).  The regression bar (5e-6) is 10x what
    is observed."""
    diff = np.abs(hip.astype(np.float64) - ref.astype(np.float64))
    if diff.ndim == 3:
        diff = diff.max(axis=2)
    ok = ~fragile
    flipped = int((diff[fragile] > REGRESSION_PIXEL_TOL).sum())
    report(name, linf_nonfragile=float(diff[ok].max()), linf_all=float(diff.max()),
           fragile_fraction=float(fragile.mean()), fragile_pixels=int(fragile.sum()), flipped_pixels=flipped,
           over_tol_all=int((diff > PIXEL_TOL).sum()))
    assert diff.max() <= PIXEL_TOL
    assert diff.max() <= REGRESSION_PIXEL_TOL and flipped == 0
    assert fragile.mean() < 0.02


def test_blend_forward(ops, scene, ofwd):
    s = scene
    out = ops.blend_forward(dev(ofwd["tile_start"]), dev(ofwd["tile_end"]), dev(ofwd["payload"]),
                            dev(pack_attrs(ofwd)), s.width, s.height, ops.PER_TILE_LISTS)
    image, depth, acc_alpha, last_eff, count = [t.cpu().numpy() for t in out]
    fragile = ofwd["margin"] < FRAGILE_MARGIN
    _check_image("blend_forward.image", image, ofwd["image"], fragile)
    _check_image("blend_forward.acc_alpha", acc_alpha, ofwd["acc_alpha"], fragile)
    assert np.allclose(depth, ofwd["depth"], rtol=1e-4, atol=1e-4)
    assert np.array_equal(last_eff, ofwd["last_eff"])     # every pixel, the fragile ones included
    assert np.array_equal(count, ofwd["count"])


# Gradient bars (relative L2), each within 10x of what is observed (gpurun_out/pytest_r02b.log, round 2):
STAGE_GRAD_TOL = 5e-6     # one kernel vs the oracle on IDENTICAL inputs: observed 1.5e-7 .. 5.1e-7
MASKED_GRAD_TOL = 3e-5    # whole operator, upstream gradient zeroed on the fragile pixels (no flipped threshold decision
                          # can contribute): observed 4.8e-6 (1e4 Gaussians) .. 1.2e-5 (4e5 .. 2e6).  This is the fp32
                          # conditioning of the front end, not the blend kernels: exp(s) differs by an ulp between the
                          # device and glibc and the 2x2 inverse amplifies it (conic: 2e-5 relative, test_preprocess);
                          # test_operator_is_as_close_to_the_f64_spec_as_the_fp32_oracle puts a number on that
FLIP_GRAD_TOL = 1e-4      # whole operator, all pixels: a flipped (pixel, Gaussian) pair is a discrete change of the
                          # gradient; observed 4.8e-6 (1e4 Gaussians) .. 4.0e-5 (4e5, 99 % tied keys)


def _check_acc(name, hip, ref, tol=STAGE_GRAD_TOL, frac_needed=None):
    """Relative L2 below `tol`, and (almost) every entry within rel 2e-4 of the reference (abs floor 2e-6 of the
    largest entry)."""
    if frac_needed is None:
        frac_needed = 0.9999 if tol <= STAGE_GRAD_TOL else 0.999
    scale = float(np.abs(ref).max()) + 1e-30
    frac = close_fraction(hip, ref, rtol=2e-4, atol=2e-6 * scale)
    r = rel_l2(hip, ref)
    report(name, close_fraction=frac, rel_l2=r, scale=scale, tol=tol)
    assert r < tol, name
    assert frac >= frac_needed, name


def _operator_vs_oracle(tag, scene, f, band=3):
    """Whole operator against the oracle: image; gradients with the upstream gradient zeroed on the fragile pixels
    (a flipped threshold decision then contributes nothing: tight bar); gradients on all pixels (loose bar)."""
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image
    fragile = f["margin"] < FRAGILE_MARGIN
    g = make_grad_image(scene.height, scene.width)
    g_masked = g * torch.from_numpy(~fragile)[:, :, None]
    for kind, gi, tol in (("masked", g_masked, MASKED_GRAD_TOL), ("all_pixels", g, FLIP_GRAD_TOL)):
        ob = O.backward(f, gi.numpy(), band)
        image, depth, count, xyz, feat = _run_operator(scene, gi, band)
        if kind == "masked":
            _check_image(f"{tag}.image", image.detach().cpu().numpy(), f["image"], fragile)
            ok = ~fragile
            assert np.array_equal(count.cpu().numpy()[ok], f["count"][ok])
        _check_acc(f"{tag}.{kind}.grad_feat", feat.grad.cpu().numpy(), ob["grad_feat"], tol)
        _check_acc(f"{tag}.{kind}.grad_xyz", xyz.grad.cpu().numpy(), ob["grad_xyz"], tol)


def test_blend_backward(ops, scene, ofwd, obwd):
    s = scene
    g, ob = obwd
    slot_offsets = dev(ofwd["offsets"].astype(np.int32))
    n_slots = int(ofwd["num_overlap_tiles"].sum())
    args = (dev(ofwd["tile_start"]), dev(ofwd["payload"]), dev(pack_attrs(ofwd)), dev(g),
            dev(ofwd["acc_alpha"]), dev(ofwd["last_eff"]), slot_offsets, dev(ofwd["num_overlap_tiles"]), n_slots,
            s.width, s.height, ops.PER_TILE_LISTS)
    acc, mag = ops.blend_backward(*args)
    acc2, mag2 = ops.blend_backward(*args)
    assert torch.equal(acc.view(torch.int32), acc2.view(torch.int32)) and torch.equal(mag, mag2), \
        "the atomic-free backward is bitwise reproducible"
    acc = acc.cpu().numpy()
    names = ["duv_u", "duv_v", "dcov00", "dcov01", "dcov11", "dr", "dg", "db", "dlogit", "magnitude"]
    for c, nme in enumerate(names):
        _check_acc(f"blend_backward.{nme}", acc[:, c], ob["acc"][:, c])
    npix = acc[:, 10].copy().view(np.int32)
    ref_npix = ob["hook"]["num_affected_pixels"]
    mism = int((npix != ref_npix).sum())
    n_fragile = int((ofwd["margin"] < FRAGILE_MARGIN).sum())
    report("blend_backward.num_affected_pixels", mismatched_points=mism,
           total_abs=int(np.abs(npix - ref_npix).sum()), fragile_pixels=n_fragile)
    # integer output: a (pixel, Gaussian) pair can only be counted differently on a fragile pixel, one pair per pixel
    assert int(np.abs(npix - ref_npix).sum()) <= n_fragile
    _check_acc("blend_backward.magnitude_image", mag.cpu().numpy(), ob["hook"]["magnitude_grad_viewspace_on_image"])



def test_point_backward(ops, scene, ofwd, obwd):
    s = scene
    g, ob = obwd
    acc = pack_acc(ob["acc"], ob["hook"]["num_affected_pixels"])
    gx, gf, gxv, gfv = ops.point_backward(
        dev(s.point_cloud), dev(ofwd["feat"]), dev(s.point_object_id), dev(s.camera_intrinsics), dev(ofwd["q_cp"]),
        dev(ofwd["t_cp"]), dev(ofwd["t_pc"]), dev(ofwd["ids"]), dev(acc), dev(pack_attrs(ofwd)), 3, O.GRAD_Q_FACTOR,
        O.GRAD_S_FACTOR, O.GRAD_ALPHA_FACTOR, O.GRAD_COLOR_FACTOR, O.GRAD_HIGH_ORDER_COLOR_FACTOR, want_visible=True)
    gx, gf = gx.cpu().numpy(), gf.cpu().numpy()
    for name, hip, ref in (("xyz", gx, ob["grad_xyz"]), ("q", gf[:, :4], ob["grad_feat"][:, :4]),
                           ("s", gf[:, 4:7], ob["grad_feat"][:, 4:7]), ("logit", gf[:, 7], ob["grad_feat"][:, 7]),
                           ("sh", gf[:, 8:], ob["grad_feat"][:, 8:])):
        scale = float(np.abs(ref).max()) + 1e-30
        frac = close_fraction(hip, ref, rtol=1e-4, atol=1e-6 * scale)
        report(f"point_backward.{name}", close_fraction=frac, rel_l2=rel_l2(hip, ref))
        assert frac >= 0.9999 and rel_l2(hip, ref) < 1e-5
    # rows of points that are not visible are exactly zero; compact hook copies match the dense rows
    invisible = np.setdiff1d(np.arange(gx.shape[0]), ofwd["ids"])
    assert not gx[invisible].any() and not gf[invisible].any()
    assert np.array_equal(gxv.cpu().numpy(), gx[ofwd["ids"]])
    assert np.array_equal(gfv.cpu().numpy(), gf[ofwd["ids"]])


@pytest.mark.parametrize("band,keep", [(0, 1), (1, 4), (2, 9), (3, 16)])
def test_point_backward_sh_band_clearing(ops, scene, ofwd, obwd, band, keep):
    s = scene
    g, ob = obwd
    acc = pack_acc(ob["acc"], ob["hook"]["num_affected_pixels"])
    _, gf, _, _ = ops.point_backward(
        dev(s.point_cloud), dev(ofwd["feat"]), dev(s.point_object_id), dev(s.camera_intrinsics), dev(ofwd["q_cp"]),
        dev(ofwd["t_cp"]), dev(ofwd["t_pc"]), dev(ofwd["ids"]), dev(acc), dev(pack_attrs(ofwd)), band, O.GRAD_Q_FACTOR,
        O.GRAD_S_FACTOR, O.GRAD_ALPHA_FACTOR, O.GRAD_COLOR_FACTOR, O.GRAD_HIGH_ORDER_COLOR_FACTOR, want_visible=False)
    gf = gf.cpu().numpy()
    ref = ob["grad_feat"].copy()  # computed with band 3
    O.clear_grad_by_color_max_sh_band(ref, band)
    for base in (8, 24, 40):
        assert not gf[:, base + keep: base + 16].any()
    assert rel_l2(gf, ref) < 1e-5


# ------------------------------------------------------------------------------- whole operator
def _run_operator(scene, grad_image, band=3, hook=None, row=(0, 1), op=None, row_end=None, bin_shift=None, split_forward=None):
    from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Op
    s = scene.to("cuda")
    xyz = s.point_cloud.clone().requires_grad_(True)
    feat = s.point_cloud_features.clone().requires_grad_(True)
    if op is None:
        op = Op(Op.GaussianPointCloudRasterisationConfig(near_plane=s.near_plane, far_plane=s.far_plane,
                                                         depth_to_sort_key_scale=s.depth_to_sort_key_scale),
                backward_valid_point_hook=hook)
        op.tile_row_begin, op.tile_row_step = row
        if row_end is not None:
            op.tile_row_end = row_end
        if bin_shift is not None:
            op.bin_shift = bin_shift
        if split_forward is not None:
            op.split_small_grid_forward = split_forward
    inp = Op.GaussianPointCloudRasterisationInput(
        point_cloud=xyz, point_cloud_features=feat, point_object_id=s.point_object_id,
        point_invalid_mask=s.point_invalid_mask,
        camera_info=CameraInfo(camera_intrinsics=s.camera_intrinsics, camera_height=s.height,
                               camera_width=s.width, camera_id=0),
        q_pointcloud_camera=s.q_pointcloud_camera, t_pointcloud_camera=s.t_pointcloud_camera,
        color_max_sh_band=band)
    image, depth, count = op(inp)
    if grad_image is not None:
        (image * grad_image.to("cuda")).sum().backward()
    return image, depth, count, xyz, feat


def test_operator_end_to_end(scene, ofwd, obwd):
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image
    g, ob = obwd
    got = {}
    image, depth, count, xyz, feat = _run_operator(scene, make_grad_image(scene.height, scene.width), 3,
                                                   hook=lambda h: got.setdefault("h", h))
    assert image.shape == (scene.height, scene.width, 3) and image.dtype == torch.float32
    assert depth.shape == (scene.height, scene.width) and count.dtype == torch.int32
    fragile = ofwd["margin"] < FRAGILE_MARGIN
    _check_image("operator.image", image.detach().cpu().numpy(), ofwd["image"], fragile)
    ok = ~fragile
    assert np.allclose(depth.detach().cpu().numpy()[ok], ofwd["depth"][ok], rtol=1e-4, atol=1e-4)
    assert np.array_equal(count.cpu().numpy()[ok], ofwd["count"][ok])
    # side effect: visible quaternions normalised in place in the caller's tensor
    assert np.allclose(feat.detach().cpu().numpy()[:, :4], ofwd["feat"][:, :4], atol=1e-7)
    # gradients over all pixels (flips included); the masked comparison is test_operator_masked_gradients
    _check_acc("operator.grad_xyz", xyz.grad.cpu().numpy(), ob["grad_xyz"], FLIP_GRAD_TOL)
    _check_acc("operator.grad_feat", feat.grad.cpu().numpy(), ob["grad_feat"], FLIP_GRAD_TOL)
    h, ho = got["h"], ob["hook"]
    m = len(ofwd["ids"])
    assert np.array_equal(h.point_id_in_camera_list.cpu().numpy(), ho["point_id_in_camera_list"])
    assert h.grad_point_in_camera.shape == (m, 3) and h.grad_pointfeatures_in_camera.shape == (m, 56)
    assert h.grad_viewspace.shape == (m, 2) and h.magnitude_grad_viewspace.shape == (m,)
    assert h.magnitude_grad_viewspace_on_image.shape == (scene.height, scene.width, 2)
    assert h.num_affected_pixels.dtype == torch.int32 and h.num_overlap_tiles.dtype == torch.int32
    assert np.array_equal(h.num_overlap_tiles.cpu().numpy(), ho["num_overlap_tiles"])
    assert np.array_equal(h.point_depth.cpu().numpy(), ho["point_depth"])
    assert np.array_equal(h.point_uv_in_camera.cpu().numpy(), ho["point_uv_in_camera"])
    _check_acc("operator.hook.grad_viewspace", h.grad_viewspace.cpu().numpy(), ho["grad_viewspace"], FLIP_GRAD_TOL)
    _check_acc("operator.hook.grad_pointfeatures", h.grad_pointfeatures_in_camera.cpu().numpy(),
               ho["grad_pointfeatures_in_camera"], FLIP_GRAD_TOL)


def test_operator_masked_gradients(scene, ofwd):
    _operator_vs_oracle("operator10k", scene, ofwd)


def test_operator_is_as_close_to_the_f64_spec_as_the_fp32_oracle(scene, ofwd):
    """Where the remaining operator-level gradient difference comes from: against the float64 build of the oracle
    (the spec), the HIP operator is no further away than the fp32 oracle itself -- both carry the fp32 rounding of the
    projection / conic chain.  Upstream gradient zeroed where either precision sits on a threshold."""
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image
    f64 = oracle_forward(scene, precision="f64")
    same_decisions = (f64["count"] == ofwd["count"]) & (f64["last_eff"] == ofwd["last_eff"])
    keep = same_decisions & (ofwd["margin"] >= FRAGILE_MARGIN) & (f64["margin"] >= FRAGILE_MARGIN)
    g = make_grad_image(scene.height, scene.width) * torch.from_numpy(keep)[:, :, None]
    spec = O.backward(f64, g.numpy().astype(np.float64), 3)
    o32 = O.backward(ofwd, g.numpy(), 3)
    image, depth, count, xyz, feat = _run_operator(scene, g)
    for name, hip, a32, a64 in (("grad_feat", feat.grad.cpu().numpy(), o32["grad_feat"], spec["grad_feat"]),
                                ("grad_xyz", xyz.grad.cpu().numpy(), o32["grad_xyz"], spec["grad_xyz"])):
        e_hip, e_o32 = rel_l2(hip, a64), rel_l2(a32, a64)
        report(f"f64_spec.{name}", hip_vs_f64=e_hip, fp32_oracle_vs_f64=e_o32, kept_pixels=float(keep.mean()))
        assert e_hip <= 3.0 * e_o32 + 1e-6


def _stages_to_ranges(ops, s, layout=None):
    """HIP front end on a device scene -> everything the two blend kernels need."""
    q_cp, t_cp = ops.pose_inverse(s.q_pointcloud_camera, s.t_pointcloud_camera)
    mask, ids, counters = ops.filter_compact(s.point_cloud, s.point_invalid_mask, s.point_object_id,
                                             s.camera_intrinsics, q_cp, t_cp, s.near_plane, s.far_plane, s.width,
                                             s.height)
    feat = s.point_cloud_features.clone()
    layout = layout or ops.ListLayout()
    attrs, ntiles, nowned, bsums, bsums_full = ops.preprocess(
        s.point_cloud, feat, s.point_object_id, s.camera_intrinsics, q_cp, t_cp, ids, s.width, s.height, layout,
        depth_to_sort_key_scale=s.depth_to_sort_key_scale, counters=counters)
    k, n_slots, max_dq, _ = ops.scan_block_sums(bsums, counters, bsums_full)
    num_bins = layout.num_bins(s.width, s.height)
    kdb, db, tb = ops.key_layout(s.near_plane, s.far_plane, s.depth_to_sort_key_scale, num_bins, max_dq)
    keys, payload, slot_offsets = ops.make_keys(attrs, nowned, bsums, k, s.width, s.height, s.depth_to_sort_key_scale,
                                                layout, key_depth_bits=kdb, num_overlap_tiles=ntiles,
                                                block_offsets_full=bsums_full)
    ops.sort_pairs(keys, payload, db, tb, kdb)
    start, end = ops.tile_ranges(keys, num_bins, kdb)
    return dict(attrs=attrs, ntiles=ntiles, payload=payload, start=start, end=end, slot_offsets=slot_offsets,
                n_slots=n_slots, k=k, layout=layout)


def test_screen_filling_gaussians_among_ordinary_ones(ops):
    """A few Gaussians that cover the whole image (floaters close to the camera) among 1e4 ordinary ones: their wave
    counts and writes its (bin, Gaussian) pairs through the lane-shared walk, every other wave as before.  Keys, payload,
    tile counts and ranges stay bit-exact against the oracle (per-tile keys), the binned layouts blend the same pixels,
    and the whole operator matches the oracle."""
    s = small_scene(n=10_000, size=256, seed=5, sh_degree=3)
    heavy = torch.tensor([70, 4100, 4101, 9000])
    s.point_cloud[heavy] = torch.tensor([[0.0, 0.0, 0.0], [0.1, -0.1, 0.2], [-0.2, 0.1, -0.1], [0.05, 0.05, 0.0]])
    s.point_cloud_features[heavy, 4:7] = 1.0     # log-scale: sigma ~ 2.7, the whole view
    s.point_cloud_features[heavy, 7] = -3.0      # faint: blended everywhere, saturates nothing
    f = oracle_forward(s)
    assert (f["num_overlap_tiles"][np.isin(f["ids"], heavy.numpy())] == 256).all()   # every tile of the 16 x 16 grid
    d = s.to("cuda")
    per_tile = _stages_to_ranges(ops, d, ops.ListLayout(bin_shift=0, exact_cull=False))
    assert per_tile["k"] == len(f["keys"])
    assert np.array_equal(per_tile["ntiles"].cpu().numpy(), f["num_overlap_tiles"])
    assert np.array_equal(per_tile["payload"].cpu().numpy(), f["payload"])            # sorted order incl. stable ties
    assert np.array_equal(per_tile["start"].cpu().numpy(), f["tile_start"])
    assert np.array_equal(per_tile["end"].cpu().numpy(), f["tile_end"])
    outs = {}
    for name, layout in (("tile", ops.ListLayout(bin_shift=0)), ("bin2", ops.ListLayout(bin_shift=1)),
                         ("bin4", ops.ListLayout(bin_shift=2))):
        st = _stages_to_ranges(ops, d, layout)
        outs[name] = ops.blend_forward(st["start"], st["end"], st["payload"], st["attrs"], d.width, d.height, layout)
        report(f"screen_filling.{name}", keys=st["k"])
    for name in ("bin2", "bin4"):
        for i in (0, 1, 2, 4):   # image, depth, acc_alpha, count (last_effective is a list position)
            assert torch.equal(outs[name][i], outs["tile"][i]), (name, i)
    _operator_vs_oracle("screen_filling", s, f)


def test_needle_shaped_gaussians(ops):
    """Long, thin Gaussians in every orientation: their reference tile boxes (a square around the 3-sigma circle) are
    mostly empty.  The bin walks look only at the bounding box of the alpha >= 1/255 level set and the slot reduction
    only at the rows that level set crosses in each tile column -- both must lose nothing: keys (cull off) bit-exact, the
    culled layouts blend the same pixels as the reference's lists (the whole operator against the oracle and the f64
    spec: test_needles_against_the_f64_spec)."""
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image
    s, needles = _needle_scene()
    f = oracle_forward(s)
    big = f["num_overlap_tiles"][np.isin(f["ids"], needles.numpy())]
    report("needles", median_box_tiles=float(np.median(big)), max_box_tiles=int(big.max()))
    assert (big > 128).mean() > 0.3      # heavy enough for the wave-shared slot reduction
    d = s.to("cuda")
    ref = _stages_to_ranges(ops, d, ops.ListLayout(bin_shift=0, exact_cull=False))
    assert ref["k"] == len(f["keys"]) and np.array_equal(ref["payload"].cpu().numpy(), f["payload"])
    base = ops.blend_forward(ref["start"], ref["end"], ref["payload"], ref["attrs"], d.width, d.height, ref["layout"])
    for layout in (ops.ListLayout(bin_shift=0), ops.ListLayout(bin_shift=1), ops.ListLayout(bin_shift=2)):
        st = _stages_to_ranges(ops, d, layout)
        out = ops.blend_forward(st["start"], st["end"], st["payload"], st["attrs"], d.width, d.height, layout)
        report(f"needles.bin_shift{layout.bin_shift}", keys=st["k"], keys_without_cull=ref["k"])
        assert st["k"] < ref["k"]
        for i in (0, 1, 2, 4):
            assert torch.equal(out[i], base[i]), (layout.bin_shift, i)
        # backward: the slot reduction, which skips the rows of a heavy Gaussian's box its level set cannot reach, against
        # a plain sum over ALL flagged slots
        g = make_grad_image(d.height, d.width).cuda()
        partials, flags, _ = ops.blend_backward_partials(st["start"], st["payload"], st["attrs"], g, out[2], out[3],
                                                         st["slot_offsets"], st["n_slots"], d.width, d.height, layout)
        acc = ops.reduce_partials(st["slot_offsets"], st["ntiles"], flags, partials, None, st["attrs"], d.width, d.height)
        owner = torch.repeat_interleave(torch.arange(st["ntiles"].shape[0], device="cuda"), st["ntiles"].long())
        raised = flags.bool()
        plain = torch.zeros(acc.shape, dtype=torch.float64, device="cuda")
        plain.index_add_(0, owner[raised], partials[raised].double())
        npix = torch.zeros(acc.shape[0], dtype=torch.int64, device="cuda")
        npix.index_add_(0, owner[raised], partials[raised][:, 10].contiguous().view(torch.int32).long())
        assert torch.equal(acc[:, 10].contiguous().view(torch.int32).long(), npix), "a flagged slot was not visited"
        scale = plain[:, :10].abs().amax(dim=0).clamp_min(1e-30)
        assert ((acc[:, :10].double() - plain[:, :10]).abs() / scale).max() < 2e-6


# A needle's conic is the inverse of a 2x2 matrix with a condition number of several thousand: an ulp in exp(scale) or in a
# product of the projection chain moves its alpha by 1e-4 .. 1e-3 RELATIVE.  Two correct fp32 implementations therefore
# disagree on such a scene far more than on ordinary Gaussians -- the fp32 oracle itself is 3e-4 (image) / 3e-3 (gradients)
# away from the float64 spec build -- so the bar for the HIP operator is the SPEC, measured in units of the fp32 oracle's
# own distance to it, and the pixels left out are the ones the f64 conditioning says can flip: alpha or T' within
# NEEDLE_MARGIN of a threshold in either precision (the largest margin at which the two oracle builds were seen to take
# different decisions on this scene is 1.8e-5).
NEEDLE_MARGIN = 4e-5


def _needle_scene():
    s = small_scene(n=6_000, size=384, seed=11, sh_degree=3)
    rng = np.random.default_rng(5)
    needles = torch.from_numpy(rng.choice(6_000, size=400, replace=False))
    s.point_cloud_features[needles, 4] = torch.from_numpy(rng.uniform(-1.2, -0.2, 400).astype(np.float32))  # long axis
    s.point_cloud_features[needles, 5:7] = torch.from_numpy(rng.uniform(-6.5, -5.0, (400, 2)).astype(np.float32))
    return s, needles


def _operator_vs_f64_spec(tag, s, f32, f64, margin, factor, band=3):
    """HIP operator, fp32 oracle and f64 spec on one scene.  Pixels whose decisions are within `margin` of a threshold
    in either oracle precision, or on which the two precisions blend a different number of Gaussians, are left out
    (image) / get no upstream gradient.  Asserts that the operator is no further from the spec than `factor` times the
    fp32 oracle's own distance; returns the operator's distances to the fp32 oracle for the caller's regression bars."""
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image
    keep = (f32["count"] == f64["count"]) & (f32["margin"] >= margin) & (f64["margin"] >= margin)
    g = make_grad_image(s.height, s.width) * torch.from_numpy(keep)[:, :, None]
    spec = O.backward(f64, g.numpy().astype(np.float64), band)
    o32 = O.backward(f32, g.numpy(), band)
    image, depth, count, xyz, feat = _run_operator(s, g, band)
    img = image.detach().cpu().numpy().astype(np.float64)
    d_hip = np.abs(img - f64["image"]).max(axis=2)
    d_o32 = np.abs(f32["image"].astype(np.float64) - f64["image"]).max(axis=2)
    d_pair = np.abs(img - f32["image"]).max(axis=2)
    report(f"{tag}.image_vs_f64", kept_pixels=float(keep.mean()), hip_max=float(d_hip[keep].max()),
           fp32_oracle_max=float(d_o32[keep].max()), hip_over_1e4=int((d_hip[keep] > PIXEL_TOL).sum()),
           fp32_oracle_over_1e4=int((d_o32[keep] > PIXEL_TOL).sum()), hip_vs_fp32_oracle_max=float(d_pair[keep].max()),
           hip_max_all_pixels=float(d_hip.max()))
    assert d_hip[keep].max() <= factor * d_o32[keep].max() + 1e-6
    assert (d_hip[keep] > PIXEL_TOL).sum() <= factor * (d_o32[keep] > PIXEL_TOL).sum() + 2
    assert d_hip.max() <= FRAGILE_PIXEL_BOUND and np.array_equal(count.cpu().numpy()[keep], f32["count"][keep])
    out = {"image": float(d_pair[keep].max())}
    for name, hip, a32, a64 in (("grad_feat", feat.grad.cpu().numpy(), o32["grad_feat"], spec["grad_feat"]),
                                ("grad_xyz", xyz.grad.cpu().numpy(), o32["grad_xyz"], spec["grad_xyz"])):
        e_hip, e_o32, e_pair = rel_l2(hip, a64), rel_l2(a32, a64), rel_l2(hip, a32)
        report(f"{tag}.{name}_vs_f64", hip=e_hip, fp32_oracle=e_o32, hip_vs_fp32_oracle=e_pair)
        assert e_hip <= factor * e_o32 + 1e-6, name
        out[name] = e_pair
    return out


def test_needles_against_the_f64_spec():
    """the needle scene against the float64 spec.  The HIP operator must be within 2x of the fp32
    oracle's own distance to the spec (image and gradients, pixels near a threshold by the f64 conditioning left out);
    its distance to the fp32 oracle is held to <= 10x what was observed when the bars were set (round 3)."""
    s, _ = _needle_scene()
    f32, f64 = oracle_forward(s), oracle_forward(s, precision="f64")
    d = _operator_vs_f64_spec("needles", s, f32, f64, NEEDLE_MARGIN, factor=2.0)
    assert d["image"] <= NEEDLE_IMAGE_TOL and d["grad_feat"] <= NEEDLE_GRAD_TOL and d["grad_xyz"] <= NEEDLE_GRAD_TOL


# HIP operator vs fp32 oracle on the kept pixels, round 3: image 1.7e-4, grad_feat 2.3e-4, grad_xyz 5.0e-5 (relative L2);
# against the f64 spec the operator sits at 2.8e-4 / 4.6e-4 / 1.8e-4 where the fp32 oracle sits at 3.2e-4 / 4.0e-4 / 1.8e-4
NEEDLE_IMAGE_TOL = 1e-3
NEEDLE_GRAD_TOL = 2e-3


def _chain_scene(n=600, size=96, seed=21):
    """Every pixel blends 320-570 faint, screen-filling Gaussians before it saturates: the longest recurrences the
    operator runs (forward T products, backward T recovery by repeated division, RAS:643)."""
    from taichi_3d_gaussian_splatting_amd.synthetic import make_scene
    s = make_scene(n=n, height=size, width=size, s_min=1.5, s_max=4.0, sh_degree=3, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    s.point_cloud_features[:, 7] = -4.4 + 1.6 * torch.rand(n, generator=g)   # opacity 0.012 .. 0.057
    return s


def test_long_saturating_chains_against_the_f64_spec():
    """the backward pass recovers T by T <- T * rcp(1 - alpha) (1-ulp reciprocal) where the reference
    divides (RAS:643).  On a scene where every pixel walks back through >= 300 blended entries the operator must still be
    as close to the f64 spec as the fp32 oracle (which divides) is -- within a factor 2."""
    s = _chain_scene()
    f32, f64 = oracle_forward(s), oracle_forward(s, precision="f64")
    assert f32["count"].min() >= 120 and (f32["acc_alpha"] > 0.999).all()   # long chains that run into the T' < 1e-4 stop
    d = _operator_vs_f64_spec("chains", s, f32, f64, FRAGILE_MARGIN, factor=2.0)
    report("chains.sizes", min_blended=int(f32["count"].min()), max_blended=int(f32["count"].max()))
    assert d["image"] <= REGRESSION_PIXEL_TOL and d["grad_feat"] <= CHAIN_GRAD_TOL and d["grad_xyz"] <= CHAIN_GRAD_TOL


CHAIN_GRAD_TOL = 5e-5     # observed 5.3e-6 (features) / 8.2e-6 (positions) against the fp32 oracle, round 3


@pytest.mark.parametrize("n,seed", [(131, 31), (137, 32), (150, 33), (259, 34)])
def test_pixels_that_stop_in_the_last_partial_batch(ops, n, seed):
    """the four-waves-per-tile forward counted the list positions of the LAST, partial batch as a full
    batch of 128 when it formed a pixel's stop bracket (the bracket came out narrower than the proven one).  Lists of
    128 + k (and 256 + k) entries on which pixels stop inside the tail batch: every pixel's count and stop position must be
    the oracle's, in the four-wave form the library picks for such a grid and in the two-wave form."""
    from taichi_3d_gaussian_splatting_amd.synthetic import make_scene
    s_cpu = make_scene(n=n, height=64, width=64, s_min=1.5, s_max=4.0, sh_degree=3, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    lo = -2.9 if n < 200 else -3.6    # opacity ~0.05-0.11 (~0.027-0.06): the transmittance reaches 1e-4 near the end of the list
    s_cpu.point_cloud_features[:, 7] = lo + 0.8 * torch.rand(n, generator=g)
    f = oracle_forward(s_cpu, want_margin=False)
    tail_first = (n - 1) // 128 * 128
    stopped = f["acc_alpha"] > 0.9998
    in_tail = stopped & (f["count"] >= tail_first)
    report(f"tail_batch.n{n}", pixels=int(stopped.size), stopped=int(stopped.sum()), stopped_in_tail_batch=int(in_tail.sum()),
           min_count=int(f["count"].min()), max_count=int(f["count"].max()))
    assert in_tail.sum() >= 200, "the scene no longer stops pixels in the tail batch"
    s = s_cpu.to("cuda")
    st = _stages_to_ranges(ops, s, ops.ListLayout(bin_shift=0))
    for arm in ("four_waves", "two_waves", None):
        out = ops.blend_forward(st["start"], st["end"], st["payload"], st["attrs"], s.width, s.height, st["layout"], arm=arm)
        count = out[4].cpu().numpy()
        assert np.array_equal(count, f["count"]), (arm, int((count != f["count"]).sum()))
        assert np.abs(out[0].cpu().numpy() - f["image"]).max() <= 5e-6


@pytest.mark.parametrize("workload,bin_shift", [("cfg2_100k_800", 0), ("headline_1m_1080p", 0), ("headline_1m_1080p", 1),
                                                ("headline_1m_1080p", 2), ("cfg3_400k_1080p", 0)])
def test_forward_and_backward_blend_the_same_pairs(ops, workload, bin_shift):
    """Which (pixel, Gaussian) pairs each pass blends, at full size, against the oracle's two passes -- EVERY pixel, every
    Gaussian, no tolerance: the forward's per-pixel number of blended Gaussians and stop positions equal the oracle's
    forward (RAS:451, RAS:458 decided as the reference decides them: csrc/gs_common.h, "threshold decisions"), the
    backward's number of affected pixels per Gaussian equals the oracle's backward (RAS:631).  The reference evaluates alpha
    by two differently rounded expressions in its two passes (UTL:275-284 / UTL:331-348), so ITS passes disagree on a pair
    in a few millions; each HIP pass follows its counterpart, and the pixels on which the two HIP passes differ (compared
    through {count, hash of the blended Gaussians' list offsets}) are reported and bounded."""
    from oracle import gs_oracle as O
    from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene, make_grad_image
    s_cpu = make_config_scene(workload)
    s = s_cpu.to("cuda")
    st = _stages_to_ranges(ops, s, ops.ListLayout(bin_shift=bin_shift))
    image, depth, acc_alpha, last_eff, count, dbg_f = ops.blend_forward(
        st["start"], st["end"], st["payload"], st["attrs"], s.width, s.height, st["layout"], debug_hits=True)
    assert torch.equal(dbg_f[:, :, 0], count)
    g = make_grad_image(s.height, s.width).cuda()
    partials, flags, mag, dbg_b = ops.blend_backward_partials(
        st["start"], st["payload"], st["attrs"], g, acc_alpha, last_eff, st["slot_offsets"], st["n_slots"],
        s.width, s.height, st["layout"], debug_hits=True)
    differing = int((dbg_f != dbg_b).any(dim=2).sum())
    f = oracle_forward(s_cpu, want_margin=False)
    ob = O.backward(f, make_grad_image(s.height, s.width).numpy(), 3)
    count_mismatch = int((count.cpu().numpy() != f["count"]).sum())
    acc = ops.reduce_partials(st["slot_offsets"], st["ntiles"], flags, partials)
    npix = acc[:, 10].contiguous().view(torch.int32).cpu().numpy()
    npix_mismatch = int((npix != ob["hook"]["num_affected_pixels"]).sum())
    report(f"hit_sets.{workload}.bin_shift{bin_shift}", pixels=s.height * s.width, keys=st["k"], blended_pairs=int(count.sum()),
           pixels_with_different_count_than_the_oracle=count_mismatch,
           gaussians_with_different_pixel_count_than_the_oracle_backward=npix_mismatch,
           pixels_on_which_the_two_passes_differ=differing, backward_pairs=int(npix.sum()))
    assert count_mismatch == 0
    assert npix_mismatch == 0
    assert differing <= max(4, int(count.sum()) // 1_000_000)
    # and the debug build changes nothing: same partial sums as the production kernel, bit for bit
    partials2, flags2, mag2 = ops.blend_backward_partials(
        st["start"], st["payload"], st["attrs"], g, acc_alpha, last_eff, st["slot_offsets"], st["n_slots"],
        s.width, s.height, st["layout"])
    raised = flags.bool()
    assert torch.equal(flags, flags2) and torch.equal(mag, mag2)
    assert torch.equal(partials[raised].view(torch.int32), partials2[raised].view(torch.int32))


def test_four_waves_per_tile_arm_at_a_larger_size(ops):
    """The small-grid arm of the two blend kernels (four waves per tile, one pixel per lane) forced on a 2,500-tile frame
    and compared with the two-wave kernels on the same lists: every forward output bit-identical, identical hit sets in
    both passes, slot sums equal up to the order of the per-pixel terms; the default picks it by tile count."""
    from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene, make_grad_image
    s = make_config_scene("cfg2_100k_800").to("cuda")
    st = _stages_to_ranges(ops, s, ops.ListLayout(bin_shift=0))
    g = make_grad_image(s.height, s.width).cuda()
    out, part = {}, {}
    for arm in ("two_waves", "four_waves", None):
        out[arm] = ops.blend_forward(st["start"], st["end"], st["payload"], st["attrs"], s.width, s.height, st["layout"],
                                     debug_hits=True, arm=arm)
        part[arm] = ops.blend_backward_partials(st["start"], st["payload"], st["attrs"], g, out[arm][2], out[arm][3],
                                                st["slot_offsets"], st["n_slots"], s.width, s.height, st["layout"],
                                                debug_hits=True, arm=arm)
    for i in range(6):
        assert torch.equal(out["four_waves"][i], out["two_waves"][i]), i
    p2, f2, m2, d2 = part["two_waves"]
    p4, f4, m4, d4 = part["four_waves"]
    assert torch.equal(d4, d2) and torch.equal(f4, f2) and torch.equal(m4, m2)
    assert int((d4 != out["two_waves"][5]).any(dim=2).sum()) <= 2   # (backward vs forward pairs: see test_forward_and_backward_...)
    raised = f2.bool()
    assert torch.equal(p4[raised][:, 10].contiguous().view(torch.int32), p2[raised][:, 10].contiguous().view(torch.int32))
    a2 = ops.reduce_partials(st["slot_offsets"], st["ntiles"], f2, p2)
    a4 = ops.reduce_partials(st["slot_offsets"], st["ntiles"], f4, p4)
    scale = a2[:, :10].abs().amax(dim=0).clamp_min(1e-30)
    worst = float(((a4[:, :10] - a2[:, :10]).abs() / scale).max())
    report("four_waves.cfg2", tiles=(s.width // 16) * (s.height // 16), max_scaled_difference_of_sums=worst)
    assert worst < 2e-6
    # 2,500 tiles <= 3,840: the library's own choice here is the four-wave form, bit for bit
    assert torch.equal(part[None][0][raised].view(torch.int32), p4[raised].view(torch.int32))
    for i in range(6):
        assert torch.equal(out[None][i], out["four_waves"][i]), i


@pytest.mark.parametrize("workload", ["cfg2_100k_800", "headline_1m_1080p"])
def test_one_wave_per_tile_backward(ops, workload):
    """The backward blend with ONE wave per tile (four pixels per lane, cross-lane sums through LDS: round 6) against the
    two-wave kernel on the same lists: which pairs are blended (per-pixel count + hash), the slot flags and the per-pixel
    |grad uv| image bit for bit, every slot's pixel count equal, slot sums equal up to the order of the per-pixel terms;
    bitwise reproducible from run to run.  (An explicit arm: the library's own choice on large grids stays the two-wave kernel
    with the same LDS reduction, which measured faster -- profiles/r06_backward_arms.md.)"""
    from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene, make_grad_image
    s = make_config_scene(workload).to("cuda")
    st = _stages_to_ranges(ops, s, ops.ListLayout(bin_shift=0))
    g = make_grad_image(s.height, s.width).cuda()
    fwd = ops.blend_forward(st["start"], st["end"], st["payload"], st["attrs"], s.width, s.height, st["layout"], arm="two_waves")
    part = {}
    for arm in ("two_waves", "one_wave", "one_wave_again", None, "two_waves_skewed"):
        part[arm] = ops.blend_backward_partials(st["start"], st["payload"], st["attrs"], g, fwd[2], fwd[3], st["slot_offsets"],
                                                st["n_slots"], s.width, s.height, st["layout"], debug_hits=True,
                                                arm=None if arm is None else arm.replace("_again", ""))
    p2, f2, m2, d2 = part["two_waves"]
    p1, f1, m1, d1 = part["one_wave"]
    assert torch.equal(d1, d2) and torch.equal(f1, f2) and torch.equal(m1, m2)
    raised = f2.bool()
    assert torch.equal(p1[raised][:, 10].contiguous().view(torch.int32), p2[raised][:, 10].contiguous().view(torch.int32))
    assert torch.equal(p1[raised].view(torch.int32), part["one_wave_again"][0][raised].view(torch.int32))   # reproducible
    a2 = ops.reduce_partials(st["slot_offsets"], st["ntiles"], f2, p2)
    a1 = ops.reduce_partials(st["slot_offsets"], st["ntiles"], f1, p1)
    scale = a2[:, :10].abs().amax(dim=0).clamp_min(1e-30)
    worst = float(((a1[:, :10] - a2[:, :10]).abs() / scale).max())
    report(f"one_wave.{workload}", tiles=(s.width // 16) * (s.height // 16), max_scaled_difference_of_sums=worst)
    assert worst < 2e-6
    assert torch.equal(part[None][2], m1) and torch.equal(part[None][1], f1)
    # GS_BLEND_SKEWED_WALKS: the two-wave kernel with its cross-lane sums in registers (what the operator picks for frames with
    # skewed walk lengths): the same pairs, flags, counts and |grad uv| image, sums equal to rounding
    ps, fs, ms, ds = part["two_waves_skewed"]
    assert torch.equal(ds, d2) and torch.equal(fs, f2) and torch.equal(ms, m2)
    assert torch.equal(ps[raised][:, 10].contiguous().view(torch.int32), p2[raised][:, 10].contiguous().view(torch.int32))
    a_s = ops.reduce_partials(st["slot_offsets"], st["ntiles"], fs, ps)
    worst_s = float(((a_s[:, :10] - a2[:, :10]).abs() / scale).max())
    report(f"two_waves_skewed.{workload}", max_scaled_difference_of_sums=worst_s)
    assert worst_s < 2e-6


@pytest.mark.parametrize("size,n,bin_shift", [(256, 10_000, 0), (512, 60_000, 0), (128, 6_000, 0), (400, 2_000, 0)])
def test_split_backward_on_small_grids(ops, size, n, bin_shift):
    """List splitting (include/gsplat_hip.h): on grids of at most 1,024 tiles with per-tile lists the forward pass leaves
    every pixel's transmittance and colour at every 128th position of its tile's list, and the backward pass gives a tile
    four (<= 512 tiles) or two workgroups, each starting from such a state.  Against the un-split backward on the same
    lists: identical hit sets and slot flags, pixel counts exact, slot sums and the per-Gaussian accumulators equal to
    rounding (the state at a cut comes from the forward pass instead of from divisions / a running sum), the |grad uv|
    image likewise -- and two split runs give the same bits.  Lists of 200-600 entries (several cuts per tile), of a
    dozen (no cut at all: 400 x 400 with 2,000 Gaussians)."""
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image, make_scene
    s = make_scene(n=n, height=size, width=size, s_min=0.01, s_max=0.08, seed=size + bin_shift).to("cuda")
    layout = ops.ListLayout(bin_shift=bin_shift)
    st = _stages_to_ranges(ops, s, layout)
    g = make_grad_image(size, size).cuda()
    emit = bin_shift > 0
    n_list = st["payload"].shape[0] << (2 * bin_shift if emit else 0)
    nbytes = ops.boundary_states_bytes(n_list, size, size, layout, emit)
    assert nbytes > 0
    boundary = torch.full((nbytes,), 0x7f, dtype=torch.uint8, device="cuda")      # NaN-ish garbage: only written slots are read
    work = torch.empty(ops.num_owned_tiles(size, size, layout), dtype=torch.int32, device="cuda")
    fwd = ops.blend_forward(st["start"], st["end"], st["payload"], st["attrs"], size, size, layout, ordered=True,
                            tile_work=work, emit_walked_lists=emit, boundary=boundary, debug_hits=True)
    plain = ops.blend_forward(st["start"], st["end"], st["payload"], st["attrs"], size, size, layout, debug_hits=True)
    for i in ((0, 1, 2, 4) if emit else range(5)):   # (walked lists: last_effective counts positions of the tile's own list)
        assert torch.equal(fwd[i], plain[i]), i                                    # recording the states changes nothing
    image, acc_alpha, last_eff = fwd[0], fwd[2], fwd[3]
    if emit:
        b_start, b_list, b_layout = fwd[5], fwd[6], ops.walked_layout(layout)
    else:
        b_start, b_list, b_layout = st["start"], st["payload"], layout
    args = (b_start, b_list, st["attrs"], g, acc_alpha, last_eff, st["slot_offsets"], st["n_slots"], size, size, b_layout)
    ws = ops.Workspaces()
    p0, f0, m0, d0 = [t.clone() for t in ops.blend_backward_partials(*args, tile_work=work, debug_hits=True)]
    runs = []
    for _ in range(2):
        out = ops.blend_backward_partials(*args, tile_work=work, debug_hits=True, ws=ws, image=image, boundary=boundary)
        runs.append([t.clone() for t in out])
    p1, f1, m1, d1 = runs[0]
    assert torch.equal(d1, d0)                                                    # the same (pixel, Gaussian) pairs
    # (against the FORWARD pass's pairs: each pass follows its reference counterpart, whose two expressions for alpha round
    #  differently -- a pair in a few millions sits between them)
    assert int((d1 != fwd[-1]).any(dim=2).sum()) <= 2
    assert torch.equal(f1, f0)
    raised = f0.bool()
    assert torch.equal(p1[raised][:, 10].contiguous().view(torch.int32), p0[raised][:, 10].contiguous().view(torch.int32))
    for a, b in zip(runs[0], runs[1]):                                            # bitwise reproducible
        assert torch.equal(a[raised].view(torch.int32) if a is runs[0][0] else a, b[raised].view(torch.int32) if b is runs[1][0] else b)
    a0 = ops.reduce_partials(st["slot_offsets"], st["ntiles"], f0, p0)
    a1 = ops.reduce_partials(st["slot_offsets"], st["ntiles"], f1, p1)
    scale = a0[:, :10].abs().amax(dim=0).clamp_min(1e-30)
    worst = float(((a1[:, :10] - a0[:, :10]).abs() / scale).max())
    rel = float((a1[:, :10] - a0[:, :10]).norm() / a0[:, :10].norm())
    mag = float((m1 - m0).abs().max() / m0.abs().max().clamp_min(1e-30))
    report("split_backward", size=size, bin_shift=bin_shift, tiles=(size // 16) ** 2, max_scaled_difference_of_sums=worst,
           rel_l2_of_sums=rel, magnitude_image=mag)
    assert worst < 5e-6 and rel < 2e-6 and mag < 1e-5


@pytest.mark.parametrize("size,n,state,rgb_only", [(256, 10_000, True, False), (512, 60_000, True, False), (128, 6_000, True, False),
                                                   (400, 2_000, True, False), (256, 10_000, False, False), (256, 10_000, False, True),
                                                   (192, 30_000, True, False), (96, 30_000, True, False)])
def test_split_forward_on_small_grids(ops, size, n, state, rgb_only):
    """Forward list splitting (include/gsplat_hip.h, round 6): on grids of at most 320 tiles with per-tile lists a tile gets
    four workgroups forward -- probe, blend from the product of the segments in front, combine (forced here on the larger
    grids too, where it does not pay and is not the default).  Against
    the un-split forward on the same lists: the SAME (pixel, Gaussian) pairs blended on every pixel (count + hash), the same
    per-pixel count and last effective position, image / depth / transmittance equal to rounding; two split runs give the
    same bits; and the split backward pass, started from the boundary states the split forward leaves, gives the un-split
    backward's pairs and its sums to rounding.  Lists of 200-600 entries (every segment holds batches), of a dozen (one
    batch: the other segments are empty), and -- 192 x 192 with 30,000 Gaussians -- lists that run into the T' < 1e-4 stop
    in the first segments, so that later ones start dead; 96 x 96 with 30,000: lists of 1,000-3,000 (several batches per
    segment, pixels that stop in the middle of a backward segment whose upper cut lies in a forward segment they enter dead:
    the state such a cut hands the backward pass must be the pixel's FINAL transmittance -- fuzz draw 12174 found the first
    form of the combine step handing it the product of the probes instead)."""
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image, make_scene
    kw = dict(s_min=0.03, s_max=0.12) if (n == 30_000 and size == 192) else dict(s_min=0.01, s_max=0.08)
    s = make_scene(n=n, height=size, width=size, seed=size + 7, **kw).to("cuda")
    layout = ops.ListLayout(bin_shift=0)
    st = _stages_to_ranges(ops, s, layout)
    assert ops.forward_split_bytes(size, size, layout, force=True) > 0
    assert (ops.forward_split_bytes(size, size, layout) > 0) == ((size // 16) ** 2 <= 320)
    common = dict(rgb_only=rgb_only, need_state=state, debug_hits=True)
    boundary = {}
    if state:
        nbytes = ops.boundary_states_bytes(st["payload"].shape[0], size, size, layout, False)
        boundary = {k: torch.full((nbytes,), 0x7f, dtype=torch.uint8, device="cuda") for k in ("plain", "split")}
    work = {k: torch.empty(ops.num_owned_tiles(size, size, layout), dtype=torch.int32, device="cuda") for k in ("plain", "split")}
    run = lambda key, split, ws=None: ops.blend_forward(   # noqa: E731
        st["start"], st["end"], st["payload"], st["attrs"], size, size, layout, ordered=True,
        tile_work=work[key] if state else None, boundary=boundary.get(key), split=split, ws=ws, **common)
    plain = run("plain", False)
    ws = ops.Workspaces()
    first = [None if t is None else t.clone() for t in run("split", "force", ws)]
    again = run("split", "force", ws)
    for a, b in zip(first, again):                                               # bitwise reproducible
        assert (a is None and b is None) or torch.equal(a, b)
    image, depth, acc_alpha, last_eff, count, hits = first
    assert torch.equal(hits, plain[5])                                           # the same pairs on every pixel
    stopped = int((plain[2] > 1 - 1.05e-4).sum()) if state else -1
    if not rgb_only:
        assert torch.equal(count, plain[4])
        ok = count > 0
        d = (depth - plain[1]).abs()[ok] / plain[1][ok].abs().clamp_min(1e-6)
        assert float(d.max()) < 1e-5
    worst = float((image - plain[0]).abs().max())
    report("split_forward", size=size, n=n, tiles=(size // 16) ** 2, image_linf=worst, pixels_at_the_stop=stopped)
    assert worst < 2e-6
    if state:
        assert torch.equal(last_eff, plain[3])
        assert torch.equal(work["split"], work["plain"])
        assert float((acc_alpha - plain[2]).abs().max()) < 1e-6
        if n == 30_000 and size == 192:
            assert stopped > 1000   # the case is there for the pixels that stop
        # the split backward pass from the split forward's boundary states, against the un-split backward of the un-split forward
        g = make_grad_image(size, size).cuda()
        base = ops.blend_backward_partials(st["start"], st["payload"], st["attrs"], g, plain[2], plain[3], st["slot_offsets"],
                                           st["n_slots"], size, size, layout, tile_work=work["plain"], debug_hits=True)
        p0, f0, m0, d0 = [t.clone() for t in base]
        p1, f1, m1, d1 = ops.blend_backward_partials(st["start"], st["payload"], st["attrs"], g, acc_alpha, last_eff,
                                                     st["slot_offsets"], st["n_slots"], size, size, layout, tile_work=work["split"],
                                                     debug_hits=True, ws=ws, image=image, boundary=boundary["split"])
        assert torch.equal(d1, d0) and torch.equal(f1, f0)
        a0 = ops.reduce_partials(st["slot_offsets"], st["ntiles"], f0, p0)
        a1 = ops.reduce_partials(st["slot_offsets"], st["ntiles"], f1, p1)
        scale = a0[:, :10].abs().amax(dim=0).clamp_min(1e-30)
        worst_b = float(((a1[:, :10] - a0[:, :10]).abs() / scale).max())
        rel = float((a1[:, :10] - a0[:, :10]).norm() / a0[:, :10].norm())
        report("split_forward.backward", size=size, max_scaled_difference_of_sums=worst_b, rel_l2_of_sums=rel)
        # (looser than test_split_backward_on_small_grids' 5e-6 / 2e-6: a segment's transmittance starts from the product of
        #  the probes in front of it, which is the chain's own value to ~sqrt(hits) roundings only -- one common factor
        #  1 +- ~5e-7 on everything behind a cut, where the un-split chain's roundings are independent per entry)
        assert worst_b < 5e-5 and rel < 5e-6


@pytest.mark.parametrize("rows", [dict(row_begin=1, row_step=2), dict(row_begin=3, row_step=1, row_end=11)])
def test_split_forward_on_owned_tile_rows(ops, rows):
    """The forward list split on a rank's share of a frame (every other tile row; a band of eight rows): the owned rows come
    out as the un-split pass renders them -- same pairs, counts and last effective positions, image to rounding -- and the
    rows of other ranks are not touched."""
    from taichi_3d_gaussian_splatting_amd.synthetic import make_scene
    size = 256
    s = make_scene(n=10_000, height=size, width=size, s_min=0.01, s_max=0.08, seed=77).to("cuda")
    layout = ops.ListLayout(bin_shift=0, **rows)
    st = _stages_to_ranges(ops, s, layout)
    plain = ops.blend_forward(st["start"], st["end"], st["payload"], st["attrs"], size, size, layout, debug_hits=True)
    split = ops.blend_forward(st["start"], st["end"], st["payload"], st["attrs"], size, size, layout, debug_hits=True,
                              split="force", ws=ops.Workspaces())
    owned = torch.zeros(size, dtype=torch.bool, device="cuda")
    for r in layout.owned_rows(size):
        owned[16 * r:16 * r + 16] = True
    assert torch.equal(split[5], plain[5]) and torch.equal(split[4], plain[4]) and torch.equal(split[3][owned], plain[3][owned])
    assert float((split[0] - plain[0]).abs().max()) < 2e-6 and float((split[2] - plain[2])[owned].abs().max()) < 1e-6
    assert not split[0][~owned].any() and not split[4][~owned].any()      # (sharded outputs start as zeros)
    assert int(split[4][owned].sum()) > 0


def test_ordered_dispatch_on_a_4k_grid(ops):
    """Longest-first dispatch on a grid of 32,400 tiles (3840 x 2160: more tiles than the ordering kernel keeps in
    registers, and not a multiple of its eight lists' length): every tile is still rendered exactly once -- outputs
    pre-filled with NaN come back identical to the image-order launch -- in both passes."""
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image, make_scene
    s = make_scene(n=200_000, height=2160, width=3840, s_min=0.004, s_max=0.03, seed=13).to("cuda")
    st = _stages_to_ranges(ops, s, ops.ListLayout(bin_shift=1))
    args = (st["start"], st["end"], st["payload"], st["attrs"], s.width, s.height, st["layout"])
    plain = ops.blend_forward(*args)
    nan_f = lambda *shape: torch.full(shape, float("nan"), device="cuda")   # noqa: E731
    out = (nan_f(s.height, s.width, 3), nan_f(s.height, s.width), nan_f(s.height, s.width),
           torch.full((s.height, s.width), -7, dtype=torch.int32, device="cuda"),
           torch.full((s.height, s.width), -7, dtype=torch.int32, device="cuda"))
    work = torch.full((ops.num_owned_tiles(s.width, s.height, st["layout"]),), -1, dtype=torch.int32, device="cuda")
    ordered = ops.blend_forward(*args, out=out, ordered=True, tile_work=work)
    for a, b in zip(plain, ordered):
        assert torch.equal(a, b)
    assert int(work.min()) >= 0
    g = make_grad_image(s.height, s.width).cuda()
    bargs = (st["start"], st["payload"], st["attrs"], g, plain[2], plain[3], st["slot_offsets"], st["n_slots"], s.width,
             s.height, st["layout"])
    p0, f0, m0 = ops.blend_backward_partials(*bargs)
    p1, f1, m1 = ops.blend_backward_partials(*bargs, tile_work=work)
    assert torch.equal(f0, f1) and torch.equal(m0, m1) and torch.equal(p0[f0.bool()].view(torch.int32), p1[f1.bool()].view(torch.int32))


def test_operator_options_that_must_not_change_a_bit(scene):
    """`fused_slot_reduction` (slot sums inside the per-point kernel), `always_store_normalised_rotation` (the write-back a
    training iteration pays, for benchmarks of static scenes) and `ordered_dispatch` off: same image, same gradients, same
    in-place normalised quaternions, bit for bit."""
    from taichi_3d_gaussian_splatting_amd import GaussianPointCloudRasterisation as Op
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image
    g = make_grad_image(scene.height, scene.width)
    cfg = Op.GaussianPointCloudRasterisationConfig(near_plane=scene.near_plane, far_plane=scene.far_plane,
                                                   depth_to_sort_key_scale=scene.depth_to_sort_key_scale)
    ref = _run_operator(scene, g, op=Op(cfg))
    for attr, value in (("fused_slot_reduction", True), ("always_store_normalised_rotation", True),
                        ("ordered_dispatch", False)):
        op = Op(cfg)
        setattr(op, attr, value)
        got = _run_operator(scene, g, op=op)
        for i in range(3):
            assert torch.equal(got[i], ref[i]), (attr, i)
        assert torch.equal(got[3].grad.view(torch.int32), ref[3].grad.view(torch.int32)), attr
        assert torch.equal(got[4].grad.view(torch.int32), ref[4].grad.view(torch.int32)), attr
        assert torch.equal(got[4].detach(), ref[4].detach()), attr     # the caller's features after the in-place normalisation


@pytest.mark.parametrize("bin_shift,no_grad,rgb_only", [(None, False, False), (0, False, False), (1, False, False),
                                                         (2, False, False), (1, True, False), (0, True, True)])
def test_one_entry_point_per_pass_is_bit_identical(scene, bin_shift, no_grad, rgb_only):
    """Speculative frames through gs_frame_forward / gs_frame_backward (one foreign call per pass, frame_path.py) against
    the same frames issued stage by stage: the same kernels with the same arguments -- image, depth, count, dense gradients,
    the in-place normalised quaternions and every hook field equal bit for bit, frame after frame (the first frame of an
    operator always runs stage by stage: there is nothing to speculate from)."""
    from taichi_3d_gaussian_splatting_amd import GaussianPointCloudRasterisation as Op
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image
    g = None if no_grad else make_grad_image(scene.height, scene.width)
    cfg = Op.GaussianPointCloudRasterisationConfig(near_plane=scene.near_plane, far_plane=scene.far_plane,
                                                   depth_to_sort_key_scale=scene.depth_to_sort_key_scale, rgb_only=rgb_only)
    runs = {}
    for frames in (True, False):
        hooks = []
        op = Op(cfg, backward_valid_point_hook=hooks.append)
        op.frame_entry_points = frames
        op.bin_shift = bin_shift
        outs = []
        for _ in range(3):
            if no_grad:
                with torch.no_grad():
                    outs.append(_run_operator(scene, None, op=op))
            else:
                outs.append(_run_operator(scene, g, op=op))
        runs[frames] = (outs, hooks, dict(op.speculation_stats))
    assert runs[True][2] == runs[False][2] and runs[True][2]["frames"] == 3 and runs[True][2]["redone"] == 0
    for a, b in zip(runs[True][0], runs[False][0]):
        for i in range(3):
            assert torch.equal(a[i], b[i]), i
        assert torch.equal(a[4].detach(), b[4].detach())
        if not no_grad:
            assert torch.equal(a[3].grad.view(torch.int32), b[3].grad.view(torch.int32))
            assert torch.equal(a[4].grad.view(torch.int32), b[4].grad.view(torch.int32))
    for ha, hb in zip(runs[True][1], runs[False][1]):
        for name in ("point_id_in_camera_list", "grad_point_in_camera", "grad_pointfeatures_in_camera", "grad_viewspace",
                     "magnitude_grad_viewspace", "magnitude_grad_viewspace_on_image", "num_overlap_tiles",
                     "num_affected_pixels", "point_depth", "point_uv_in_camera"):
            x, y = getattr(ha, name), getattr(hb, name)
            assert x.shape == y.shape and x.dtype == y.dtype, name
            assert torch.equal(x.view(torch.int32) if x.dtype == torch.float32 else x,
                               y.view(torch.int32) if y.dtype == torch.float32 else y), name
    assert len(runs[True][1]) == (0 if no_grad else 3)


def test_colours_on_their_own_are_bit_identical(ops, scene, ofwd):
    """gs_preprocess_geometry + gs_view_colours (the colour half of the projection as a kernel of its own, which
    gs_frame_forward runs on a second stream beside the list stages) leave the same records, bit for bit, as gs_preprocess
    -- also launched on a stream of its own."""
    s = scene
    layout = ops.ListLayout(bin_shift=0, exact_cull=True)
    args = (dev(s.point_cloud), None, dev(s.point_object_id), dev(s.camera_intrinsics), dev(ofwd["q_cp"]),
            dev(ofwd["t_cp"]), dev(ofwd["ids"]), s.width, s.height, layout)

    def run(colours):
        feat = dev(s.point_cloud_features).clone()
        a = list(args)
        a[1] = feat
        return feat, ops.preprocess(*a, colours=colours)

    feat0, (attrs0, ntiles0, nkeys0, _, _) = run(True)
    feat1, (attrs1, ntiles1, nkeys1, _, _) = run(False)
    emits = nkeys0 > 0
    assert torch.equal(nkeys0, nkeys1) and torch.equal(ntiles0, ntiles1) and torch.equal(feat0, feat1)
    assert bool(emits.any()) and torch.equal(attrs1[emits][:, 8:11], torch.zeros_like(attrs1[emits][:, 8:11]))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    ops.view_colours(args[0], feat1, args[2], args[4], args[5], args[6], nkeys1, attrs1, stream=side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    assert torch.equal(attrs0[:, 0:4].view(torch.int32), attrs1[:, 0:4].view(torch.int32))
    assert torch.equal(attrs0[emits].view(torch.int32), attrs1[emits].view(torch.int32))
    assert float(attrs0[emits][:, 8:11].min()) > 0.0   # sigmoid outputs: the colours really are there


def test_colours_beside_the_list_stages_do_not_change_a_bit(scene):
    """The operator with the colours evaluated on its second stream beside key generation, sort and ranges
    (`colours_beside_list_stages`, the default of the per-pass entry points) against the colours inside gs_preprocess:
    image, depth, count, gradients and the stored quaternions equal bit for bit over five frames."""
    from taichi_3d_gaussian_splatting_amd import GaussianPointCloudRasterisation as Op
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image
    g = make_grad_image(scene.height, scene.width)
    cfg = Op.GaussianPointCloudRasterisationConfig(near_plane=scene.near_plane, far_plane=scene.far_plane,
                                                   depth_to_sort_key_scale=scene.depth_to_sort_key_scale)
    runs = {}
    for beside in (True, False):
        op = Op(cfg)
        op.colours_beside_list_stages = beside
        runs[beside] = [_run_operator(scene, g, op=op) for _ in range(5)]
        assert op.speculation_stats == {"frames": 5, "redone": 0}
        assert bool(op._aux_streams) == beside
    for a, b in zip(runs[True], runs[False]):
        for i in range(3):
            assert torch.equal(a[i], b[i]), i
        assert torch.equal(a[3].grad.view(torch.int32), b[3].grad.view(torch.int32))
        assert torch.equal(a[4].grad.view(torch.int32), b[4].grad.view(torch.int32))
        assert torch.equal(a[4].detach(), b[4].detach())


def test_one_entry_point_per_pass_overflow_falls_back(scene):
    """A frame that outgrows the capacities learnt from the previous one (four times the Gaussians on screen) is redone by
    the stage-by-stage path with exact sizes: the same outputs as a fresh operator."""
    from taichi_3d_gaussian_splatting_amd import GaussianPointCloudRasterisation as Op
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image, make_scene
    g = make_grad_image(256, 256)
    small = make_scene(n=2000, height=256, width=256, s_min=0.01, s_max=0.05, seed=1)
    big = make_scene(n=20000, height=256, width=256, s_min=0.01, s_max=0.08, seed=2)
    op = Op(Op.GaussianPointCloudRasterisationConfig())
    op.bin_shift = 0
    _run_operator(small, g, op=op)
    _run_operator(small, g, op=op)                     # through the frame entry points
    got = _run_operator(big, g, op=op)                 # does not fit: redone
    assert op.speculation_stats["redone"] == 1 and op.speculation_stats["frames"] == 3
    fresh = Op(Op.GaussianPointCloudRasterisationConfig())
    fresh.bin_shift = 0
    ref = _run_operator(big, g, op=fresh)
    for i in range(3):
        assert torch.equal(got[i], ref[i])
    assert torch.equal(got[4].grad.view(torch.int32), ref[4].grad.view(torch.int32))
    again = _run_operator(big, g, op=op)               # and the next frame speculates again, through the entry points
    assert torch.equal(again[0], ref[0]) and op.speculation_stats["redone"] == 1


def test_backward_on_walked_lists_is_bit_identical(ops):
    """Binned layouts: the forward pass writes out every tile's own list as far as it walks it and the backward pass runs on
    those per-tile lists.  Same entries in the same order through the same arithmetic: gradients, hook fields and the
    magnitude image equal the ones of the backward pass that filters the bin lists again, bit for bit, for 2x2- and 4x4-tile
    bins (8,040 tiles: both run the two-wave kernels; on grids of at most 3,840 tiles the walked lists let the backward pass
    take the four-wave form, which adds the same per-pixel terms in another order); and the emitted lists are the tiles'
    lists of the per-tile layout, truncated where the forward pass stopped."""
    from taichi_3d_gaussian_splatting_amd import GaussianPointCloudRasterisation as Op
    from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene, make_grad_image
    s = make_config_scene("cfg3_400k_1080p")
    g = make_grad_image(s.height, s.width)
    cfg = Op.GaussianPointCloudRasterisationConfig(near_plane=s.near_plane, far_plane=s.far_plane,
                                                   depth_to_sort_key_scale=s.depth_to_sort_key_scale)
    for bin_shift in (1, 2):
        res = {}
        for walked in (True, False):
            got = {}
            op = Op(cfg, backward_valid_point_hook=lambda h: got.__setitem__("h", h))
            op.bin_shift, op.backward_on_walked_lists = bin_shift, walked
            out = _run_operator(s, g, op=op)
            res[walked] = (out, got["h"])
        (a, ha), (b, hb) = res[True], res[False]
        for i in range(3):
            assert torch.equal(a[i], b[i])
        assert torch.equal(a[3].grad.view(torch.int32), b[3].grad.view(torch.int32)), bin_shift
        assert torch.equal(a[4].grad.view(torch.int32), b[4].grad.view(torch.int32)), bin_shift
        assert torch.equal(ha.magnitude_grad_viewspace_on_image, hb.magnitude_grad_viewspace_on_image)
        assert torch.equal(ha.num_affected_pixels, hb.num_affected_pixels)
    # the emitted lists themselves, against the per-tile layout's sorted lists
    d = s.to("cuda")
    tile = _stages_to_ranges(ops, d, ops.ListLayout(bin_shift=0))
    binned = _stages_to_ranges(ops, d, ops.ListLayout(bin_shift=1))
    ft = ops.blend_forward(tile["start"], tile["end"], tile["payload"], tile["attrs"], d.width, d.height, tile["layout"],
                           arm="two_waves")
    fb = ops.blend_forward(binned["start"], binned["end"], binned["payload"], binned["attrs"], d.width, d.height,
                           binned["layout"], emit_walked_lists=True)
    walked_start, walked_list = fb[5], fb[6]
    for i in (0, 1, 2, 4):
        assert torch.equal(fb[i], ft[i])
    tw = d.width // 16
    reach_t = (ft[3].view(d.height // 16, 16, tw, 16).amax(dim=(1, 3)).flatten() - tile["start"]).cpu()
    reach_b = (fb[3].view(d.height // 16, 16, tw, 16).amax(dim=(1, 3)).flatten() - walked_start).cpu()
    assert torch.equal(reach_t, reach_b)      # last blended entry at the same position of the tile's list
    ts, ws_, pl, wl = tile["start"].cpu(), walked_start.cpu(), tile["payload"].cpu(), walked_list.cpu()
    for t in torch.randperm(ts.shape[0], generator=torch.Generator().manual_seed(0))[:200].tolist():
        n = int(reach_t[t])
        assert torch.equal(pl[ts[t]:ts[t] + n], wl[ws_[t]:ws_[t] + n]), t


def test_rgb_only_and_inference_paths(ops, scene):
    """rgb_only (RAS:464-469,478-484) and the no-gradient path: the image is bit-identical to the full forward; depth and
    count are zeros under rgb_only; gradients under rgb_only equal those of the default configuration."""
    from taichi_3d_gaussian_splatting_amd import GaussianPointCloudRasterisation as Op
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image
    g = make_grad_image(scene.height, scene.width)
    full = _run_operator(scene, g)
    cfg = dict(near_plane=scene.near_plane, far_plane=scene.far_plane,
               depth_to_sort_key_scale=scene.depth_to_sort_key_scale)
    rgb = _run_operator(scene, g, op=Op(Op.GaussianPointCloudRasterisationConfig(rgb_only=True, **cfg)))
    assert torch.equal(rgb[0], full[0]) and not rgb[1].any() and not rgb[2].any()
    assert torch.equal(rgb[4].grad, full[4].grad) and torch.equal(rgb[3].grad, full[3].grad)
    with torch.no_grad():
        inf = _run_operator(scene, None)
        inf_rgb = _run_operator(scene, None, op=Op(Op.GaussianPointCloudRasterisationConfig(rgb_only=True, **cfg)))
    assert torch.equal(inf[0], full[0]) and torch.equal(inf[1], full[1]) and torch.equal(inf[2], full[2])
    assert torch.equal(inf_rgb[0], full[0]) and not inf_rgb[1].any()
    # stage level: every flag combination writes the same image
    st = _stages_to_ranges(ops, scene.to("cuda"))
    ref = ops.blend_forward(st["start"], st["end"], st["payload"], st["attrs"], scene.width, scene.height, st["layout"])
    for rgb_only in (False, True):
        for need_state in (False, True):
            out = ops.blend_forward(st["start"], st["end"], st["payload"], st["attrs"], scene.width, scene.height,
                                    st["layout"], rgb_only=rgb_only, need_state=need_state)
            assert torch.equal(out[0], ref[0])
            assert (out[1] is None) == rgb_only and (out[2] is None) == (not need_state)
            if not rgb_only:
                assert torch.equal(out[1], ref[1]) and torch.equal(out[4], ref[4])
            if need_state:
                assert torch.equal(out[2], ref[2]) and torch.equal(out[3], ref[3])


def test_speculative_sizes_overflow_is_redone(scene):
    """The list stages are launched from device-side counts with the previous frame's capacities (no blocking size
    read-back in the steady state).  A frame that does not fit -- many more keys, or a deeper depth range than the
    previous frame -- is detected when the sizes arrive and redone: its outputs equal those of a fresh operator."""
    from taichi_3d_gaussian_splatting_amd import GaussianPointCloudRasterisation as Op
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image
    g = make_grad_image(scene.height, scene.width)
    few = small_scene(n=500, size=scene.height, seed=3)           # 20x fewer keys, same image size and planes
    far = small_scene(n=3000, size=scene.height, seed=4)
    far.point_cloud[:, 2] += 40.0                                # quantised depths ~4300: more bits than the ~400 before
    cfg = Op.GaussianPointCloudRasterisationConfig()
    op = Op(cfg)
    fresh = lambda sc: _run_operator(sc, g, op=Op(cfg))           # noqa: E731
    seq = [few, few, scene, scene, far, far, few]
    expect_redone = [0, 0, 1, 1, 2, 2, 2]                         # frame 0 has nothing to speculate from
    for sc, redone in zip(seq, expect_redone):
        got, ref = _run_operator(sc, g, op=op), fresh(sc)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2])
        assert torch.equal(got[3].grad, ref[3].grad) and torch.equal(got[4].grad, ref[4].grad)
        assert op.speculation_stats["redone"] == redone, (op.speculation_stats, redone)
    assert op.speculation_stats["frames"] == len(seq)
    op.speculative_sizes = False                                  # the two-halves path still works
    got, ref = _run_operator(scene, g, op=op), fresh(scene)
    assert torch.equal(got[0], ref[0]) and torch.equal(got[4].grad, ref[4].grad)
    # frames with nothing on screen in between (every point invalid): zero keys under a non-zero capacity and back
    import copy
    nothing = copy.deepcopy(scene)
    nothing.point_invalid_mask = torch.ones_like(scene.point_invalid_mask)
    op3 = Op(cfg)
    for sc in (scene, nothing, scene, nothing, nothing, scene, scene):
        got, ref = _run_operator(sc, g, op=op3), fresh(sc)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[2], ref[2]) and torch.equal(got[4].grad, ref[4].grad)
    assert not _run_operator(nothing, g, op=op3)[0].any()
    # a data set that mixes image sizes keeps speculating: one size guess per (image size, layout, planes)
    small = small_scene(n=2000, size=128, seed=6)
    g_small = make_grad_image(small.height, small.width)
    op2 = Op(cfg)
    for sc, gi in ((scene, g), (small, g_small)) * 3:
        got, ref = _run_operator(sc, gi, op=op2), _run_operator(sc, gi, op=Op(cfg))
        assert torch.equal(got[0], ref[0]) and torch.equal(got[4].grad, ref[4].grad)
    assert len(op2._size_guesses) == 2 and op2.speculation_stats == {"frames": 6, "redone": 0}


@pytest.mark.parametrize("mix", ["ordinary", "sixteen_lanes"])
def test_slot_sums_against_a_host_reference(ops, mix):
    """gs_reduce_partials on synthetic slot layouts: every Gaussian's flagged slots summed, the others never looked at (they
    hold NaNs), for slot runs of every length around the group sizes of the flag gather (gs_slots.h: the flags of a run are
    fetched as the aligned dwords that hold them -- 1..13 slots from the first four dwords whatever the alignment, up to 32
    per group, above 128 the whole wave) starting at every byte alignment.  Integer-valued records: the sums are exact in
    any order, so the comparison is bit for bit."""
    rng = np.random.default_rng(11)
    if mix == "ordinary":
        n = np.concatenate([np.zeros(50, int), rng.integers(1, 14, 3000), rng.integers(14, 33, 600), rng.integers(33, 129, 200),
                            rng.integers(129, 700, 12), np.arange(0, 70)])
    else:   # hundreds of slots per Gaussian on average: sixteen lanes per Gaussian (reduce_partials_kernel<16>)
        n = np.concatenate([rng.integers(600, 1500, 40), np.arange(0, 9), rng.integers(1, 14, 10)])
    rng.shuffle(n)
    m, total = len(n), int(n.sum())
    if mix == "sixteen_lanes":
        assert total > 512 * m
    offsets = np.concatenate([[0], np.cumsum(n)[:-1]]).astype(np.int32)
    flags = (rng.random(total) < 0.6).astype(np.uint8)
    rows = rng.integers(-8, 9, (total, 12)).astype(np.float32)
    npix = rng.integers(0, 257, total).astype(np.int32)
    rows[:, 10] = npix.view(np.float32)
    rows[:, 11] = 0.0
    stored = rows.copy()
    stored[flags == 0] = np.nan                                   # a slot that was never written
    owner = np.repeat(np.arange(m), n)
    expect = np.zeros((m, 12), np.float64)
    np.add.at(expect, owner[flags == 1], rows[flags == 1].astype(np.float64))
    expect_npix = np.zeros(m, np.int64)
    np.add.at(expect_npix, owner[flags == 1], npix[flags == 1])
    padded = np.zeros((total + 15) & ~15, np.uint8)              # (the header's allocation rule for the flags)
    padded[:total] = flags
    acc = ops.reduce_partials(torch.from_numpy(offsets).cuda(), torch.from_numpy(n.astype(np.int32)).cuda(),
                              torch.from_numpy(padded).cuda()[:total], torch.from_numpy(stored).cuda()).cpu().numpy()
    assert np.array_equal(acc[:, :10], expect[:, :10].astype(np.float32))
    assert np.array_equal(acc[:, 10].view(np.int32), expect_npix.astype(np.int32))


def test_sizes_as_stamped_words_and_behind_an_event_are_the_same_frames(scene, monkeypatch):
    """The frame's sizes reach the host as stamped 64-bit words the host polls (default: no event behind the scan) or as
    int32 counters behind an event (GS_SIZE_STAMPS=0): same frames, same speculation history, over a sequence whose sizes
    change every frame (more keys, fewer keys, a deeper depth range, nothing on screen)."""
    import copy
    from taichi_3d_gaussian_splatting_amd import GaussianPointCloudRasterisation as Op
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image
    g = make_grad_image(scene.height, scene.width)
    few = small_scene(n=500, size=scene.height, seed=3)
    far = small_scene(n=3000, size=scene.height, seed=4)
    far.point_cloud[:, 2] += 40.0
    nothing = copy.deepcopy(scene)
    nothing.point_invalid_mask = torch.ones_like(scene.point_invalid_mask)
    seq = [few, scene, scene, nothing, far, few, scene, nothing, nothing, scene]
    cfg = Op.GaussianPointCloudRasterisationConfig()
    stamped = Op(cfg)
    monkeypatch.setenv("GS_SIZE_STAMPS", "0")
    evented = Op(cfg)
    evented._counter_readback(torch.device("cuda", torch.cuda.current_device()))   # (the transport is chosen when the read-back is made)
    monkeypatch.delenv("GS_SIZE_STAMPS")
    for sc in seq:
        a, b = _run_operator(sc, g, op=stamped), _run_operator(sc, g, op=evented)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
        assert torch.equal(a[3].grad, b[3].grad) and torch.equal(a[4].grad, b[4].grad)
    assert stamped.speculation_stats == evented.speculation_stats and stamped.speculation_stats["frames"] == len(seq)
    rb_s, rb_e = list(stamped._readbacks.values())[0], list(evented._readbacks.values())[0]
    assert rb_s._raw is not None and rb_s._stamp == len(seq) - 1   # (frame 0 has nothing to speculate from: stage by stage)
    assert rb_e._raw is None and rb_e.next_stamp() == 0


def test_speculative_depth_overflow_cannot_write_outside_the_ranges(ops):
    """a speculative frame builds 32-bit keys with the PREVIOUS frame's depth width.  When this frame's
    quantised depths need more bits, the excess used to spill into the bin field, and with a bin count that is not a power
    of two `tile_ranges` wrote up to ~600 bytes past its arrays before the host noticed the overflow and redid the frame.
    Now the depth field is masked and `tile_ranges` checks bin ids: keys of such a frame stay inside the bin range and a
    guard region behind the ranges stays untouched."""
    from taichi_3d_gaussian_splatting_amd import _lib
    s = small_scene(n=3000, size=240, seed=4)          # 15 x 15 = 225 tiles: not a power of two
    s.point_cloud[:, 2] += 40.0                        # quantised depths ~4300: 13 bits
    d = s.to("cuda")
    layout = ops.ListLayout(bin_shift=0)
    q_cp, t_cp = ops.pose_inverse(d.q_pointcloud_camera, d.t_pointcloud_camera)
    mask, ids, counters = ops.filter_compact(d.point_cloud, d.point_invalid_mask, d.point_object_id, d.camera_intrinsics,
                                             q_cp, t_cp, d.near_plane, d.far_plane, d.width, d.height)
    attrs, ntiles, nowned, bsums, bsums_full = ops.preprocess(
        d.point_cloud, d.point_cloud_features.clone(), d.point_object_id, d.camera_intrinsics, q_cp, t_cp, ids, d.width,
        d.height, layout, depth_to_sort_key_scale=d.depth_to_sort_key_scale, counters=counters)
    k, n_slots, max_dq, _ = ops.scan_block_sums(bsums, counters, bsums_full)
    num_bins = layout.num_bins(d.width, d.height)
    assert num_bins == 225 and max_dq >= 8 * 400       # an 8x deeper range than the width the keys are built with
    stale_kdb = 9                                       # what a frame with depths ~400 would have left behind
    keys, payload, _ = ops.make_keys(attrs, nowned, bsums, k, d.width, d.height, d.depth_to_sort_key_scale, layout,
                                     key_depth_bits=stale_kdb, num_overlap_tiles=ntiles, block_offsets_full=bsums_full)
    assert k > 0 and int((keys.view(torch.int32).long() & 0xffffffff).max() >> stale_kdb) < num_bins
    ops.sort_pairs(keys, payload, stale_kdb, 8, stale_kdb)
    guard = 256
    buf = torch.full((2 * num_bins + guard,), -12345, dtype=torch.int32, device="cuda")
    _lib.call("gs_tile_ranges", _lib.ptr(keys), k, None, stale_kdb, _lib.ptr(buf[:num_bins]),
              _lib.ptr(buf[num_bins:2 * num_bins]), num_bins, _lib.current_stream(buf.device))
    torch.cuda.synchronize()
    assert bool((buf[2 * num_bins:] == -12345).all()), "tile_ranges wrote behind its arrays"
    # and a hand-made key with a bin id outside the range is ignored, not written
    bad = torch.tensor([(3 << stale_kdb) | 5, (300 << stale_kdb) | 1, (301 << stale_kdb) | 1], dtype=torch.int32, device="cuda")
    buf.fill_(-12345)
    _lib.call("gs_tile_ranges", _lib.ptr(bad), 3, None, stale_kdb, _lib.ptr(buf[:num_bins]),
              _lib.ptr(buf[num_bins:2 * num_bins]), num_bins, _lib.current_stream(buf.device))
    torch.cuda.synchronize()
    assert bool((buf[2 * num_bins:] == -12345).all()) and int(buf[num_bins + 3]) == 1


def test_hook_feature_gradients_can_be_switched_off(scene):
    from taichi_3d_gaussian_splatting_amd import GaussianPointCloudRasterisation as Op
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image
    g = make_grad_image(scene.height, scene.width)
    got = {}
    full = _run_operator(scene, g, hook=lambda h: got.setdefault("full", h))
    op = Op(Op.GaussianPointCloudRasterisationConfig(near_plane=scene.near_plane, far_plane=scene.far_plane,
                                                     depth_to_sort_key_scale=scene.depth_to_sort_key_scale),
            backward_valid_point_hook=lambda h: got.setdefault("lean", h))
    op.hook_feature_gradients = False
    lean = _run_operator(scene, g, op=op)
    assert got["lean"].grad_pointfeatures_in_camera is None and got["full"].grad_pointfeatures_in_camera is not None
    for name in ("point_id_in_camera_list", "grad_point_in_camera", "grad_viewspace", "magnitude_grad_viewspace",
                 "num_overlap_tiles", "num_affected_pixels", "point_depth", "point_uv_in_camera"):
        assert torch.equal(getattr(got["lean"], name), getattr(got["full"], name)), name
    assert torch.equal(full[4].grad, lean[4].grad) and torch.equal(full[3].grad, lean[3].grad)


@pytest.mark.parametrize("bin_shift", [0, 1, 2])
@pytest.mark.parametrize("split", ["interleaved", "bands"])
def test_operator_tile_row_sharding_matches_single(scene, split, bin_shift):
    """Image-space sharding: rendering two sets of tile rows separately -- rows {0,2,4,..} / {1,3,5,..}, or the bands
    [0,7) / [7,16) whose boundary cuts through the 2x2- and 4x4-tile bins -- and merging equals the un-sharded render
    bit-for-bit, with per-tile keys and with binned lists; partial gradients add up.  (Bit-for-bit between forward passes
    that do not split their lists: the split form -- the default on grids this small with per-tile keys -- takes the same
    decisions and agrees to rounding, which the last lines check.)"""
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image
    g = make_grad_image(scene.height, scene.width)
    image, depth, count, xyz, feat = _run_operator(scene, g, bin_shift=0, split_forward=False)
    rows = torch.arange(scene.height, device="cuda") // 16
    if split == "interleaved":
        parts = [_run_operator(scene, g, row=(r, 2), bin_shift=bin_shift, split_forward=False) for r in range(2)]
        first = rows % 2 == 0
    else:
        parts = [_run_operator(scene, g, row=(0, 1), row_end=7, bin_shift=bin_shift, split_forward=False),
                 _run_operator(scene, g, row=(7, 1), bin_shift=bin_shift, split_forward=False)]
        first = rows < 7
    merged = torch.where(first[:, None, None], parts[0][0], parts[1][0])
    assert torch.equal(merged, image)
    merged_count = torch.where(first[:, None], parts[0][2], parts[1][2])
    assert torch.equal(merged_count, count)
    if bin_shift == 0:
        image_s, depth_s, count_s, xyz_s, feat_s = _run_operator(scene, g, bin_shift=0, split_forward=True)
        assert torch.equal(count_s, count) and float((image_s - image).abs().max()) < 2e-6
        assert rel_l2(feat_s.grad.cpu().numpy(), feat.grad.cpu().numpy()) < 1e-5
    gsum = parts[0][4].grad + parts[1][4].grad
    assert rel_l2(gsum.cpu().numpy(), feat.grad.cpu().numpy()) < 1e-4
    gx = parts[0][3].grad + parts[1][3].grad
    assert rel_l2(gx.cpu().numpy(), xyz.grad.cpu().numpy()) < 1e-4


def test_operator_edge_cases():
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image
    # (a) every point invalid -> M = 0: zero image, zero grads, no crash (RAS:915-918,934,959,980)
    s = small_scene(n=300, size=64, seed=1, invalid_fraction=1.1)
    image, depth, count, xyz, feat = _run_operator(s, make_grad_image(64, 64))
    assert not image.any() and not count.any() and not xyz.grad.any() and not feat.grad.any()
    # (b) a single Gaussian
    s1 = small_scene(n=1, size=64, seed=2)
    s1.point_cloud[:] = torch.tensor([[0.05, -0.02, 0.0]])
    f = oracle_forward(s1)
    image, *_ = _run_operator(s1, None)
    assert np.abs(image.detach().cpu().numpy() - f["image"])[f["margin"] >= FRAGILE_MARGIN].max() <= PIXEL_TOL
    assert image.max() > 0.01
    # (c) ragged sizes: N not a multiple of any block size, non-square image
    from taichi_3d_gaussian_splatting_amd.synthetic import make_scene
    s2 = make_scene(n=1237, height=48, width=112, s_min=0.02, s_max=0.1, seed=3)
    f2 = oracle_forward(s2)
    image, depth, count, *_ = _run_operator(s2, None)
    assert np.abs(image.detach().cpu().numpy() - f2["image"])[f2["margin"] >= FRAGILE_MARGIN].max() <= PIXEL_TOL


def test_operator_multi_object_poses():
    """Kobj = 2: per-point object ids select the camera pose (RAS:56-59,272-275,727-732)."""
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image
    s = small_scene(n=4000, size=128, seed=5)
    s.point_object_id = (torch.arange(4000) % 2).to(torch.int32)
    q2 = torch.tensor([[0.0, 0.0, 0.0, 1.0], [0.02, -0.03, 0.01, 1.0]])
    s.q_pointcloud_camera = q2 / q2.norm(dim=1, keepdim=True)
    s.t_pointcloud_camera = torch.tensor([[0.0, 0.0, -3.0], [0.1, -0.05, -3.2]])
    f = oracle_forward(s)
    g = make_grad_image(128, 128)
    ob = O.backward(f, g.numpy(), 3)
    image, depth, count, xyz, feat = _run_operator(s, g)
    _check_image("multi_object.image", image.detach().cpu().numpy(), f["image"], f["margin"] < FRAGILE_MARGIN)
    _check_acc("multi_object.grad_feat", feat.grad.cpu().numpy(), ob["grad_feat"], FLIP_GRAD_TOL)
    _check_acc("multi_object.grad_xyz", xyz.grad.cpu().numpy(), ob["grad_xyz"], FLIP_GRAD_TOL)


def test_operator_cfg1_as_stated():
    """BASELINE config 1 exactly as stated: 10k random Gaussians, 256 x 256, SH-degree-0 DATA (only the DC coefficients are
    non-zero, `synthetic.CONFIGS["cfg1_10k_256"]`) rendered with colour band 0 -- forward and backward against the oracle.
    (The module's fixture is the same size with degree-3 data; not a different code path, SURVEY 0.7, but the config as
    written deserves its own line.)"""
    from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene
    s = make_config_scene("cfg1_10k_256")
    assert float(s.point_cloud_features[:, 9:24].abs().max()) == 0.0      # degree-0 data
    f = oracle_forward(s)
    report("cfg1.sizes", M=len(f["ids"]), K=len(f["keys"]))
    _operator_vs_oracle("cfg1", s, f, band=0)


def test_operator_cfg2_size_forward_backward():
    """BASELINE config 2: 1e5 Gaussians, 800x800, SH degree 3, forward + backward vs the oracle."""
    from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene, make_grad_image
    s = make_config_scene("cfg2_100k_800")
    f = oracle_forward(s)
    report("cfg2.sizes", M=len(f["ids"]), K=len(f["keys"]))
    _operator_vs_oracle("cfg2", s, f)


def test_headline_size_properties(ops):
    """Full size (1e6 Gaussians @1920x1072): size-independent properties instead of the oracle."""
    from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene
    s = make_config_scene("headline_1m_1080p").to("cuda")
    q_cp, t_cp = ops.pose_inverse(s.q_pointcloud_camera, s.t_pointcloud_camera)
    mask, ids, counters = ops.filter_compact(s.point_cloud, s.point_invalid_mask, s.point_object_id,
                                             s.camera_intrinsics, q_cp, t_cp, s.near_plane, s.far_plane, s.width,
                                             s.height)
    assert torch.equal(ids.long(), torch.nonzero(mask).flatten())  # ordered compaction
    feat = s.point_cloud_features.clone()
    layout = ops.ListLayout()
    attrs, ntiles, nowned, block_sums, block_sums_full = ops.preprocess(
        s.point_cloud, feat, s.point_object_id, s.camera_intrinsics, q_cp, t_cp, ids, s.width, s.height, layout,
        depth_to_sort_key_scale=s.depth_to_sort_key_scale, counters=counters)
    max_dq = ops.read_counters(counters)[ops.COUNTER_MAX_DEPTH_KEY]
    assert max_dq == int((attrs[:, 2] * s.depth_to_sort_key_scale).int().max().item())
    assert torch.allclose(feat[ids.long(), :4].norm(dim=1), torch.ones(ids.shape[0], device="cuda"), atol=1e-6)
    assert (nowned <= ntiles).all()
    total = int(nowned.sum().item())
    k, n_slots, _, m_dev = ops.scan_block_sums(block_sums, counters, block_sums_full)
    assert m_dev == ids.shape[0]
    assert k == total and n_slots == int(ntiles.sum().item())
    num_tiles = layout.num_bins(s.width, s.height)   # per-tile lists (default layout)
    assert num_tiles == 120 * 67
    kdb, db, tb = ops.key_layout(s.near_plane, s.far_plane, s.depth_to_sort_key_scale, num_tiles, max_dq)
    assert 0 < kdb < 17  # production layout at this size: compressed keys, depth field sized to the bits in use
    keys, payload, slot_offsets = ops.make_keys(attrs, nowned, block_sums, k, s.width, s.height,
                                                s.depth_to_sort_key_scale, layout, key_depth_bits=kdb,
                                                num_overlap_tiles=ntiles, block_offsets_full=block_sums_full)
    assert torch.equal(slot_offsets.long(), torch.cumsum(ntiles.long(), 0) - ntiles.long())
    # histogram of payload = key counts (every point emits exactly its count)
    assert torch.equal(torch.bincount(payload.long(), minlength=ids.shape[0]).int(), nowned)
    k0, p0 = keys.clone(), payload.clone()
    ops.sort_pairs(keys, payload, db, tb, kdb)
    ref_keys, perm = torch.sort(k0.long() & 0xffffffff, stable=True)
    assert torch.equal(keys.long() & 0xffffffff, ref_keys) and torch.equal(payload, p0[perm])  # sorted + stable
    start, end = ops.tile_ranges(keys, num_tiles, kdb)
    tile_of = ((keys.long() & 0xffffffff) >> kdb).int()
    cnt = torch.bincount(tile_of.long(), minlength=num_tiles).int()
    assert torch.equal(end - start, cnt)
    image, depth, acc_alpha, last_eff, count = ops.blend_forward(start, end, payload, attrs, s.width, s.height, layout)
    assert torch.isfinite(image).all() and image.min() >= 0 and image.max() <= 1.0 + 1e-5
    assert acc_alpha.min() >= 0 and acc_alpha.max() <= 1.0 - 1e-4 + 1e-6  # T never drops below 1e-4
    bins_v = torch.arange(s.height, device="cuda") // 16
    bins_u = torch.arange(s.width, device="cuda") // 16
    tid = bins_v[:, None] * 120 + bins_u[None, :]
    assert (last_eff >= start[tid]).all() and (last_eff <= end[tid]).all()
    assert (count <= last_eff - start[tid]).all()
    report("headline.sizes", M=ids.shape[0], K_reference=int(ntiles.sum().item()), K_after_cull=k)


def test_rccl_collectives_on_device_world1():
    """The collectives of the tile-row sharded path issued through RCCL (backend "nccl") on HIP tensors.
    World size 1 (one GPU per gpurun box): exercises process-group init, dtypes and call shapes of
    all_gather_into_tensor / all_reduce on the device; the multi-rank logic is covered by the gloo tests."""
    import os
    import socket
    import torch.distributed as dist
    from taichi_3d_gaussian_splatting_amd import distributed as D
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        h, w = 80, 64
        image = torch.rand(h, w, 3, device="cuda"); depth = torch.rand(h, w, device="cuda")
        count = torch.randint(0, 99, (h, w), device="cuda", dtype=torch.int32)
        ref = [t.clone() for t in (image, depth, count)]
        D.all_gather_tile_rows([image, depth, count], 0, 1, force=True)
        assert all(torch.equal(a, b) for a, b in zip(ref, (image, depth, count)))
        acc = torch.rand(1000, 12, device="cuda")
        acc[:, 10] = torch.randint(0, 5000, (1000,), device="cuda", dtype=torch.int32).view(torch.float32)
        ref_acc = acc.clone()
        D.all_reduce_accumulators(acc)
        assert torch.equal(acc.view(torch.int32), ref_acc.view(torch.int32))
        # the sparse exchange through RCCL with the device stages (compact -> all-gather -> merge in rank order)
        nk = (torch.rand(1000, device="cuda") < 0.3).to(torch.int32)
        masked = ref_acc * (nk > 0)[:, None]
        masked[:, 10] = (ref_acc[:, 10].contiguous().view(torch.int32) * nk).view(torch.float32)
        stats = {}
        got = D.exchange_accumulators_sparse(masked.clone(), nk, stats=stats)
        assert torch.equal(got[:, :10], masked[:, :10]) and not got[:, 11].any()
        assert torch.equal(got[:, 10].contiguous().view(torch.int32), masked[:, 10].contiguous().view(torch.int32))
        assert stats["rows_sent"] == int(nk.sum())
        # the sharded operator end to end under a (trivial) process group
        s = small_scene(n=2000, size=128, seed=9)
        g = make_grad_image(128, 128)
        base = _run_operator(s, g)
        from taichi_3d_gaussian_splatting_amd import GaussianPointCloudRasterisation as Op
        op = D.shard_rasteriser_across_tile_rows(Op(Op.GaussianPointCloudRasterisationConfig()), force=True)
        assert op.image_gather is not None and op.grad_accumulator_reduce is not None
        sharded = _run_operator(s, g, op=op)
        assert torch.equal(base[0], sharded[0]) and torch.equal(base[4].grad, sharded[4].grad)
    finally:
        dist.destroy_process_group()


def test_operator_cfg3_truck_like_forward_backward():
    """BASELINE config 3 stand-in (the Truck scene is not in the container): 4e5 Gaussians, 1920x1072, near 0.4,
    far 2000, depth_to_sort_key_scale 10 (config/tat_truck_every_8_test.yaml:44-47) -> ~96 % of sorted entries
    share their quantised depth with a neighbour, so this is the stress test of the stable tie order."""
    from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene, make_grad_image
    s = make_config_scene("cfg3_400k_1080p")
    f = oracle_forward(s)
    keys = f["keys"]
    ties = float(np.mean(keys[1:] == keys[:-1]))
    report("cfg3.sizes", M=len(f["ids"]), K=len(keys), tie_fraction=ties)
    assert ties > 0.5
    _operator_vs_oracle("cfg3", s, f)


@pytest.mark.parametrize("workload,tag", [("headline_1m_1080p", "headline"), ("cfg4_2m_1080p", "cfg4")])
def test_operator_full_size_forward_backward(workload, tag):
    """The headline workload (1e6 Gaussians) and BASELINE config 4 (2e6), 1920x1072, on one GPU against the oracle at
    full size: image on non-fragile pixels, depth-independent counts, dense gradients."""
    from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene, make_grad_image
    s = make_config_scene(workload)
    f = oracle_forward(s)
    report(f"{tag}.full_size", M=len(f["ids"]), K=len(f["keys"]))
    _operator_vs_oracle(tag, s, f)
    # the comparisons above use a fresh operator per call: first frames, i.e. sizes read before the lists are built and
    # per-tile keys.  A long-lived operator at this size switches to 2x2-tile bins after its first frame and launches the
    # list stages speculatively from its third: same bits, whatever the frame's history
    from taichi_3d_gaussian_splatting_amd import GaussianPointCloudRasterisation as Op
    g = make_grad_image(s.height, s.width)
    op = Op(Op.GaussianPointCloudRasterisationConfig(near_plane=s.near_plane, far_plane=s.far_plane,
                                                     depth_to_sort_key_scale=s.depth_to_sort_key_scale))
    frames = [_run_operator(s, g, op=op) for _ in range(4)]
    layouts = [op._auto_bin_shift]   # (after the last frame)
    assert op.list_layout(s.height, s.width).bin_shift == 1 and op.speculation_stats == {"frames": 4, "redone": 0}
    for image, depth, count, xyz, feat in frames[1:]:
        assert torch.equal(image, frames[0][0]) and torch.equal(depth, frames[0][1]) and torch.equal(count, frames[0][2])
        assert torch.equal(feat.grad, frames[0][4].grad) and torch.equal(xyz.grad, frames[0][3].grad)


def test_reference_stress_distribution_against_the_oracle():
    """VERDICT r2 missing #3 / weak #2: the reference's stress distribution (T_RAS:111-150: everything U[0,1), every
    Gaussian over every tile) at a size the CPU oracle can check -- 3000 valid rows at 960 x 544, 2040 tiles, 5.6e6
    (tile, Gaussian) pairs.  The operator's own choices on such a frame (4 x 4-tile bins, the sixteen-lanes-per-Gaussian
    slot reduction) are in force in the second frame; image, gradients and hook fields at the standard bars."""
    from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Op
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image, make_reference_stress_scene
    s = make_reference_stress_scene(seed=3, n=6000, n_valid=3000, height=544, width=960)
    f = oracle_forward(s)
    m = len(f["ids"])
    assert f["num_overlap_tiles"].sum() > 512 * m          # -> reduce_partials_kernel<16>
    fragile = f["margin"] < FRAGILE_MARGIN
    g = make_grad_image(s.height, s.width)
    g_masked = g * torch.from_numpy(~fragile)[:, :, None]
    got = {}
    op = Op(Op.GaussianPointCloudRasterisationConfig(), backward_valid_point_hook=lambda h: got.__setitem__("h", h))
    _run_operator(s, g_masked, op=op)                       # first frame: per-tile lists; teaches the operator the sizes
    for kind, gi, tol in (("masked", g_masked, MASKED_GRAD_TOL), ("all_pixels", g, FLIP_GRAD_TOL)):
        ob = O.backward(f, gi.numpy(), 3)
        image, depth, count, xyz, feat = _run_operator(s, gi, op=op)
        assert op.list_layout(s.height, s.width).bin_shift == 2, "the operator did not switch to 64-pixel bins"
        if kind == "masked":
            _check_image("stress_small.image", image.detach().cpu().numpy(), f["image"], fragile)
            ok = ~fragile
            assert np.array_equal(count.cpu().numpy()[ok], f["count"][ok])
            assert np.allclose(depth.detach().cpu().numpy()[ok], f["depth"][ok], rtol=1e-4, atol=1e-4)
        _check_acc(f"stress_small.{kind}.grad_feat", feat.grad.cpu().numpy(), ob["grad_feat"], tol)
        _check_acc(f"stress_small.{kind}.grad_xyz", xyz.grad.cpu().numpy(), ob["grad_xyz"], tol)
        h, ho = got["h"], ob["hook"]
        assert np.array_equal(h.point_id_in_camera_list.cpu().numpy(), ho["point_id_in_camera_list"])
        assert np.array_equal(h.num_overlap_tiles.cpu().numpy(), ho["num_overlap_tiles"])
        assert np.array_equal(h.point_depth.cpu().numpy(), ho["point_depth"])
        assert np.array_equal(h.point_uv_in_camera.cpu().numpy(), ho["point_uv_in_camera"])
        _check_acc(f"stress_small.{kind}.hook.grad_viewspace", h.grad_viewspace.cpu().numpy(), ho["grad_viewspace"], tol)
        _check_acc(f"stress_small.{kind}.hook.magnitude", h.magnitude_grad_viewspace.cpu().numpy(),
                   ho["magnitude_grad_viewspace"], tol)
        _check_acc(f"stress_small.{kind}.hook.magnitude_image", h.magnitude_grad_viewspace_on_image.cpu().numpy(),
                   ho["magnitude_grad_viewspace_on_image"], tol)
        _check_acc(f"stress_small.{kind}.hook.grad_pointfeatures", h.grad_pointfeatures_in_camera.cpu().numpy(),
                   ho["grad_pointfeatures_in_camera"], tol)
        npix, ref_npix = h.num_affected_pixels.cpu().numpy(), ho["num_affected_pixels"]
        if kind == "masked":   # integer output: a pair can only be counted differently on a fragile pixel
            assert int(np.abs(npix.astype(np.int64) - ref_npix).sum()) <= int(fragile.sum())


def test_reference_stress_distribution_runs():
    """The reference's own stress test (T_RAS:111-150): 1e5 rows of U[0,1) data, only the first 8000 valid,
    1920x1088, f = 500, camera 0.5 behind the cloud -> every Gaussian covers every tile (4.6e7 (tile, Gaussian)
    pairs in the reference's binning).  The reference only checks that it runs; we also check finiteness, that the
    exact cull and the bin layout (the one the operator switches to on such scenes) leave the image bit-identical, and
    run-to-run reproducibility of the gradients."""
    from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Op
    from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene
    s = make_config_scene("stress_t_ras").to("cuda")
    cam = CameraInfo(s.camera_intrinsics, s.height, s.width, 0)
    outs = []
    for cull, bin_shift in ((True, 0), (False, 0), (True, 0), (True, 2), (True, None), (True, None)):
        xyz = s.point_cloud.clone().requires_grad_(True); feat = s.point_cloud_features.clone().requires_grad_(True)
        if bin_shift is not None or len(outs) == 4:
            op = Op(Op.GaussianPointCloudRasterisationConfig())   # the two `None` runs share one operator
        op.exact_tile_cull, op.bin_shift = cull, bin_shift
        image, depth, count = op(Op.GaussianPointCloudRasterisationInput(
            point_cloud=xyz, point_cloud_features=feat, point_object_id=s.point_object_id,
            point_invalid_mask=s.point_invalid_mask, camera_info=cam, q_pointcloud_camera=s.q_pointcloud_camera,
            t_pointcloud_camera=s.t_pointcloud_camera))
        image.sum().backward()  # as T_RAS:149-150
        outs.append((image.detach(), count, feat.grad.clone(), xyz.grad.clone(), op._auto_bin_shift))
    img, cnt, gf, gx, _ = outs[0]
    assert torch.isfinite(img).all() and torch.isfinite(gf).all() and torch.isfinite(gx).all()
    assert img.max() > 0.1 and cnt.max() > 0
    assert not gf[8000:].any() and not gx[8000:].any()          # invalid rows get no gradient
    assert torch.equal(img, outs[1][0]) and torch.equal(cnt, outs[1][1])   # exact cull: bit-identical image
    assert rel_l2(gf.cpu().numpy(), outs[1][2].cpu().numpy()) < 1e-5
    assert torch.equal(gf, outs[2][2]) and torch.equal(gx, outs[2][3])     # reproducible gradients
    for k in (3, 4, 5):   # bin layout (forced, then chosen by the operator after its first frame): same bits
        assert torch.equal(img, outs[k][0]) and torch.equal(cnt, outs[k][1])
        assert torch.equal(gf, outs[k][2]) and torch.equal(gx, outs[k][3])
    assert outs[4][4] == 2 and outs[5][4] == 2   # after one frame of this scene the operator picks 64-pixel bins
    report("stress.sizes", max_count=int(cnt.max()), mean_count=float(cnt.float().mean()))


def _fake_target(size=32):
    # the target of the reference's coverage tests, T_RAS:212-221 / T_RAS:289-298
    img = torch.zeros(size, size, 3)
    img[:5, :2, 0] = 1.0; img[:5, :2, 1] = 0.7
    img[8:24, 8:24, 0] = 0.5; img[8:24, 8:24, 1] = 0.7
    img[20:28, 20:28, 0] = 0.8; img[20:28, 20:28, 1] = 0.1
    return img.cuda()


def test_backward_coverage_adam_descends():
    """The reference's integration test T_RAS:284-351 (and T_ADC-style use): Adam on a 32x32 fake image through the
    operator, SH band = iteration // 100 (the reference runs 10k steps with band = it // 1000; 400 steps here),
    final loss < initial loss -- exercises forward, backward, the in-place q normalisation and band clearing together."""
    from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Op
    g = torch.Generator().manual_seed(0)
    n = 10_000
    target = _fake_target()
    xyz = torch.nn.Parameter(((torch.rand(n, 3, generator=g) - 0.5) * 3).cuda())
    tmp = torch.rand(n, 56, generator=g)
    tmp[:, 4:7] = -4.60517018599
    tmp[:, 7] = 0.5
    feat = torch.nn.Parameter(tmp.cuda())
    cam = CameraInfo(torch.tensor([[32., 0, 16], [0, 32., 16], [0, 0, 1]], device="cuda"), 32, 32, 0)
    q = torch.tensor([[0., 0., 0., 1.]], device="cuda"); t = torch.tensor([[0., 0., -2.]], device="cuda")
    hook_calls = []
    op = Op(Op.GaussianPointCloudRasterisationConfig(near_plane=1., far_plane=10.),
            backward_valid_point_hook=lambda h: hook_calls.append(int(h.point_id_in_camera_list.shape[0])))
    opt = torch.optim.Adam([xyz, feat], lr=0.001)
    losses = []
    for it in range(400):
        opt.zero_grad()
        image, _, _ = op(Op.GaussianPointCloudRasterisationInput(
            point_cloud=xyz, point_cloud_features=feat,
            point_object_id=torch.zeros(n, dtype=torch.int32, device="cuda"),
            point_invalid_mask=torch.zeros(n, dtype=torch.int8, device="cuda"), camera_info=cam,
            q_pointcloud_camera=q, t_pointcloud_camera=t, color_max_sh_band=it // 100))
        loss = ((image - target) ** 2).sum()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    report("coverage", initial=losses[0], final=losses[-1], hook_calls=len(hook_calls))
    assert np.isfinite(losses).all() and losses[-1] < 0.5 * losses[0]
    assert len(hook_calls) == 400 and min(hook_calls) > 0


def test_rasterisation_two_points_smoke():
    """T_RAS:152-205: two points on the optical axis, one masked out; 16x16 image, band 0 (prints only in the
    reference; here the single visible Gaussian must show up at the image centre and match the oracle)."""
    s = small_scene(n=2, size=16, seed=0)
    s.point_cloud = torch.tensor([[0.0, 0.0, 1.0], [0.0, 0.0, 2.0]])
    f = torch.zeros(2, 56); f[:, 3] = 1.0; f[0, 4:7] = 1.0; f[1, 4:7] = 4.0; f[:, 8] = 5.0; f[:, 24] = 1.0; f[:, 40] = 1.0
    s.point_cloud_features = f
    s.point_invalid_mask = torch.tensor([1, 0], dtype=torch.int8)
    s.camera_intrinsics = torch.tensor([[1., 0, 8], [0, 1., 8], [0, 0, 1]])
    s.q_pointcloud_camera = torch.tensor([[0., 0., 0., 1.]]); s.t_pointcloud_camera = torch.zeros(1, 3)
    s.near_plane, s.far_plane = 0.0, 10.0
    ref = oracle_forward(s)
    image, depth, count, *_ = _run_operator(s, None, band=0)
    img = image.detach().cpu().numpy()
    assert np.abs(img - ref["image"]).max() <= PIXEL_TOL
    assert img[8, 8, 0] > 0.3 and count.max().item() == 1


def test_operator_reference_key_layout_path():
    """near_plane < 0 disables the compressed keys: the operator then sorts the reference's 64-bit
    (tile << 32) + depth keys as signed integers (8 radix passes).  Same image and gradients as the oracle."""
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image
    from taichi_3d_gaussian_splatting_amd import hip_ops
    s = small_scene(n=3000, size=128, seed=4, near_plane=-1.0)
    assert hip_ops.key_layout(s.near_plane, s.far_plane, s.depth_to_sort_key_scale, 64)[0] == 0
    f = oracle_forward(s)
    g = make_grad_image(128, 128)
    ob = O.backward(f, g.numpy(), 3)
    image, depth, count, xyz, feat = _run_operator(s, g)
    _check_image("key64.image", image.detach().cpu().numpy(), f["image"], f["margin"] < FRAGILE_MARGIN)
    _check_acc("key64.grad_feat", feat.grad.cpu().numpy(), ob["grad_feat"], FLIP_GRAD_TOL)


def test_operator_degenerate_sizes():
    """N = 0, and M > 0 with K = 0 (points inside the 48-px guard band but right of / below the image get an
    empty tile box, Appendix A.3): zero image, zero gradients, hook still called with M rows."""
    from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Op
    cam = CameraInfo(torch.tensor([[48., 0, 32], [0, 48., 32], [0, 0, 1]], device="cuda"), 64, 64, 0)
    q = torch.tensor([[0., 0., 0., 1.]], device="cuda"); t = torch.tensor([[0., 0., -3.]], device="cuda")
    got = []
    op = Op(Op.GaussianPointCloudRasterisationConfig(), backward_valid_point_hook=got.append)
    for xyz0 in (torch.zeros(0, 3), torch.tensor([[4.4, 0.0, 0.0], [4.5, 4.5, 0.5]])):  # u = 32 + 48*4.4/3 = 102.4
        n = xyz0.shape[0]
        xyz = xyz0.cuda().requires_grad_(True)
        feat = torch.zeros(n, 56, device="cuda"); feat[:, 3] = 1.0; feat[:, 4:7] = -6.0
        feat.requires_grad_(True)
        image, depth, count = op(Op.GaussianPointCloudRasterisationInput(
            point_cloud=xyz, point_cloud_features=feat, point_object_id=torch.zeros(n, dtype=torch.int32, device="cuda"),
            point_invalid_mask=torch.zeros(n, dtype=torch.int8, device="cuda"), camera_info=cam,
            q_pointcloud_camera=q, t_pointcloud_camera=t))
        image.sum().backward()
        assert not image.any() and not depth.any() and not count.any()
        assert xyz.grad.shape == (n, 3) and feat.grad.shape == (n, 56) and not feat.grad.any()
    assert [int(h.point_id_in_camera_list.shape[0]) for h in got] == [0, 2]
    assert got[1].num_overlap_tiles.tolist() == [0, 0]


def test_operator_survives_non_finite_inputs():
    """NaN / Inf rows (what a diverging optimiser produces; the controller prunes them at the next densification,
    ADC:201-206) must not hang or crash the kernels.  They do poison the pixels they are blended into (and through
    those the gradients of the Gaussians sharing them -- as in the reference); everything out of their reach is that
    of the clean render."""
    from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image
    s = small_scene(n=3000, size=128, seed=13)
    clean = _run_operator(s, make_grad_image(128, 128))
    bad = small_scene(n=3000, size=128, seed=13)
    f = bad.point_cloud_features
    f[0, 4] = float("nan")          # scale
    f[1, 0:4] = 0.0                 # zero quaternion -> 0/0 in the in-place normalisation
    f[2, 7] = float("inf")          # opacity logit
    f[3, 8:56] = float("nan")       # colour
    f[4, 5] = 80.0                  # exp(80) overflows fp32 -> inf covariance
    bad.point_cloud[5] = float("nan")
    bad.point_cloud[6, 2] = float("inf")
    image, depth, count, xyz, feat = _run_operator(bad, make_grad_image(128, 128))
    torch.cuda.synchronize()
    finite_rows = torch.isfinite(xyz.grad).all(dim=1) & torch.isfinite(feat.grad).all(dim=1)
    assert finite_rows.float().mean() > 0.5
    assert (xyz.grad[5] == 0).all() and (xyz.grad[6] == 0).all()          # not in the frustum: no gradient
    unaffected = finite_rows & ((feat.grad - clean[4].grad).abs().amax(dim=1) < 1e-6)
    assert unaffected.float().mean() > 0.3                                # rows out of reach: the clean gradients
    finite_px = torch.isfinite(image).all(dim=2)
    assert finite_px.float().mean() > 0.5
    same = (image - clean[0]).abs().amax(dim=2) < 1e-6                    # pixels the broken rows never reach
    assert (same & finite_px).float().mean() > 0.3
