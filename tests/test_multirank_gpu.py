"""Tile-row sharding with REAL ranks on the device: two (and three) processes share the box's single GPU and talk
through gloo (RCCL refuses two ranks on one device; the RCCL call path itself is covered by the world-size-1 test in
test_hip_parity.py).  Every rank renders only its band of tile rows (or its interleaved rows) with the HIP kernels, the collectives of
``distributed.py`` assemble image / depth / count and sum the backward accumulators, and every rank must end with
the image and the dense gradients of the un-sharded operator -- bit for bit for everything that is not a sum over
ranks, and to fp32 summation-order noise for the gradients of Gaussians that straddle rows of different ranks."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, height, width, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Op
        from taichi_3d_gaussian_splatting_amd.distributed import shard_rasteriser_across_tile_rows
        from taichi_3d_gaussian_splatting_amd.synthetic import make_grad_image, make_scene
        dev = torch.device("cuda", 0)
        s = make_scene(n=n, height=height, width=width, s_min=0.01, s_max=0.08, seed=3).to(dev)
        g = make_grad_image(height, width).to(dev)
        hooks = {}

        def run(sharded):
            xyz = s.point_cloud.clone().requires_grad_(True)
            feat = s.point_cloud_features.clone().requires_grad_(True)
            op = Op(Op.GaussianPointCloudRasterisationConfig(),
                    backward_valid_point_hook=lambda h: hooks.__setitem__(sharded, h))
            if sharded:
                shard_rasteriser_across_tile_rows(op, mode=mode)
                lay = op.list_layout(height)
                th = height // 16
                if mode == "interleaved":
                    assert (lay.row_begin, lay.row_step) == (rank, world)
                else:
                    block = -(-th // world)   # equal blocks of ceil(rows / world): the all-gather runs in place
                    assert (lay.row_begin, lay.row_step, lay.row_end) == (min(rank * block, th), 1, min((rank + 1) * block, th))
            image, depth, count = op(Op.GaussianPointCloudRasterisationInput(
                point_cloud=xyz, point_cloud_features=feat, point_object_id=s.point_object_id,
                point_invalid_mask=s.point_invalid_mask,
                camera_info=CameraInfo(camera_intrinsics=s.camera_intrinsics, camera_height=height,
                                       camera_width=width, camera_id=0),
                q_pointcloud_camera=s.q_pointcloud_camera, t_pointcloud_camera=s.t_pointcloud_camera,
                color_max_sh_band=3))
            (image * g).sum().backward()
            if sharded:
                hooks["stats"] = dict(op.exchange_stats)
            return image.detach(), depth.detach(), count, xyz.grad, feat.grad

        base = run(False)
        shard = run(True)
        again = run(True)
        # the accumulator exchange adds the ranks' contributions in a fixed order: run-to-run bitwise reproducible
        assert torch.equal(shard[3].view(torch.int32), again[3].view(torch.int32))
        assert torch.equal(shard[4].view(torch.int32), again[4].view(torch.int32))
        if mode == "bands":   # sparse exchange: a rank sends the rows it produced, not all M
            st = hooks["stats"]
            assert 0 < st["rows_sent"] < hooks[True].point_id_in_camera_list.shape[0] and st["bytes_sent"] < st["dense_bytes"]
        assert torch.equal(base[0], shard[0]) and torch.equal(base[1], shard[1]) and torch.equal(base[2], shard[2])
        for a, b in ((base[3], shard[3]), (base[4], shard[4])):
            assert (a - b).abs().max() <= 2e-5 * a.abs().max()           # sums over ranks: order differs
            assert float(((a - b).abs() > 0).float().mean()) < 0.5
        hb, hs = hooks[False], hooks[True]
        assert torch.equal(hb.point_id_in_camera_list, hs.point_id_in_camera_list)
        assert torch.equal(hb.num_affected_pixels, hs.num_affected_pixels)          # integer sum: exact
        assert torch.equal(hb.num_overlap_tiles, hs.num_overlap_tiles)
        assert torch.allclose(hb.magnitude_grad_viewspace, hs.magnitude_grad_viewspace, rtol=1e-5, atol=1e-12)
        # every rank holds the same result, bit for bit (digest: wrap-around sums of the gradients' bit patterns)
        mine = torch.stack([shard[3].view(torch.int32).long().sum(), shard[4].view(torch.int32).long().sum(),
                            shard[0].view(torch.int32).long().sum()]).cpu()
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        assert all(torch.equal(everyone[0], e) for e in everyone)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,height,mode", [(2, 144, "bands"), (3, 80, "bands"), (2, 144, "interleaved")])
def test_sharded_operator_with_real_ranks(world, height, mode):
    mp.spawn(_worker, args=(world, _free_port(), 6000, height, 160, mode), nprocs=world, join=True)


def _train_worker(rank, world, port, root):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from taichi_3d_gaussian_splatting_amd.GaussianPointTrainer import GaussianPointCloudTrainer as TRN
        cfg = TRN.TrainConfig(
            train_dataset_json_path=os.path.join(root, "train.json"), val_dataset_json_path=os.path.join(root, "val.json"),
            pointcloud_parquet_path=os.path.join(root, "points.parquet"), num_iterations=61, val_interval=10 ** 6,
            feature_learning_rate=5e-3, position_learning_rate=5e-5, initial_downsample_factor=2,
            half_downsample_factor_interval=30, log_loss_interval=10, log_metrics_interval=10 ** 6,
            log_image_interval=10 ** 6, summary_writer_log_dir=os.path.join(root, f"logs_rank{rank}"),
            num_data_loader_workers=0)
        cfg.adaptive_controller_config.num_iterations_warm_up = 20      # densify (with random splits) at 20 and 40
        cfg.adaptive_controller_config.num_iterations_densify = 20
        cfg.gaussian_point_cloud_scene_config.max_num_points_ratio = 2.0
        cfg.gaussian_point_cloud_scene_config.initial_alpha = 0.5
        trainer = TRN(cfg)
        assert trainer.rasterisation.shard == (rank, world, "bands")
        n_before = int((trainer.scene.point_invalid_mask == 0).sum())
        trainer.train()
        live = trainer.scene.point_invalid_mask == 0
        assert int(live.sum()) != n_before                                # the controller acted
        # replicated state is BIT-identical on every rank (same collective results, same seeds)
        digest = torch.cat([trainer.scene.point_cloud.detach().double().sum().view(1),
                            trainer.scene.point_cloud_features.detach().double().sum().view(1),
                            live.double().sum().view(1)]).cpu()
        everyone = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(everyone, digest)
        assert all(torch.equal(everyone[0], e) for e in everyone), everyone
        assert torch.isfinite(digest).all()
        if rank == 0:
            assert os.path.exists(os.path.join(root, "logs_rank0", "metrics.jsonl"))
        else:
            assert not os.path.exists(os.path.join(root, f"logs_rank{rank}", "metrics.jsonl"))   # only rank 0 logs
    finally:
        dist.destroy_process_group()


def test_trainer_replicated_state_stays_identical_across_ranks(tmp_path):
    from tests.test_training_gpu import _write_dataset
    root = str(tmp_path)
    _write_dataset(root, torch.device("cuda:0"))
    mp.spawn(_train_worker, args=(2, _free_port(), root), nprocs=2, join=True)


def _owner_train_worker(rank, world, port, root):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pandas as pd
        from taichi_3d_gaussian_splatting_amd.GaussianPointTrainer import GaussianPointCloudTrainer as TRN
        cfg = TRN.TrainConfig(
            train_dataset_json_path=os.path.join(root, "train.json"), val_dataset_json_path=os.path.join(root, "val.json"),
            pointcloud_parquet_path=os.path.join(root, "points.parquet"), num_iterations=61, val_interval=60,
            feature_learning_rate=5e-3, position_learning_rate=5e-5, initial_downsample_factor=2,
            half_downsample_factor_interval=30, log_loss_interval=10, log_metrics_interval=10 ** 6,
            log_image_interval=10 ** 6, log_validation_image=False,
            summary_writer_log_dir=os.path.join(root, f"owner_logs_rank{rank}"), num_data_loader_workers=0,
            distributed_mode="owner", owner_rebalance_every=7)   # (boundaries move while the resolution changes)
        cfg.adaptive_controller_config.num_iterations_warm_up = 20      # densify at 20 and 40
        cfg.adaptive_controller_config.num_iterations_densify = 20
        cfg.gaussian_point_cloud_scene_config.max_num_points_ratio = 2.0
        cfg.gaussian_point_cloud_scene_config.initial_alpha = 0.5
        trainer = TRN(cfg)
        assert trainer.owner_sharded
        capacity = trainer.scene.point_cloud.shape[0]
        total = torch.tensor([capacity], device="cuda")
        dist.all_reduce(total)
        n_before = int((trainer.scene.point_invalid_mask == 0).sum())
        first = trainer.scene.point_cloud_features.detach().clone()
        trainer.train()
        live = trainer.scene.point_invalid_mask == 0
        assert trainer.scene.point_cloud.shape[0] == capacity            # the rank still holds its block, nothing more
        assert int(live.sum()) != n_before                               # its share of the controller's work happened
        assert not torch.equal(first, trainer.scene.point_cloud_features.detach())   # its parameters were stepped
        assert torch.isfinite(trainer.scene.point_cloud.detach()[live]).all()
        live_all = torch.tensor([int(live.sum())], device="cuda")
        dist.all_reduce(live_all)
        if rank == 0:   # the validation checkpoint holds the blocks of ALL ranks
            df = pd.read_parquet(os.path.join(root, f"owner_logs_rank0", "scene_60.parquet"))
            assert abs(len(df) - int(live_all)) <= int(0.2 * int(live_all)) and len(df) > int(live.sum())
    finally:
        dist.destroy_process_group()


def test_trainer_with_owner_sharded_gaussians(tmp_path):
    """The training loop on the owner-sharded rasteriser (TrainConfig.distributed_mode = "owner"): two ranks share the GPU,
    each owns half of the fixed-capacity point cloud -- parameters, Adam state, controller statistics -- trains through two
    densifications and a validation, and rank 0's checkpoint holds both halves."""
    from tests.test_training_gpu import _write_dataset
    root = str(tmp_path)
    _write_dataset(root, torch.device("cuda:0"))
    mp.spawn(_owner_train_worker, args=(2, _free_port(), root), nprocs=2, join=True)


def test_bench_launches_its_own_ranks_owner_sharded(tmp_path):
    """`python bench.py --gpus 2` with NO launcher (how the driver may call it): bench.py re-runs itself under
    torch.distributed.run, one process per rank (gloo here: the box has one GPU, RCCL refuses two ranks on one device), in
    the default multi-GPU mode -- owner-sharded Gaussians -- and rank 0 prints the one JSON line, reporting the rank count the
    back end saw."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GS_BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--workload", "cfg2_100k_800", "--no-cpu-baseline"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["ranks_seen_by_backend"] == 2 and rec["config"]["backend"] == "gloo"
    assert rec["config"]["sharding"] == "tile-row bands + owner-sharded Gaussians/2" and rec["value"] > 0
    assert rec["config"]["owner_sharding"]["records_sent"] > 0 and rec["scaling"] == "strong"
    # the record carries BOTH partitionings -- the line's own (owner) and the north star's (replicated
    # cloud, bands, one all-gather of the rows) -- each with the ranks the back end saw, every rank's pass times and the bytes
    # its collectives moved
    assert len(rec["config"]["per_rank"]) == 2
    for r in rec["config"]["per_rank"]:
        assert r["step"] > 0 and r["forward"] > 0 and r["backward"] > 0
        assert r["collectives_bytes_per_step"]["forward_all_to_all_sent"] > 0
        assert r["collectives_bytes_per_step"]["image_all_gather_contributed"] > 0
    bands = rec["variants"]["shard_mode_bands"]
    assert bands["sharding"] == "tile-row bands/2" and bands["ranks_seen_by_backend"] == 2 and bands["value"] > 0
    assert len(bands["per_rank"]) == 2
    for r in bands["per_rank"]:
        assert r["step"] > 0 and r["collectives_bytes_per_step"]["image_all_gather_contributed"] > 0
        assert r["collectives_bytes_per_step"]["accumulator_exchange_sent"] > 0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_bench_launches_its_own_ranks_over_rccl():
    """The same on a multi-GPU node over RCCL: one rank per GPU (skipped on the one-GPU boxes of this pool)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GS_BENCH_DIST_BACKEND"):
        env.pop(key, None)
    n = min(torch.cuda.device_count(), 8)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "5", "--warmup", "2",
                          "--no-cpu-baseline"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == n and rec["config"]["ranks_seen_by_backend"] == n and rec["config"]["backend"] == "nccl"


def test_bench_multi_rank_contract(tmp_path):
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one process per rank), on this
    1-GPU box with the gloo transport: rank 0 prints exactly one JSON line with the contract's fields."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GS_BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
         "--workload", "cfg2_100k_800", "--no-cpu-baseline", "--shard-mode", "bands"],
        cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "strong"
    assert rec["config"]["sharding"] == "tile-row bands/2" and rec["value"] > 0 and rec["unit"] == "Mpixels/s"
    assert rec["roofline"]["bound"] == "hbm" and 0 < rec["roofline"]["frac"] < 1
    owner = rec["variants"]["shard_mode_owner"]   # (the line asked for bands: the other mode rides along)
    assert owner["sharding"] == "tile-row bands + owner-sharded Gaussians/2" and owner["value"] > 0
    assert len(owner["per_rank"]) == 2 and len(rec["config"]["per_rank"]) == 2
