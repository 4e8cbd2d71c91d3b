#!/usr/bin/env python
"""bench.py -- rendered Mpixels/s (forward + backward) of the MI355X rasteriser.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one forward + backward pass of the drop-in operator over one synthetic frame with the
inputs already resident in HBM, INCLUDING the backward hook every training iteration installs, as the
trainer configures it between densifications (a no-op consumer; the operator produces the hook's nine compact
M-indexed fields, the tenth -- the [M,56] copy of the feature gradients, which the trainer only asks for on the
iteration of a densification -- with --hook-feature-copy; --no-hook removes the hook).
Default workload = the configuration BASELINE.json's metric is quoted on: 1e6 random-init Gaussians,
1920x1080 -> rendered at 1920x1072 (the reference asserts H % 16 == 0 and its data path crops,
RAS:1193-1194, ImagePoseDataset.py:86-88), SH degree 3.
N > 1: the same frame is sharded over contiguous bands of 16-pixel tile rows (one process per GPU,
RCCL all-gather of the rendered rows + all-reduce of the per-Gaussian gradient accumulators), so the
scaling is STRONG: total work is fixed, value = frame pixels / max-over-ranks step time.

Warm-up: the W warm-up steps are followed by untimed steps until the GPU has been under the operator's load for
WARMUP_FLOOR_MS = 40 ms (config.warmup_floor_ms, config.warmup_steps_run): a GPU that has idled needs ~20 ms of this load to
reach the shader clock it then holds, and the VALU-bound blend kernels run 10 % slower at the start of that ramp
(profiles/r05_clock_ramp.txt) -- `--warmup 5` alone timed the ramp, not the path.  GS_BENCH_WARMUP_FLOOR_MS=0: exactly W.
The literal protocol is measured too and reported beside it: `ms_per_step_strict_warmup` = the same K steps timed straight
behind exactly W warm-up steps (they then count as warm-up load towards the floor).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  step_ms      : per-step GPU time from HIP events on the launch stream around each step: median, p90, min -- taken over K
                 more steps AFTER the timed region (an event between two steps costs the stream ~6 us; ms_per_step is the
                 wall-clock mean over exactly K steps between two barriers + synchronisations, nothing else enqueued)
  roofline     : achieved algorithmic HBM GB/s of the dominant kernel vs the 8 TB/s peak, measured live with HIP
                 events on the launch stream (plus all stage times and the whole-path figure); `traffic` = PMC HBM
                 bytes per launch from the committed profile, {raw, gfx950_corrected} (the guide's correction doubles
                 the fetch counter: right for streaming reads, too much for gathers) -- only while the kernel sources are
                 the profiled ones (hash recorded next to the profile), else null; `valu` = the VALU-issue view of the
                 same kernel (the blend kernels are bound by VALU issue, not HBM: DESIGN.md section 5);
                 `lane_utilisation` = how often the hit path runs and with how many live lanes, how often a decision is
                 settled by the reference's own expression (counting build, tools/blend_stats.py; same hash rule)
  variants     : the same K steps in other configurations -- without the [M,56] hook copy; over a seeded camera orbit
                 (`camera_path`: lists, sizes and the size speculation change every step)
  cpu_baseline : the CPU oracle (a from-source port of the reference kernels; the reference itself
                 cannot run here -- taichi is absent and its kernels are CUDA-only) timed on this
                 box's host cores on one full frame of the same workload.

Other lines (not the driver's): --forward-only (inference: no_grad, with --rgb-only the reference's rgb_only
configuration; metric "rendered Mpixels/s (fwd)"), --workload {cfg1_10k_256, cfg2_100k_800, cfg3_400k_1080p,
cfg4_2m_1080p, stress_t_ras}.
"""
from __future__ import annotations

import argparse
import gc
import hashlib
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# VALU issue ceiling: 256 CUs x 4 SIMD-32, one wave64 fp32 instruction per 2 cycles per SIMD at 2.4 GHz
VALU_PEAK_WAVE_INSTR_PER_S = 256 * 4 * 2.4e9 / 2.0
PROFILE_TAG = "r06"
# Warm-up floor (ms of continuous load before the timed region; profiles/r05_clock_ramp.txt): the W warm-up steps are
# followed by untimed steps until the GPU has been under this load that long.  0 = exactly W steps.
WARMUP_FLOOR_MS = float(os.environ.get("GS_BENCH_WARMUP_FLOOR_MS", "40"))


def algorithmic_bytes(n, m, k, p, key_bytes=8, k_tile=None):
    """Compulsory HBM bytes per launch of each stage: every
    stage input read once, every output written once; N points, M visible, K sort keys, k_tile (tile, Gaussian) pairs
    that are blended (= K with per-tile keys; with binned lists fewer keys are sorted but the same pairs are
    blended), P pixels; key_bytes = 4 (compressed keys) or 8 (reference layout)."""
    pair = key_bytes + 4
    k_tile = k if k_tile is None else k_tile
    return {
        "filter_compact": 18 * n + 4 * m,
        "preprocess": 244 * m + 16 * m + 48 * m + 8 * m + 4 * m,   # row+xyz+ids+obj, q write, attrs, counts
        "scan_block_sums": 8 * (m // 256 + 1),
        "make_keys": 32 * m + pair * k,
        "sort_pairs": 2 * pair * k,
        "tile_ranges": key_bytes * k,
        "blend_forward": 48 * k_tile + 28 * p,
        "blend_backward": 44 * k_tile + 28 * p + 48 * m,   # list re-gather, per-pixel in/out, one record per Gaussian
        "reduce_partials": 48 * m + 8 * m,            # (the slot records themselves are blend_backward's output)
        "point_backward": 244 * m + 48 * m + 248 * n,  # (fused form, 1 GPU: the 48 m are one slot record per Gaussian)
    }


def kernel_source_hash() -> str:
    """sha256 over the HIP sources (the profile summaries record the hash they were taken with)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "taichi_3d_gaussian_splatting_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(csrc, name), "rb") as fh:
                h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def profiled_counters(kernel: str):
    """(HBM bytes per launch, VALU wave-instructions per launch) of `kernel` from the committed rocprofv3 PMC
    summaries (profiles/<tag>_hbm_traffic.csv: separate --pmc FETCH_SIZE / WRITE_SIZE passes over this same command,
    tools/profile.sh, with the gfx950 correction of the MI355X guide -- FETCH_SIZE counts 128-B requests as 64 B ->
    doubled; profiles/<tag>_sq_counters.csv: SQ_INSTS_VALU).  (None, None) when the profile is absent or was taken
    with other kernel sources than the ones in the tree (profiles/<tag>_source_hash.txt)."""
    try:
        with open(os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_source_hash.txt")) as fh:
            if fh.read().split()[0] != kernel_source_hash():
                return None, None
    except (OSError, IndexError):
        return None, None
    import csv
    traffic = valu = None
    try:
        with open(os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_hbm_traffic.csv")) as fh:
            for row in csv.DictReader(fh):
                if row[""].startswith(kernel + "_kernel"):
                    # both figures: the guide's gfx950 correction doubles FETCH_SIZE (128-B requests counted as 64 B) -- right
                    # for streaming reads, too much for kernels whose reads are 16-B / 64-B gathers (profiles/README.md)
                    traffic = {"raw": int((float(row["hbm_read_MB_raw"]) + float(row["hbm_write_MB"])) * 1e6),
                               "gfx950_corrected": int((float(row["hbm_read_MB_gfx950_corrected"]) +
                                                        float(row["hbm_write_MB"])) * 1e6)}
    except (OSError, KeyError, ValueError):
        pass
    try:
        with open(os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_sq_counters.csv")) as fh:
            for row in csv.DictReader(fh):
                if row["kernel"].startswith(kernel + "_kernel"):
                    valu = float(row["SQ_INSTS_VALU"])
    except (OSError, KeyError, ValueError):
        pass
    return traffic, valu


def camera_orbit(q0, t0, n_poses: int, seed: int = 7):
    """n poses T_pointcloud_camera = Rot * T0: the camera of the workload carried around the cloud's centre -- azimuth 2 pi k / n
    (+ a seeded jitter), elevation within +- 0.2 rad -- at its own distance, looking at the centre as before.
    Quaternions (x, y, z, w) as everywhere in the operator."""
    import math
    g = torch.Generator().manual_seed(seed)
    out = []
    q0c, t0c = q0.detach().cpu().double(), t0.detach().cpu().double()

    def qmul(a, b):
        ax, ay, az, aw = a
        bx, by, bz, bw = b
        return torch.tensor([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                             aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz], dtype=torch.float64)

    def rotate(q, v):
        x, y, z, w = q
        R = torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                          [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                          [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]], dtype=torch.float64)
        return R @ v
    for k in range(n_poses):
        az = 2 * math.pi * (k + 0.3 * float(torch.rand(1, generator=g))) / n_poses
        el = 0.4 * float(torch.rand(1, generator=g)) - 0.2
        qy = torch.tensor([0.0, math.sin(az / 2), 0.0, math.cos(az / 2)], dtype=torch.float64)
        qx = torch.tensor([math.sin(el / 2), 0.0, 0.0, math.cos(el / 2)], dtype=torch.float64)
        rot = qmul(qy, qx)
        qs = torch.stack([qmul(rot, q0c[i]) for i in range(q0c.shape[0])])
        ts = torch.stack([rotate(rot, t0c[i]) for i in range(t0c.shape[0])])
        out.append((qs.to(q0.dtype).to(q0.device), ts.to(t0.dtype).to(t0.device)))
    return out


def profiled_lane_utilisation():
    """Path statistics of the two blend kernels at the headline size (profiles/<tag>_blend_path_stats.json: counters of a
    -DGS_STATS=1 build, tools/blend_stats.py) -- the share of (wave, entry) visits that run the hit path and the live pixels /
    lanes on it, the visits settled by the exact-decision paths.  None when absent or taken with other kernel sources."""
    try:
        with open(os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_source_hash.txt")) as fh:
            if fh.read().split()[0] != kernel_source_hash():
                return None
        with open(os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_blend_path_stats.json")) as fh:
            d = json.load(fh)
        return {"workload": d["workload"], "forward": d.get("forward"), "backward": d.get("backward"),
                "exact_decisions": {k: d["counters"][k] for k in ("fwd_careful_entries", "fwd_exact_alpha", "fwd_replays",
                                                                  "bwd_bracketed", "bwd_exact_alpha")}}
    except (OSError, KeyError, ValueError, IndexError):
        return None


def self_launch(n_ranks: int) -> int:
    """Re-run this command line as `n_ranks` processes (torch.distributed.run, one per GPU of this node, rendezvous on
    127.0.0.1 at a free port); the children see WORLD_SIZE and take the normal path.  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="headline_1m_1080p")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-profile", action="store_true")
    ap.add_argument("--no-hook", action="store_true", help="time the backward without a backward hook")
    ap.add_argument("--hook-feature-copy", action="store_true", help="(default since round 4; kept for old command lines)")
    ap.add_argument("--no-hook-feature-copy", action="store_true",
                    help="the hook does not receive the [M,56] compact copy of the feature gradients (RAS:1132) -- how "
                         "GaussianPointTrainer configures the operator between densifications.  The default line times "
                         "the DEFAULT operator (copy on, as the reference always gathers it, RAS:1131-1133) and reports "
                         "this variant beside it (`variants`)")
    ap.add_argument("--forward-only", action="store_true", help="inference line: forward under no_grad")
    ap.add_argument("--rgb-only", action="store_true", help="with --forward-only: the reference's rgb_only config")
    ap.add_argument("--shard-mode", default="owner", choices=["owner", "bands", "interleaved"],
                    help="N > 1: 'owner' (default) = tile-row bands for the pixels + owner-sharded Gaussians with a routed "
                         "exchange (owner_sharding.py: nothing per-Gaussian is replicated); 'bands' / 'interleaved' = the "
                         "replicated point cloud of distributed.py")
    ap.add_argument("--bin-shift", type=int, default=None,
                    help="force the list layout (0 = per-tile keys, 1 = 2x2-tile bins, 2 = 4x4; default: the operator's own "
                         "choice per frame) -- for A/B measurements of that choice")
    ap.add_argument("--no-pin", action="store_true",
                    help="leave the host threads where the scheduler puts them (default: all threads of the process on one "
                         "L3 complex of the GPU's NUMA node, taichi_3d_gaussian_splatting_amd/host_affinity.py)")
    ap.add_argument("--camera-path", type=int, default=8,
                    help="poses of the moving-camera variant reported beside the static line (`variants.camera_path`): a "
                         "seeded orbit at the workload's camera distance, one pose per step in turn, so that list layout, sizes "
                         "and the size speculation change every step (the reference's own protocol renders dataset poses, "
                         "benchmark/inference_benchmark.py:109-160); 0 = off")
    ap.add_argument("--static-scene", action="store_true",
                    help="let the forward skip the write-back of quaternions that are already normalised (default: "
                         "training-like -- the in-place normalisation RAS:196-205 writes every visible row, every frame)")
    args = ap.parse_args()

    import torch.distributed as dist
    from taichi_3d_gaussian_splatting_amd import _lib
    if not os.path.exists(_lib.LIB_PATH) and int(os.environ.get("LOCAL_RANK", "0")) == 0:
        _lib.build()  # in-tree build of the HIP library (normally done by __graft_entry__.build())
    from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Op
    from taichi_3d_gaussian_splatting_amd import hip_ops
    from taichi_3d_gaussian_splatting_amd.distributed import shard_rasteriser_across_tile_rows
    from taichi_3d_gaussian_splatting_amd.synthetic import make_config_scene, make_grad_image

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: launch the ranks ourselves -- one process per GPU under torch.distributed.run
        # on 127.0.0.1 (the container's hostname may not resolve), RCCL backend -- and relay rank 0's line
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    assert torch.cuda.is_available(), "bench.py needs HIP devices"
    # one rank per GPU.  (Test hook: with fewer devices than ranks -- a 1-GPU box -- ranks share devices, which
    # RCCL refuses, so GS_BENCH_DIST_BACKEND=gloo lets the multi-rank plumbing be exercised there.)
    backend = os.environ.get("GS_BENCH_DIST_BACKEND", "nccl")  # nccl == RCCL on ROCm
    if world > 1 and backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"--gpus {world} but this node exposes {torch.cuda.device_count()} GPU(s): RCCL needs one device "
                         f"per rank")
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    device = torch.device("cuda", device_index)
    # the launching threads (this one, autograd's, the HIP runtime's) on cores that share an L3: frames bound by the host
    # run up to 1.8x faster than with the threads scattered over a 256-thread box; GPU-bound frames are indifferent
    from taichi_3d_gaussian_splatting_amd import host_affinity
    torch.cuda.init()
    pinned = None if args.no_pin else host_affinity.pin_host_threads(device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)

    host_scene = make_config_scene(args.workload)
    s = host_scene.to(device)
    grad_image = make_grad_image(s.height, s.width).to(device)
    cfg = Op.GaussianPointCloudRasterisationConfig(near_plane=s.near_plane, far_plane=s.far_plane,
                                                   depth_to_sort_key_scale=s.depth_to_sort_key_scale,
                                                   rgb_only=bool(args.rgb_only and args.forward_only))
    hook_calls = []
    hook = None if (args.no_hook or args.forward_only) else (lambda h: hook_calls.append(1))
    owner_mode = world > 1 and args.shard_mode == "owner"
    cam = CameraInfo(camera_intrinsics=s.camera_intrinsics, camera_height=s.height, camera_width=s.width, camera_id=0)

    def build_mode(shard_mode):
        """The operator of one sharding mode with this rank's inputs -> dict(module, op, xyz, feat, inp, rows, owner).
        "owner": this rank OWNS a contiguous block of the point cloud (its inputs and gradients are that block);
        "bands" / "interleaved": the north-star partitioning -- replicated cloud, tile rows per rank, all-gather of the rows."""
        owner = world > 1 and shard_mode == "owner"
        rws = slice(None)
        if owner:
            from taichi_3d_gaussian_splatting_amd.owner_sharding import OwnerShardedRasterisation, owned_point_rows
            block = owned_point_rows(s.point_cloud.shape[0], rank, world)
            rws = slice(block.start, block.stop)
            mod = OwnerShardedRasterisation(cfg, backward_valid_point_hook=hook)
            core = mod.core                     # (the options below live on the rank's core)
        else:
            core = mod = Op(cfg, backward_valid_point_hook=hook)
        # the operator's default (True): the reference always gathers the [M,56] field (RAS:1131-1133).  The trainer switches
        # it off between densifications: that variant is timed as well and reported in `variants`
        core.hook_feature_gradients = not args.no_hook_feature_copy
        if world > 1 and not owner:
            shard_rasteriser_across_tile_rows(core, mode=shard_mode)
        x = s.point_cloud[rws].clone().requires_grad_(True)
        f = s.point_cloud_features[rws].clone().requires_grad_(True)
        i = Op.GaussianPointCloudRasterisationInput(
            point_cloud=x, point_cloud_features=f, point_object_id=s.point_object_id[rws],
            point_invalid_mask=s.point_invalid_mask[rws], camera_info=cam, q_pointcloud_camera=s.q_pointcloud_camera,
            t_pointcloud_camera=s.t_pointcloud_camera, color_max_sh_band=3)
        # Training-like steady state: an optimiser step leaves every quaternion off unit length, so every forward's in-place
        # normalisation (RAS:196-205) writes the visible rows back.  On this static scene the operator would skip that write
        # after the first frame (the stored quaternions are already normalised); `always_store_normalised_rotation` makes it
        # pay the write every frame, as training does (same memory contents).  --static-scene: the skip stays.
        core.always_store_normalised_rotation = not args.static_scene
        if args.bin_shift is not None:
            core.bin_shift = args.bin_shift
        return {"module": mod, "op": core, "xyz": x, "feat": f, "inp": i, "rows": rws, "owner": owner}

    cur = build_mode(args.shard_mode)
    module, op, xyz, feat, inp, rows = cur["module"], cur["op"], cur["xyz"], cur["feat"], cur["inp"], cur["rows"]

    def step():
        if args.forward_only:
            with torch.no_grad():
                return module(inp)[0]
        xyz.grad = None
        feat.grad = None
        image, depth, count = module(inp)
        image.backward(grad_image)
        return image

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    warmup_steps_run = []   # per timed_run call: W + the steps of the warm-up floor

    strict_warmup_ms = []   # wall-clock ms per step of K steps timed straight after exactly W warm-up steps (first timed_run)

    def wall_clock_steps(steps):
        """exactly `steps` steps between two fences, nothing else on the stream -> ms per step (max over ranks)"""
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return 1e3 * elapsed / steps

    def timed_run(warmup, steps, strict_too=False):
        """-> (wall-clock ms per step, max over ranks; {median, p90, min} of the per-step HIP-event times)"""
        # (as `timeit` does: no garbage collection inside a timed region -- a full collection of this process takes ~45 ms of
        # the host thread, which runs only ~1 ms ahead of the GPU.  Collected HERE, in front of the warm-up: 45 ms of idling
        # between warm-up and timed region would send the GPU down its clock ramp again)
        gc.collect()
        gc.disable()
        try:
            return _timed_run(warmup, steps, strict_too)
        finally:
            gc.enable()

    def _timed_run(warmup, steps, strict_too):
        t_load, after_first = None, 0
        for _ in range(warmup):
            step()
            if t_load is None:   # (the first step of a fresh operator is mostly host work: library, allocations)
                torch.cuda.synchronize()
                t_load = time.perf_counter()
            else:
                after_first += 1
        # THE LITERAL PROTOCOL first: K steps timed behind exactly W warm-up steps, reported beside the
        # floored number as ms_per_step_strict_warmup.  These K steps are load like any other: they count towards the floor.
        extra = 0
        if strict_too and WARMUP_FLOOR_MS > 0 and warmup > 0:
            strict_warmup_ms.append(wall_clock_steps(steps))
            after_first += steps
            extra += steps
        # warm-up floor: a GPU that has idled needs ~20 ms of THIS load to reach the shader clock it then holds -- the
        # VALU-bound blend kernels are 10 % slower at the start of the ramp (profiles/r05_clock_ramp.txt); five warm-up steps
        # are 6 ms.  Keep stepping, untimed, until the load has lasted WARMUP_FLOOR_MS (reported in config.warmup_steps_run).
        # The number of extra steps is computed once, the same on every rank (steps of a sharded frame hold collectives).
        if WARMUP_FLOOR_MS > 0:
            while t_load is None or after_first < 2:   # (two steps behind the first one: a step-time estimate)
                step()
                extra += 1
                if t_load is None:
                    torch.cuda.synchronize()
                    t_load = time.perf_counter()
                else:
                    after_first += 1
            torch.cuda.synchronize()
            loaded_ms = 1e3 * (time.perf_counter() - t_load)
            if world > 1:
                t = torch.tensor([loaded_ms], dtype=torch.float64, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                loaded_ms = float(t.item())
            more = int(math.ceil(max(WARMUP_FLOOR_MS - loaded_ms, 0.0) / max(loaded_ms / after_first, 1e-3)))
            for _ in range(min(more, 400)):
                step()
            extra += min(more, 400)
        warmup_steps_run.append(warmup + extra)
        # THE timed region: exactly `steps` steps between two fences, nothing else on the stream (an event record between
        # two steps holds the next kernel back by ~6 us: the per-step statistics below come from a pass of their own)
        ms = wall_clock_steps(steps)
        # per-step distribution (not the headline number): the same steps once more, two events per step on torch's current
        # stream (= the stream every kernel is launched on)
        begins = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        ends = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        for i in range(steps):
            begins[i].record()
            step()
            ends[i].record()
        fence()
        per_step = sorted(begins[i].elapsed_time(ends[i]) for i in range(steps))
        return ms, {"median": round(per_step[len(per_step) // 2], 4),
                    "p90": round(per_step[int(0.9 * (len(per_step) - 1))], 4),
                    "min": round(per_step[0], 4)}

    ms_per_step, step_ms = timed_run(args.warmup, args.steps, strict_too=True)
    # the other hook configuration, same K steps (not the driver's number): the trainer's steady state between
    # densifications leaves out the [M,56] hook copy
    variants = {}
    if hook is not None and not args.no_hook_feature_copy:
        op.hook_feature_gradients = False
        v_ms, v_step = timed_run(min(args.warmup, 5), args.steps)
        op.hook_feature_gradients = True
        variants["hook_without_feature_copy"] = {"ms_per_step": round(v_ms, 4), "step_ms": v_step,
                                                 "value": round(s.height * s.width / 1e6 / (v_ms / 1e3), 3)}
    pixels = s.height * s.width
    value = pixels / 1e6 / (ms_per_step / 1e3)

    # ---------------------------------------------------------------- N > 1: what each rank did, and the OTHER sharding mode
    # A scaling record is only interpretable if it says which partitioning it timed.  The line's number is the mode asked
    # for (default: owner-sharded Gaussians -- two all-to-alls + the all-gather of the rows); `variants.shard_mode_*` holds the
    # same K steps in the other one (the north star's: replicated cloud, tile-row bands, ONE all-gather of the rendered rows +
    # the sparse accumulator exchange of the backward pass) -- each with the ranks the backend saw, every rank's own pass times
    # (HIP events on its launch stream) and the bytes each collective moved per step.
    def per_rank_report(mode_state, steps):
        """every rank times `steps` more steps with events around its forward and backward passes -> gathered lists"""
        mod, x, f, i = mode_state["module"], mode_state["xyz"], mode_state["feat"], mode_state["inp"]
        ev = lambda: torch.cuda.Event(enable_timing=True)   # noqa: E731
        marks = []
        for _ in range(steps):
            a, b, c = ev(), ev(), ev()
            x.grad = None
            f.grad = None
            a.record()
            image, depth, count = mod(i)
            b.record()
            image.backward(grad_image)
            c.record()
            marks.append((a, b, c))
        fence()
        med = lambda v: sorted(v)[len(v) // 2]   # noqa: E731
        mine = {"forward": round(med([a.elapsed_time(b) for a, b, c in marks]), 4),
                "backward": round(med([b.elapsed_time(c) for a, b, c in marks]), 4),
                "step": round(med([a.elapsed_time(c) for a, b, c in marks]), 4)}
        core = mode_state["op"]
        h, w = s.height, s.width
        from taichi_3d_gaussian_splatting_amd.distributed import padded_image_rows
        gathered_rows = padded_image_rows(h, world)
        if mode_state["owner"]:
            st = dict(mod.last_frame_stats)
            mine["collectives_bytes_per_step"] = {
                "forward_all_to_all_sent": int(st.get("bytes_sent_forward", 0)),
                "backward_all_to_all_sent": int(st.get("records_received", 0)) * 48,   # one 48-B row back per received record
                "image_all_gather_contributed": gathered_rows // world * w * 20}          # rgb + depth + count of this rank's band
            mine["records_received"] = int(st.get("records_received", 0))
        else:
            st = dict(getattr(core, "exchange_stats", {}) or {})
            mine["collectives_bytes_per_step"] = {
                "image_all_gather_contributed": gathered_rows // world * w * 20,
                "accumulator_exchange_sent": int(st.get("bytes_sent", 0)),
                "accumulator_rows_sent": int(st.get("rows_sent", 0))}
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        return everyone

    if world > 1 and not args.forward_only:
        per_rank = per_rank_report(cur, min(args.steps, 10))
        other_mode = "bands" if owner_mode else "owner"
        main_state = (module, op, xyz, feat, inp, rows)
        other = build_mode(other_mode)
        module, op, xyz, feat, inp, rows = (other[k] for k in ("module", "op", "xyz", "feat", "inp", "rows"))
        o_ms, o_step = timed_run(min(args.warmup, 5), args.steps)
        variants["shard_mode_" + other_mode] = {
            "sharding": (f"tile-row bands + owner-sharded Gaussians/{world}" if other["owner"] else f"tile-row {other_mode}/{world}"),
            "ms_per_step": round(o_ms, 4), "step_ms": o_step, "value": round(pixels / 1e6 / (o_ms / 1e3), 3),
            "ranks_seen_by_backend": dist.get_world_size(), "backend": backend,
            "per_rank": per_rank_report(other, min(args.steps, 10))}
        module, op, xyz, feat, inp, rows = main_state
        for _ in range(2):
            step()
    else:
        per_rank = None
    # the moving-camera variant: the same K steps over a seeded orbit (the driver's number stays the static line above)
    if args.camera_path > 0 and world == 1:
        poses = camera_orbit(s.q_pointcloud_camera, s.t_pointcloud_camera, args.camera_path)
        inputs = [Op.GaussianPointCloudRasterisationInput(
            point_cloud=xyz, point_cloud_features=feat, point_object_id=s.point_object_id[rows],
            point_invalid_mask=s.point_invalid_mask[rows], camera_info=cam, q_pointcloud_camera=q, t_pointcloud_camera=t,
            color_max_sh_band=3) for q, t in poses]
        static_inp, turn = inp, [0]
        before = dict(op.speculation_stats)

        def moving_step():
            nonlocal inp
            inp = inputs[turn[0] % len(inputs)]
            turn[0] += 1
            return step_static()
        step_static, step = step, moving_step
        v_ms, v_step = timed_run(max(min(args.warmup, 5), len(inputs)), args.steps)
        step, inp = step_static, static_inp
        after = dict(op.speculation_stats)
        variants["camera_path"] = {
            "poses": len(inputs), "ms_per_step": round(v_ms, 4), "step_ms": v_step,
            "value": round(pixels / 1e6 / (v_ms / 1e3), 3), "vs_static": round(v_ms / ms_per_step, 4),
            "speculation": {k: after[k] - before[k] for k in after},
            "note": "seeded orbit about the cloud's centre at the workload's camera distance (+- 0.2 rad of elevation), one pose "
                    "per step in turn: every step rebuilds lists of another layout and size"}
        for _ in range(2):   # (the stage profile below runs on the static camera again: let its size guesses settle)
            step()

    # ---------------------------------------------------------------- per-stage timing (rank-local)
    n = s.point_cloud.shape[0]
    roofline, stages_ms, sizes = None, {}, {}
    if not args.no_stage_profile and not owner_mode:   # (owner mode: tools/owner_shard_bench.py times a rank's phases)
        layout = op.list_layout(s.height)
        ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
        reps = max(3, min(args.steps, 10))
        acc_ms = {}

        def timed(name, fn):
            a, b = ev(), ev()
            a.record()
            out = fn()
            b.record()
            acc_ms.setdefault(name, []).append((a, b))
            return out

        num_bins = layout.num_bins(s.width, s.height)
        kdb, db, tb = hip_ops.key_layout(s.near_plane, s.far_plane, s.depth_to_sort_key_scale, num_bins)
        q_cp, t_cp = hip_ops.pose_inverse(s.q_pointcloud_camera, s.t_pointcloud_camera)
        for _ in range(reps):
            f = feat.detach()
            vmask, ids, counters = timed("filter_compact", lambda: hip_ops.filter_compact(
                s.point_cloud, s.point_invalid_mask, s.point_object_id, s.camera_intrinsics, q_cp, t_cp,
                s.near_plane, s.far_plane, s.width, s.height))
            attrs, ntiles, nowned, bsums, bsums_full = timed("preprocess", lambda: hip_ops.preprocess(
                s.point_cloud, f, s.point_object_id, s.camera_intrinsics, q_cp, t_cp, ids, s.width, s.height, layout,
                s.depth_to_sort_key_scale, counters, always_store_rotation=op.always_store_normalised_rotation))
            k, n_slots, max_dq, _m = timed("scan_block_sums", lambda: hip_ops.scan_block_sums(bsums, counters, bsums_full))
            kdb, db, tb = hip_ops.key_layout(s.near_plane, s.far_plane, s.depth_to_sort_key_scale, num_bins, max_dq)
            keys, payload, slot_off = timed("make_keys", lambda: hip_ops.make_keys(
                attrs, nowned, bsums, k, s.width, s.height, s.depth_to_sort_key_scale, layout, kdb, ntiles, bsums_full))
            keys, payload = timed("sort_pairs", lambda: hip_ops.sort_pairs(keys, payload, db, tb, kdb, in_place=False,
                                                                            bins_in_any_order=True))
            start, end = timed("tile_ranges", lambda: hip_ops.tile_ranges(keys, num_bins, kdb))
            # as the operator runs them: tiles dispatched longest first, the backward on the per-tile lists the (binned)
            # forward pass wrote out
            tile_work = torch.empty(hip_ops.num_owned_tiles(s.width, s.height, layout), dtype=torch.int32, device=device)
            emit = bool(op.backward_on_walked_lists and layout.filter != 0 and layout.bin_shift <= 2)
            fwd = timed("blend_forward", lambda: hip_ops.blend_forward(
                start, end, payload, attrs, s.width, s.height, layout, ordered=op.ordered_dispatch,
                tile_work=tile_work if op.ordered_dispatch else None, emit_walked_lists=emit))
            image, depth, acc_alpha, last_eff, count = fwd[:5]
            b_start, b_list, b_layout = (fwd[5], fwd[6], hip_ops.walked_layout(layout)) if emit else (start, payload, layout)
            partials, flags, mag = timed("blend_backward", lambda: hip_ops.blend_backward_partials(
                b_start, b_list, attrs, grad_image, acc_alpha, last_eff, slot_off, n_slots, s.width, s.height, b_layout,
                tile_work=tile_work if op.ordered_dispatch else None))
            acc = None
            if world > 1 or not op.fused_slot_reduction:
                acc = timed("reduce_partials", lambda: hip_ops.reduce_partials(slot_off, ntiles, flags, partials, None, attrs,
                                                                               s.width, s.height))
            timed("point_backward", lambda: hip_ops.point_backward(
                s.point_cloud, f, s.point_object_id, s.camera_intrinsics, q_cp, t_cp, s.t_pointcloud_camera, ids,
                acc, attrs, 3, cfg.grad_q_factor, cfg.grad_s_factor, cfg.grad_alpha_factor, cfg.grad_color_factor,
                cfg.grad_high_order_color_factor, hook is not None, vmask, nowned,
                want_visible_features=hook is not None and op.hook_feature_gradients,
                want_hook_fields=hook is not None, slots=None if acc is not None else (slot_off, ntiles, flags, partials),
                width=s.width, height=s.height))
        torch.cuda.synchronize()
        m = int(ids.shape[0])
        k_tile = int(k)
        if layout.bin_shift:   # the (tile, Gaussian) pairs the blend kernels recover from the bin lists: count them once
            import dataclasses
            per_tile = dataclasses.replace(layout, bin_shift=0)
            counters_t = torch.zeros_like(counters)
            counters_t[hip_ops.COUNTER_NUM_VISIBLE] = m
            _, _, _, bs_t, bsf_t = hip_ops.preprocess(
                s.point_cloud, feat.detach(), s.point_object_id, s.camera_intrinsics, q_cp, t_cp, ids, s.width,
                s.height, per_tile, s.depth_to_sort_key_scale, counters_t)
            k_tile = int(hip_ops.scan_block_sums(bs_t, counters_t, bsf_t)[0])
        sizes = {"N": n, "M": m, "K": int(k), "K_tile": k_tile, "P": pixels,
                 "tiles": (s.width // 16) * (s.height // 16), "bin_shift": layout.bin_shift}
        def median_ms(pairs):   # median over the repetitions after the first (one slow outlier must not name a stage)
            t = sorted(a.elapsed_time(b) for a, b in (pairs[1:] or pairs))
            return t[len(t) // 2]
        stages_ms = {name: median_ms(pairs) for name, pairs in acc_ms.items()}
        p_owned = pixels if world == 1 else pixels * len(layout.owned_rows(s.height)) / (s.height // 16)
        bytes_per = algorithmic_bytes(n, m, int(k), p_owned, 4 if kdb > 0 else 8, k_tile)
        backward_stages = ("blend_backward", "reduce_partials", "point_backward")
        bytes_per = {k_: v for k_, v in bytes_per.items() if k_ in stages_ms}
        timed_stages = [k_ for k_ in stages_ms if not (args.forward_only and k_ in backward_stages)]
        # a stage whose (median) time exceeds the whole step was disturbed in this stage-by-stage profile (it is
        # launched between events, with blocking size reads the operator does not have): it cannot be the dominant kernel
        sane = [k_ for k_ in timed_stages if stages_ms[k_] <= ms_per_step] or timed_stages
        dominant = max(sane, key=stages_ms.get)
        achieved = bytes_per[dominant] / (stages_ms[dominant] * 1e-3) / 1e9
        path_bytes = sum(bytes_per[k_] for k_ in timed_stages)   # the stages inside the timed step
        # the same sum with SURVEY 8(d)'s literal accounting: K = the reference's binning (one key per (tile, Gaussian) of
        # the tile boxes, no exact cull, no bins = the backward's slot count), 8-byte keys, B = 266 N + 744 M + 136 K + 56 P
        k_ref = int(n_slots)
        survey_bytes = (18 * n + 368 * m + 92 * k_ref + 28 * p_owned) if args.forward_only else \
            (266 * n + 744 * m + 136 * k_ref + 56 * p_owned)
        # forward blend: entries a tile actually stages before every pixel of it has stopped (what the backward will walk)
        visited = int(tile_work.sum().item()) if op.ordered_dispatch else None
        # heavy-list scenes: the blend kernels leave a tile's list after a few entries (stress scene: 13 of 5,628), the
        # accounting charges them for whole lists -- the path figure is then not a bound on anything
        path_meaningful = k_tile <= 64 * max(m, 1)
        traffic, valu_instr = (None, None)
        if world == 1 and args.workload == "headline_1m_1080p":
            traffic, valu_instr = profiled_counters(dominant)
        # the HBM-class stage (streaming / sort / scan: everything but the two VALU-bound blend kernels) furthest from
        # its bound, among the stages that take at least 2 % of the step
        # (filter_compact and scan_block_sums are left out: in this stage-by-stage profile their times contain a blocking
        # size read-back that the operator's speculative launches do not have)
        hbm_class = [k_ for k_ in timed_stages
                     if k_ not in ("blend_forward", "blend_backward", "filter_compact", "scan_block_sums")
                     and stages_ms[k_] >= 0.02 * ms_per_step]
        worst = min(hbm_class, key=lambda k_: bytes_per[k_] / stages_ms[k_]) if hbm_class else None
        worst_traffic = profiled_counters({"sort_pairs": "sort_scatter"}.get(worst, worst))[0] \
            if (worst and world == 1 and args.workload == "headline_1m_1080p") else None
        valu = None
        if valu_instr is not None:
            rate = valu_instr / (stages_ms[dominant] * 1e-3)
            valu = {"wave_instructions_per_launch": valu_instr, "achieved_per_s": rate,
                    "peak_per_s": VALU_PEAK_WAVE_INSTR_PER_S, "frac": round(rate / VALU_PEAK_WAVE_INSTR_PER_S, 4),
                    "note": "peak = one plain fp32 wave64 instruction per 2 cycles per SIMD-32; packed, compare, DPP "
                            "and transcendental instructions issue at 1/2 .. 1/4 of it (profiles/r02_pmc_blend.md)"}
        roofline = {
            "bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
            "kernel_ms": round(stages_ms[dominant], 4),
            "algorithmic_bytes": int(bytes_per[dominant]),
            "valu": valu, "lane_utilisation": profiled_lane_utilisation(),
            "path": {"algorithmic_bytes": int(path_bytes),
                     "achieved": round(path_bytes / (ms_per_step * 1e-3) / 1e9, 2),
                     "frac": round(path_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                     "survey_8d_literal": {"K_reference_binning": k_ref, "algorithmic_bytes": int(survey_bytes),
                                           "achieved": round(survey_bytes / (ms_per_step * 1e-3) / 1e9, 2),
                                           "frac": round(survey_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}}
            if path_meaningful else None,
            "blend_forward_bytes": None if visited is None else {
                "whole_lists": int(48 * k_tile + 28 * p_owned), "visited_entries": visited,
                "visited": int(48 * visited + 28 * p_owned),
                "achieved_visited": round((48 * visited + 28 * p_owned) / (stages_ms["blend_forward"] * 1e-3) / 1e9, 2)},
            "hbm_stage_furthest_from_bound": None if worst is None else {
                "stage": worst, "stage_ms": round(stages_ms[worst], 4), "algorithmic_bytes": int(bytes_per[worst]),
                "achieved": round(bytes_per[worst] / (stages_ms[worst] * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(bytes_per[worst] / (stages_ms[worst] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "traffic_of_its_main_kernel": worst_traffic},
            "stages_ms": {k_: round(v, 4) for k_, v in stages_ms.items()},
            "kernel_source_hash": kernel_source_hash(),
            "note": "blend kernels are VALU-issue-bound by construction (DESIGN.md section 5)",
        }

    # ---------------------------------------------------------------- CPU baseline (rank 0, N == 1)
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.forward_only:
        from oracle import gs_oracle as O
        O.build()
        host_affinity.unpin_host_threads()   # the oracle runs on every core
        hs = host_scene
        t0 = time.perf_counter()
        f = O.forward(hs.point_cloud.numpy(), hs.point_cloud_features.numpy(), hs.point_invalid_mask.numpy(),
                      hs.point_object_id.numpy(), hs.camera_intrinsics.numpy(), hs.q_pointcloud_camera.numpy(),
                      hs.t_pointcloud_camera.numpy(), hs.height, hs.width, near_plane=hs.near_plane,
                      far_plane=hs.far_plane, depth_to_sort_key_scale=hs.depth_to_sort_key_scale)
        t1 = time.perf_counter()
        O.backward(f, grad_image.cpu().numpy(), 3)
        t2 = time.perf_counter()
        cpu_baseline = {
            "value": round(pixels / 1e6 / (t2 - t0), 4), "unit": "Mpixels/s", "cores": O.num_threads(),
            "kind": "port",
            "sample": f"1 full frame of the same workload ({args.workload}): forward {t1 - t0:.2f} s + "
                      f"backward {t2 - t1:.2f} s, OpenMP fp32 C oracle (tile-local accumulation, no atomics)",
            "host_cpus": os.cpu_count(),
        }

    if rank == 0:
        what = "fwd" if args.forward_only else "fwd+bwd"
        metric = (f"rendered Mpixels/s ({what}), 1e6 Gaussians @1920x1080" if args.workload == "headline_1m_1080p"
                  else f"rendered Mpixels/s ({what}), {args.workload}")
        out = {
            "metric": metric,
            "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "gaussians": n, "image": f"{s.width}x{s.height}",
                       "sh_degree": 3, "sharding": "none" if world == 1 else (
                           f"tile-row bands + owner-sharded Gaussians/{world}" if owner_mode else f"tile-row {args.shard_mode}/{world}"),
                       "owner_sharding": dict(module.last_frame_stats, magnitude_image=None) if owner_mode else None,
                       "ranks_seen_by_backend": dist.get_world_size() if world > 1 else 1,
                       "per_rank": per_rank,
                       "backend": (backend if world > 1 else None),
                       "backward_hook": hook is not None, "hook_feature_copy": bool(hook is not None and not args.no_hook_feature_copy),
                       "forward_only": args.forward_only, "training_like": not args.static_scene,
                       "rgb_only": bool(cfg.rgb_only), "speculation": dict(op.speculation_stats),
                       "host_threads_on_cpus": None if pinned is None else len(pinned),
                       "warmup_floor_ms": WARMUP_FLOOR_MS, "warmup_steps_run": warmup_steps_run[0], **sizes},
            # the literal protocol: the same K steps timed straight behind exactly W warm-up steps (no warm-up floor), i.e. partly on
            # the GPU's clock ramp -- reported beside the floored number (null: GS_BENCH_WARMUP_FLOOR_MS=0, ms_per_step is it)
            "ms_per_step_strict_warmup": round(strict_warmup_ms[0], 4) if strict_warmup_ms else None,
            "value_strict_warmup": round(pixels / 1e6 / (strict_warmup_ms[0] / 1e3), 3) if strict_warmup_ms else None,
            "step_ms": step_ms, "variants": variants,
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
