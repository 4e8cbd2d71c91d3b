#!/bin/bash
# Everything the round's documents quote, in one gpurun call (see profiles/README.md).  usage: tools/round_report.sh <tag>
set -u
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/report_$TAG
mkdir -p $OUT
cd $ROOT
# 0. the GPU test suite (with the [parity] / [record] lines the documents quote)
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1
grep -E "passed|failed|\[record\]" $OUT/pytest.log | tail -5
# 1. rocprofv3: kernel trace + HBM counters + SQ counters over the bench command
bash tools/profile.sh $TAG > $OUT/profile.log 2>&1
# 1b. path statistics of the blend kernels (a counting build of the library: how often the hit path runs, live lanes, how
#     often a decision is settled by the reference's own expression) at the headline size and on the trained scene
bash tools/build_variants.sh "stats:-DGS_TUNING_BUILD=1 -DGS_STATS=1" > $OUT/build_stats.log 2>&1
GS_LIB_PATH=variants/libgsplat_hip_stats.so python tools/blend_stats.py headline_1m_1080p > $OUT/blend_path_stats.json 2> $OUT/blend_stats.err
for w in cfg3_400k_1080p cfg4_2m_1080p trained_1080p; do
    GS_LIB_PATH=variants/libgsplat_hip_stats.so python tools/blend_stats.py $w 2>> $OUT/blend_stats.err >> $OUT/blend_path_stats_other_workloads.jsonl
done
# 2. bench lines of every workload (driver contract line first)
: > $OUT/bench_all_configs.jsonl
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cat $OUT/bench_default.json >> $OUT/bench_all_configs.jsonl
for w in cfg1_10k_256 cfg2_100k_800 cfg3_400k_1080p cfg4_2m_1080p stress_t_ras trained_1080p; do
    python bench.py --workload $w --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
done
python bench.py --static-scene --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
python bench.py --no-hook --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
python bench.py --no-hook-feature-copy --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
for w in headline_1m_1080p cfg3_400k_1080p stress_t_ras trained_1080p; do
    python bench.py --workload $w --forward-only --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
    python bench.py --workload $w --forward-only --rgb-only --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
done
# 3. per-stage times and per-rank shard times (one GPU, no collectives)
for bs in 1 0; do GS_BIN_SHIFT=$bs GS_TILE_ORDER=1 GS_ARMS=1 GS_AB=1 python tools/stage_bench.py headline_1m_1080p 30; done > $OUT/stage_headline.log 2>&1
GS_SHARD_EXCHANGE=1 GS_SHARD_MODE=bands python tools/shard_bench.py headline_1m_1080p > $OUT/shard_headline.log 2>&1
# owner-sharded Gaussians: every rank's phases at G = 1, 2, 4, 8 (all ranks played in lockstep on this GPU)
python tools/owner_shard_bench.py headline_1m_1080p > $OUT/owner_shard_headline.log 2>&1
GS_SHARD_WORLDS=8 python tools/owner_shard_bench.py cfg4_2m_1080p > $OUT/owner_shard_cfg4.log 2>&1
GS_SHARD_WORLDS=8 python tools/owner_shard_bench.py trained_1080p > $OUT/owner_shard_trained.log 2>&1
# kernel trace of the eight ranks of an owner-sharded frame (what a rank's time is made of: divide the totals by 8 ranks)
(cd /tmp && export TMPDIR=/tmp && cd $ROOT && GS_SHARD_WORLDS=8 GS_SHARD_REPS=10 rocprofv3 --kernel-trace --stats -d $OUT/owner_g8_prof -o g8 --output-format csv -- \
    python tools/owner_shard_bench.py headline_1m_1080p) > $OUT/owner_g8_trace.log 2>&1
cp $OUT/owner_g8_prof/g8_kernel_stats.csv $OUT/owner_g8_kernel_stats.csv 2>/dev/null
python tools/host_profile.py cfg1_10k_256 300 > $OUT/host_profile_cfg1.log 2>&1
for fe in 1 0; do for sp in 1 0; do GS_NO_CPROFILE=1 GS_FRAME_ENTRY_POINTS=$fe GS_SPLIT=$sp python tools/host_profile.py cfg1_10k_256 300 2>&1 | grep host_profile; done; done > $OUT/cfg1_arms.log
# the reference's forward-only benchmark protocol on the trained scene (benchmark/inference_benchmark.py, BENCH:109-160)
python benchmark/inference_benchmark.py --synthetic trained_1080p --warmup 200 --iterations 100 > $OUT/inference_trained.log 2>&1
python benchmark/inference_benchmark.py --synthetic headline_1m_1080p --warmup 200 --iterations 100 > $OUT/inference_headline.log 2>&1
python tools/trained_scene_probe.py > $OUT/trained_scene.log 2>&1
# 4. multi-rank bench plumbing on this one GPU (gloo transport)
GS_BENCH_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_2ranks_gloo_one_gpu.json 2> $OUT/bench_2ranks.err
GS_BENCH_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --shard-mode bands > $OUT/bench_2ranks_gloo_one_gpu_bands.json 2>> $OUT/bench_2ranks.err
# BASELINE config 5, fallback form: 7,000 iterations of the trainer (and the first 300 with the CPU oracle as the rasteriser)
python tools/train_7k.py 7000 300 800 > $OUT/train7k.log 2>&1; cp gpurun_out/train7k/summary.json $OUT/train7k_summary.json 2>/dev/null
ls -la $OUT
