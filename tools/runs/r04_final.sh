#!/bin/bash
# round 4, final report call: tools/round_report.sh without the CPU-oracle training leg (its numbers stand from the first
# report of the round: the rasteriser's results are unchanged, the oracle run costs six GPU-box minutes), plus the
# owner-sharded ranks with balanced bands and the fuzz suite on fresh seeds
set -u
TAG=r04
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/report_$TAG
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1
grep -E "passed|failed|\[record\]" $OUT/pytest.log | tail -5
bash tools/profile.sh $TAG > $OUT/profile.log 2>&1
: > $OUT/bench_all_configs.jsonl
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cat $OUT/bench_default.json >> $OUT/bench_all_configs.jsonl
for w in cfg1_10k_256 cfg2_100k_800 cfg3_400k_1080p cfg4_2m_1080p stress_t_ras trained_1080p; do
    python bench.py --workload $w --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
done
python bench.py --static-scene --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
python bench.py --no-hook --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
python bench.py --no-hook-feature-copy --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
GS_SORT_IMPL=lsd python bench.py --no-cpu-baseline 2>> $OUT/bench.err > $OUT/bench_headline_lsd_sort.json
for w in headline_1m_1080p cfg3_400k_1080p stress_t_ras trained_1080p; do
    python bench.py --workload $w --forward-only --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
    python bench.py --workload $w --forward-only --rgb-only --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
done
for bs in 1 0; do GS_BIN_SHIFT=$bs GS_TILE_ORDER=1 GS_ARMS=1 GS_AB=1 python tools/stage_bench.py headline_1m_1080p 30; done > $OUT/stage_headline.log 2>&1
GS_SHARD_EXCHANGE=1 GS_SHARD_MODE=bands python tools/shard_bench.py headline_1m_1080p > $OUT/shard_headline.log 2>&1
python tools/owner_shard_bench.py headline_1m_1080p > $OUT/owner_shard_headline.log 2>&1
GS_SHARD_WORLDS=8 python tools/owner_shard_bench.py cfg4_2m_1080p > $OUT/owner_shard_cfg4.log 2>&1
GS_SHARD_WORLDS=8 python tools/owner_shard_bench.py trained_1080p > $OUT/owner_shard_trained.log 2>&1
GS_SHARD_BALANCE=1 GS_SHARD_WORLDS=8 python tools/owner_shard_bench.py trained_1080p > $OUT/owner_shard_trained_balanced.log 2>&1
(cd /tmp && export TMPDIR=/tmp && cd $ROOT && GS_SHARD_WORLDS=8 GS_SHARD_REPS=10 rocprofv3 --kernel-trace --stats -d $OUT/owner_g8_prof -o g8 --output-format csv -- \
    python tools/owner_shard_bench.py headline_1m_1080p) > $OUT/owner_g8_trace.log 2>&1
cp $OUT/owner_g8_prof/g8_kernel_stats.csv $OUT/owner_g8_kernel_stats.csv 2>/dev/null
find $OUT/owner_g8_prof -name "*kernel_trace.csv" -delete 2>/dev/null; find $OUT -name "*.db" -delete 2>/dev/null
python tools/host_profile.py cfg1_10k_256 300 > $OUT/host_profile_cfg1.log 2>&1
for fe in 1 0; do for sp in 1 0; do GS_NO_CPROFILE=1 GS_FRAME_ENTRY_POINTS=$fe GS_SPLIT=$sp python tools/host_profile.py cfg1_10k_256 300 2>&1 | grep host_profile; done; done > $OUT/cfg1_arms.log
python benchmark/inference_benchmark.py --synthetic trained_1080p --warmup 200 --iterations 100 > $OUT/inference_trained.log 2>&1
python benchmark/inference_benchmark.py --synthetic headline_1m_1080p --warmup 200 --iterations 100 > $OUT/inference_headline.log 2>&1
python tools/trained_scene_probe.py > $OUT/trained_scene.log 2>&1
GS_BENCH_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_2ranks_gloo_one_gpu.json 2> $OUT/bench_2ranks.err
GS_BENCH_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --shard-mode bands > $OUT/bench_2ranks_gloo_one_gpu_bands.json 2>> $OUT/bench_2ranks.err
python tools/train_7k.py 7000 0 800 > $OUT/train7k.log 2>&1; cp gpurun_out/train7k/summary.json $OUT/train7k_summary.json 2>/dev/null
GS_FUZZ_CASES=200 GS_FUZZ_FIRST=3000 timeout 600 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -x > $OUT/fuzz_fresh_seeds.log 2>&1; grep -E "passed|failed" $OUT/fuzz_fresh_seeds.log | tail -2
du -sh $OUT; ls $OUT | head -60
