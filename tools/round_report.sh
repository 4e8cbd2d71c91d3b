#!/bin/bash
# Everything the round's documents quote, in one gpurun call (see profiles/README.md).  usage: tools/round_report.sh <tag>
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/report_$TAG
mkdir -p $OUT
cd $ROOT
# 0. the GPU test suite (with the [parity] / [record] lines the documents quote)
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1
grep -E "passed|failed|\[record\]" $OUT/pytest.log | tail -5
# 1. rocprofv3: kernel trace + HBM counters + SQ counters over the bench command
bash tools/profile.sh $TAG > $OUT/profile.log 2>&1
# 2. bench lines of every workload (driver contract line first)
: > $OUT/bench_all_configs.jsonl
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cat $OUT/bench_default.json >> $OUT/bench_all_configs.jsonl
for w in cfg1_10k_256 cfg2_100k_800 cfg3_400k_1080p cfg4_2m_1080p stress_t_ras; do
    python bench.py --workload $w --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
done
python bench.py --static-scene --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
python bench.py --no-hook --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
python bench.py --hook-feature-copy --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
for w in headline_1m_1080p cfg3_400k_1080p stress_t_ras; do
    python bench.py --workload $w --forward-only --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
    python bench.py --workload $w --forward-only --rgb-only --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
done
# 3. per-stage times and per-rank shard times (one GPU, no collectives)
for bs in 1 0; do GS_BIN_SHIFT=$bs GS_TILE_ORDER=1 GS_ARMS=1 GS_AB=1 python tools/stage_bench.py headline_1m_1080p 30; done > $OUT/stage_headline.log 2>&1
for m in bands interleaved; do GS_SHARD_EXCHANGE=1 GS_SHARD_MODE=$m python tools/shard_bench.py headline_1m_1080p; done > $OUT/shard_headline.log 2>&1
GS_SHARD_EXCHANGE=1 python tools/shard_bench.py cfg4_2m_1080p > $OUT/shard_cfg4.log 2>&1
# kernel trace of the middle rank of eight (what a rank's time is made of)
(cd /tmp && export TMPDIR=/tmp && cd $ROOT && GS_SHARD_WORLDS=8 rocprofv3 --kernel-trace --stats -d $OUT/shard_g8_prof -o g8 --output-format csv -- \
    python tools/shard_bench.py headline_1m_1080p) > $OUT/shard_g8_trace.log 2>&1
cp $OUT/shard_g8_prof/g8_kernel_stats.csv $OUT/shard_g8_kernel_stats.csv 2>/dev/null
python tools/host_profile.py cfg1_10k_256 300 > $OUT/host_profile_cfg1.log 2>&1
# 4. multi-rank bench plumbing on this one GPU (gloo transport)
GS_BENCH_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_2ranks_gloo_one_gpu.json 2> $OUT/bench_2ranks.err
ls -la $OUT
