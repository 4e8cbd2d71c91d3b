#!/bin/bash
# round 4, last report call: GPU suite, rocprofv3 passes and the bench lines again after the bucket-local sort's LDS
# registry (the kernel sources changed once more after tools/runs/r04_final.sh; everything else of that report stands)
set -u
TAG=r04
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/report_${TAG}b
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
GS_SORT_GROUPED=1 python tools/sort_bench.py 20 2>&1 | grep sort_bench > $OUT/sort_bench_grouped.txt
ls $OUT
