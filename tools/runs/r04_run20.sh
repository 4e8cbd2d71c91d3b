#!/bin/bash
# round 4, GPU call 20: the list-layout choice (per-tile keys against 2x2-tile bins) now that the sort is cheaper
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run20; mkdir -p $OUT
for w in headline_1m_1080p cfg3_400k_1080p cfg4_2m_1080p trained_1080p cfg2_100k_800; do
for bs in 0 1; do
  timeout 600 python bench.py --no-cpu-baseline --no-stage-profile --bin-shift $bs --workload $w > $OUT/bench_${w}_bs$bs.json 2> $OUT/bench_${w}_bs$bs.err
  python -c "
import json; d=json.load(open('$OUT/bench_${w}_bs$bs.json')); print('$w bin_shift=$bs', d['ms_per_step'], d['step_ms']['median'], d['variants']['hook_without_feature_copy']['ms_per_step'])"
done; done
for bs in 0 1 2; do GS_BIN_SHIFT=$bs GS_SHARD_WORLDS=8 timeout 300 python tools/owner_shard_bench.py headline_1m_1080p 2>&1 | grep owner_shard | cut -c1-330; done | tee $OUT/owner_g8_bins.txt
