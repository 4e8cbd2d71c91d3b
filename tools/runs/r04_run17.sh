#!/bin/bash
# round 4, GPU call 17: the bucket-local sort writes the bins' ranges (no tile_ranges launch on MSD-first frames)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run17; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "radix_sort or keys_sort_ranges or one_entry_point or speculative" > $OUT/pytest_sort.log 2>&1; tail -4 $OUT/pytest_sort.log
for arm in 1 0; do
for w in headline_1m_1080p cfg1_10k_256 trained_1080p; do
  GS_SORT_RANGES=$arm timeout 600 python bench.py --no-cpu-baseline --no-stage-profile --workload $w > $OUT/bench_${w}_ranges$arm.json 2> $OUT/bench_${w}_ranges$arm.err
  python -c "
import json; d=json.load(open('$OUT/bench_${w}_ranges$arm.json')); print('$w sort writes ranges=$arm', d['ms_per_step'], d['value'], d['step_ms'], d['variants']['hook_without_feature_copy']['ms_per_step'])"
done; done
timeout 2400 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; grep -n "passed\|failed" $OUT/pytest.log | tail -3
