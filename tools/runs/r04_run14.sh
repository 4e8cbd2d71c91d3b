#!/bin/bash
# round 4, GPU call 14: MSD-first sort (one scatter pass on the top eight bits + bucket-local LSD passes in LDS) --
# sort tests, sort_bench in both arms (GS_SORT_IMPL=lsd = three launches per eight bits), bench lines in both arms
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run14; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "radix_sort or keys_sort_ranges" > $OUT/pytest_sort.log 2>&1; tail -4 $OUT/pytest_sort.log
timeout 300 python tools/sort_bench.py 20 2>&1 | grep sort_bench | tee $OUT/sort_bench_msd.txt
GS_SORT_IMPL=lsd timeout 300 python tools/sort_bench.py 20 2>&1 | grep sort_bench | tee $OUT/sort_bench_lsd.txt
for arm in msd lsd; do
for w in headline_1m_1080p trained_1080p cfg1_10k_256 cfg2_100k_800; do
  GS_SORT_IMPL=$arm timeout 600 python bench.py --no-cpu-baseline --workload $w > $OUT/bench_${w}_$arm.json 2> $OUT/bench_${w}_$arm.err
  python -c "
import json; d=json.load(open('$OUT/bench_${w}_$arm.json')); r=d['roofline']; print('$w sort=$arm', d['ms_per_step'], d['value'], d['step_ms'], r and r['stages_ms'].get('sort_pairs'))"
done; done
