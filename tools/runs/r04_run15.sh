#!/bin/bash
# round 4, GPU call 15: MSD-first sort with an 8 / 9 / 10-bit partitioning digit chosen by the capacity
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run15; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "radix_sort or keys_sort_ranges" > $OUT/pytest_sort.log 2>&1; tail -4 $OUT/pytest_sort.log
timeout 300 python tools/sort_bench.py 20 2>&1 | grep sort_bench | tee $OUT/sort_bench_default.txt
for b in 8 9 10; do GS_SORT_MSD_BITS=$b timeout 300 python tools/sort_bench.py 20 2>&1 | grep sort_bench | sed "s/default/msd_bits=$b/" | tee $OUT/sort_bench_bits$b.txt; done
for arm in msd lsd; do
for w in headline_1m_1080p cfg3_400k_1080p cfg4_2m_1080p trained_1080p cfg1_10k_256; do
  GS_SORT_IMPL=$arm timeout 600 python bench.py --no-cpu-baseline --workload $w > $OUT/bench_${w}_$arm.json 2> $OUT/bench_${w}_$arm.err
  python -c "
import json; d=json.load(open('$OUT/bench_${w}_$arm.json')); r=d['roofline']; print('$w sort=$arm', d['ms_per_step'], d['value'], d['step_ms'], r and r['stages_ms'].get('sort_pairs'))"
done; done
