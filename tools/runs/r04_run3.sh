#!/bin/bash
# round 4, GPU call 3: sweep sort with direct aggregate sums -- correctness + per-kernel times per tile size
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run3; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "radix_sort or keys_sort_ranges" > $OUT/pytest_sort.log 2>&1; tail -2 $OUT/pytest_sort.log
cd /tmp && export TMPDIR=/tmp
for cfg in "GS_X=0" "GS_SWEEP_ROUNDS=2" "GS_SWEEP_ROUNDS=4" "GS_SWEEP_ROUNDS=8"; do
  tag=$(echo $cfg | tr '=' '_')
  env $cfg timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$tag -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/sort_bench.py 10 > $OUT/log_$tag.txt 2>&1
  grep sort_bench $OUT/log_$tag.txt
done
cd $GRAFT_REPO_ROOT && python tools/trace_by_grid.py $OUT/prof_GS_X_0 $OUT/prof_GS_SWEEP_ROUNDS_2 $OUT/prof_GS_SWEEP_ROUNDS_4 $OUT/prof_GS_SWEEP_ROUNDS_8
