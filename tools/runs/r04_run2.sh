#!/bin/bash
# round 4, GPU call 2: kernel trace of the sort benchmark (where does a sweep pass spend its time?)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "GS_X=0" "GS_SWEEP_ROUNDS=4" "GS_SORT_IMPL=lsd3"; do
  tag=$(echo $cfg | tr '=' '_')
  env $cfg timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$tag -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/sort_bench.py 10 > $OUT/log_$tag.txt 2>&1
  f=$(find $OUT/prof_$tag -name "*kernel_stats.csv" | head -1); echo "== $cfg"; cat $f | cut -d, -f1-4,6,7 | head -12
done
