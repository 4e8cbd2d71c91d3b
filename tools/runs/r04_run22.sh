#!/bin/bash
# round 4, GPU call 22: bucket-local sort ranks through an LDS OR registry instead of ballots (six-bit passes)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run22; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "radix_sort or keys_sort_ranges or one_entry_point" > $OUT/pytest_sort.log 2>&1; grep -n "passed\|failed" $OUT/pytest_sort.log | tail -2
GS_SORT_GROUPED=1 timeout 300 python tools/sort_bench.py 20 2>&1 | grep sort_bench | tee $OUT/sort_bench_grouped.txt
timeout 300 python tools/sort_bench.py 20 2>&1 | grep sort_bench | tee $OUT/sort_bench_ascending.txt
for w in headline_1m_1080p trained_1080p cfg1_10k_256 cfg4_2m_1080p; do
  timeout 600 python bench.py --no-cpu-baseline --workload $w > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python -c "
import json; d=json.load(open('$OUT/bench_$w.json')); r=d['roofline']; print('$w', d['ms_per_step'], d['step_ms'], d['variants']['hook_without_feature_copy']['ms_per_step'], r and r['stages_ms'].get('sort_pairs'))"
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stage-profile > $OUT/trace.log 2>&1
find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -name "*kernel_trace.csv" -delete
python - <<PY
import csv,re
for r in csv.DictReader(open("$OUT/trace/trace_kernel_stats.csv")):
    m=re.search(r'(\w+_kernel)',r['Name'])
    if m and 'sort' in m.group(1): print(m.group(1), r['Calls'], float(r['AverageNs'])/1000)
PY
