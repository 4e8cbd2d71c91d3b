#!/bin/bash
# round 4, GPU call 1: the single-sweep sort -- correctness, then timings against the three-launch sort
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_run1; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "radix_sort or keys_sort_ranges" > $OUT/pytest_sort.log 2>&1; tail -3 $OUT/pytest_sort.log
for cfg in "" "GS_SWEEP_TICKET=0" "GS_SORT_IMPL=lsd3" "GS_SWEEP_ROUNDS=4" "GS_SWEEP_ROUNDS=8" "GS_SWEEP_ROUNDS=11" "GS_SWEEP_ROUNDS=2" "GS_SWEEP_ROUNDS=1"; do
    env $cfg timeout 300 python tools/sort_bench.py 20 2>&1 | grep sort_bench
done > $OUT/sort_bench.log
cat $OUT/sort_bench.log
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench.json')); r=d['roofline']; print(d['ms_per_step'], d['value'], d['variants'], r['kernel'], r['frac'], r['stages_ms'], r['path'], r['blend_forward_bytes'])"
GS_SORT_IMPL=lsd3 timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_lsd3.json 2> $OUT/bench_lsd3.err; python -c "
import json; d=json.load(open('$OUT/bench_lsd3.json')); r=d['roofline']; print('lsd3', d['ms_per_step'], d['value'], r['stages_ms'])"
