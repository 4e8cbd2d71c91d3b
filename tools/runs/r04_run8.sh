#!/bin/bash
# round 4, GPU call 8: full suite with list splitting + frame entry points; bench lines
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run8; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
for w in cfg1_10k_256 cfg2_100k_800 headline_1m_1080p; do
  timeout 600 python bench.py --no-cpu-baseline --workload $w > $OUT/bench_$w.json 2> $OUT/bench_$w.err; python -c "
import json; d=json.load(open('$OUT/bench_$w.json')); print('$w', d['ms_per_step'], d['value'], d['step_ms'], d['variants'])"
done
