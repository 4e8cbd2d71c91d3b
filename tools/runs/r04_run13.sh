#!/bin/bash
# round 4, GPU call 13 (second session): colours on a second stream beside the list stages -- new tests, A/B bench
# lines (GS_COLOUR_ASYNC=0 keeps the colours inside gs_preprocess), then the whole GPU suite at this tree
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run13; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "colours or one_entry_point or test_preprocess or options_that" > $OUT/pytest_colours.log 2>&1; tail -3 $OUT/pytest_colours.log
for rep in 1 2; do
for arm in 1 0; do
for w in headline_1m_1080p trained_1080p cfg1_10k_256; do
  [ $rep = 2 ] && [ $w != headline_1m_1080p ] && continue
  GS_COLOUR_ASYNC=$arm timeout 600 python bench.py --no-cpu-baseline --no-stage-profile --workload $w > $OUT/bench_${w}_async${arm}_$rep.json 2> $OUT/bench_${w}_async${arm}_$rep.err
  python -c "
import json; d=json.load(open('$OUT/bench_${w}_async${arm}_$rep.json')); print('$w async=$arm rep $rep', d['ms_per_step'], d['value'], d['step_ms'], d.get('variants'))"
done; done; done
timeout 2400 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
grep "^\[parity\]\|^\.\[parity\]\|\[record\]" $OUT/pytest.log > $OUT/report_lines.txt
