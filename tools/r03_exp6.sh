#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/exp6
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1
grep -E "passed|failed|error|\[record\]" $OUT/pytest.log | tail -6
grep -E "^(FAILED|ERROR)" $OUT/pytest.log | head
for bs in 1 0; do GS_BIN_SHIFT=$bs GS_TILE_ORDER=1 timeout 300 python tools/stage_bench.py headline_1m_1080p 20; done > $OUT/stage.log 2>&1
grep -E "workload|blend_|identical" $OUT/stage.log
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python -c "import json; d=json.load(open('$OUT/bench.json')); print(d['ms_per_step'], d['step_ms'], d['value'], d['roofline']['stages_ms'])"
timeout 300 python tools/heavy_gaussian_check.py > $OUT/heavy.log 2>&1; tail -6 $OUT/heavy.log
timeout 300 python tools/pathological_inputs_check.py > $OUT/patho.log 2>&1; tail -12 $OUT/patho.log
timeout 600 python tools/train_7k.py 7000 0 800 > $OUT/train7k.log 2>&1; tail -30 $OUT/train7k.log | head -60
