#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/exp7
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1
grep -E "passed|failed|error|\[record\]" $OUT/pytest.log | tail -6
grep -E "^(FAILED|ERROR)" $OUT/pytest.log | head
grep -B3 -A25 "Error" $OUT/pytest.log | head -120
GS_BIN_SHIFT=1 GS_TILE_ORDER=1 timeout 300 python tools/stage_bench.py headline_1m_1080p 20 > $OUT/stage.log 2>&1
grep -E "workload|blend_|identical" $OUT/stage.log
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python -c "import json; d=json.load(open('$OUT/bench.json')); print(d['ms_per_step'], d['step_ms'], d['value'], d['roofline']['stages_ms'])"
for w in cfg3_400k_1080p cfg4_2m_1080p stress_t_ras; do timeout 300 python bench.py --workload $w --no-cpu-baseline 2>> $OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['ms_per_step'], d['step_ms'], d['roofline']['stages_ms'])"; done
