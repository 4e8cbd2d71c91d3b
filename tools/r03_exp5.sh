#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/exp5
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1
grep -E "passed|failed|error|\[record\]" $OUT/pytest.log | tail -12
grep -E "^(FAILED|ERROR)" $OUT/pytest.log | head
GS_BIN_SHIFT=0 GS_TILE_ORDER=1 GS_ARMS=1 timeout 300 python tools/stage_bench.py cfg1_10k_256 20 > $OUT/stage.log 2>&1
grep -E "blend_|arm |identical" $OUT/stage.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-stage-profile > $OUT/trace.log 2>&1
find $OUT -name "*.db" -delete 2>/dev/null
cd $ROOT
grep '^{"metric' $OUT/trace.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms'])"
python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/exp5/trace/trace_kernel_stats.csv")):
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={r['Percentage']}")
PY
