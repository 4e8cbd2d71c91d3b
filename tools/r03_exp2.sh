#!/bin/bash
# Round 3, GPU call 2: full GPU test suite with the new tests (observed values for the bars), ordered dispatch + fused
# slot reduction A/B, bench lines.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/exp2
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1
grep -E "passed|failed|error" $OUT/pytest.log | tail -5
for bs in 1 0; do
  echo "=== default build bin_shift=$bs"
  GS_BIN_SHIFT=$bs GS_TILE_ORDER=1 GS_AB=1 timeout 300 python tools/stage_bench.py headline_1m_1080p 20
done > $OUT/stage.log 2>&1
for w in cfg1_10k_256 cfg2_100k_800; do
  echo "=== default build workload=$w"
  GS_BIN_SHIFT=0 GS_TILE_ORDER=1 GS_AB=1 timeout 300 python tools/stage_bench.py $w 20
done >> $OUT/stage.log 2>&1
: > $OUT/bench.jsonl
timeout 600 python bench.py >> $OUT/bench.jsonl 2> $OUT/bench.err
timeout 300 python bench.py --static-scene --no-cpu-baseline >> $OUT/bench.jsonl 2>> $OUT/bench.err
for w in cfg1_10k_256 cfg2_100k_800 cfg3_400k_1080p cfg4_2m_1080p stress_t_ras; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline >> $OUT/bench.jsonl 2>> $OUT/bench.err
done
timeout 300 python bench.py --forward-only --no-cpu-baseline >> $OUT/bench.jsonl 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/trace.log 2>&1
find $OUT -name "*.db" -delete 2>/dev/null
cd $ROOT
grep -E "===|blend_|reduce|point_backward|identical|sum " $OUT/stage.log
python - <<'PY'
import json
for l in open("gpurun_out/exp2/bench.jsonl"):
    try: d=json.loads(l)
    except Exception: print("BAD", l[:200]); continue
    print(d["config"]["workload"], "fwd_only" if d["config"]["forward_only"] else "", d["ms_per_step"], d["step_ms"], d["value"], d["roofline"]["stages_ms"] if d["roofline"] else None)
PY
head -30 $OUT/trace/*kernel_stats.csv 2>/dev/null | cut -c1-150
grep "\[parity\] \(needles\|chains\|stress_small\)" $OUT/pytest.log | cut -c1-400
