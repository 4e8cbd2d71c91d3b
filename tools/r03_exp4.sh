#!/bin/bash
# Round 3, GPU call 4: tests with the small-grid arms, blend arms at the headline size, small frames, shard times.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/exp4
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1
grep -E "passed|failed|error|\[record\]|four_waves" $OUT/pytest.log | tail -12
grep -B5 -A25 "Error\|FAILED" $OUT/pytest.log | head -80
for bs in 0 1; do
  echo "=== default build bin_shift=$bs"
  GS_BIN_SHIFT=$bs GS_TILE_ORDER=1 GS_ARMS=1 timeout 300 python tools/stage_bench.py headline_1m_1080p 20
done > $OUT/stage.log 2>&1
for w in cfg1_10k_256 cfg3_400k_1080p; do
  echo "=== default build workload=$w bin_shift=0"
  GS_BIN_SHIFT=0 GS_TILE_ORDER=1 GS_ARMS=1 timeout 300 python tools/stage_bench.py $w 20
done >> $OUT/stage.log 2>&1
: > $OUT/bench.jsonl
timeout 600 python bench.py --no-cpu-baseline >> $OUT/bench.jsonl 2> $OUT/bench.err
for w in cfg1_10k_256 cfg2_100k_800 stress_t_ras; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline >> $OUT/bench.jsonl 2>> $OUT/bench.err
done
for bs in "" 0 1; do
  echo "=== shard_bench headline GS_BIN_SHIFT=$bs"
  GS_BIN_SHIFT=$bs GS_SHARD_WORLDS=1,8 timeout 300 python tools/shard_bench.py headline_1m_1080p
done > $OUT/shard.log 2>&1
echo "=== shard_bench cfg4" >> $OUT/shard.log
GS_SHARD_WORLDS=1,8 timeout 300 python tools/shard_bench.py cfg4_2m_1080p >> $OUT/shard.log 2>&1
GS_BIN_SHIFT=0 GS_SHARD_WORLDS=8 timeout 300 python tools/shard_bench.py cfg4_2m_1080p >> $OUT/shard.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg1 -o trace -- python $ROOT/bench.py --workload cfg1_10k_256 --steps 20 --warmup 5 --no-cpu-baseline --no-stage-profile > $OUT/trace_cfg1.log 2>&1
find $OUT -name "*.db" -delete 2>/dev/null
cd $ROOT
grep -E "===|blend_|arm |identical|sum |tile_order" $OUT/stage.log
cat $OUT/shard.log | grep -v "amdgpu.ids"
python - <<'PY'
import json, csv
for l in open("gpurun_out/exp4/bench.jsonl"):
    try: d=json.loads(l)
    except Exception: print("BAD", l[:200]); continue
    c=d["config"]; print(c["workload"], d["ms_per_step"], d["step_ms"], d["value"], d["roofline"]["stages_ms"] if d["roofline"] else None)
for r in csv.DictReader(open("gpurun_out/exp4/trace_cfg1/trace_kernel_stats.csv")):
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={r['Percentage']}")
PY
