#!/bin/bash
# The bench line of every workload the documents quote (section 2 of round_report.sh on its own).  usage: tools/bench_lines.sh <tag>
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/lines_$TAG
mkdir -p $OUT
cd $ROOT
: > $OUT/bench_all_configs.jsonl
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cat $OUT/bench_default.json >> $OUT/bench_all_configs.jsonl
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2>> $OUT/bench_default.err
for w in cfg1_10k_256 cfg2_100k_800 cfg3_400k_1080p cfg4_2m_1080p stress_t_ras trained_1080p; do
    python bench.py --workload $w --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
done
python bench.py --static-scene --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
python bench.py --no-hook --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
python bench.py --no-hook-feature-copy --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
for w in headline_1m_1080p cfg3_400k_1080p stress_t_ras trained_1080p; do
    python bench.py --workload $w --forward-only --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
    python bench.py --workload $w --forward-only --rgb-only --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
done
python - <<PY
import json
for l in open("$OUT/bench_all_configs.jsonl"):
    d = json.loads(l); c = d["config"]
    print(c.get("workload"), d["ms_per_step"], d.get("ms_per_step_strict_warmup"), "hook" if c.get("backward_hook") else "", "copy" if c.get("hook_feature_copy") else "",
          "fwd" if c.get("forward_only") else "", "rgb" if c.get("rgb_only") else "", "static" if c.get("static_scene") else "")
d = json.load(open("$OUT/bench_driver_command.json")); print("driver command", d["ms_per_step"], d.get("ms_per_step_strict_warmup"))
PY
