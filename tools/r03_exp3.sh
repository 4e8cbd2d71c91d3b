#!/bin/bash
# Round 3, GPU call 3: tests, stage A/B, bench lines, rocprofv3 profile of the default line, SQ counters of the backward
# arms, PSNR parity between back ends.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/exp3
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1
grep -E "passed|failed|error|\[record\]" $OUT/pytest.log | tail -8
for bs in 1 0; do
  echo "=== default build bin_shift=$bs"
  GS_BIN_SHIFT=$bs GS_TILE_ORDER=1 GS_AB=1 timeout 300 python tools/stage_bench.py headline_1m_1080p 20
done > $OUT/stage.log 2>&1
: > $OUT/bench.jsonl
timeout 600 python bench.py >> $OUT/bench.jsonl 2> $OUT/bench.err
timeout 300 python bench.py --static-scene --no-cpu-baseline >> $OUT/bench.jsonl 2>> $OUT/bench.err
for w in cfg1_10k_256 cfg2_100k_800 cfg3_400k_1080p cfg4_2m_1080p stress_t_ras; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline >> $OUT/bench.jsonl 2>> $OUT/bench.err
done
timeout 300 python bench.py --no-hook --no-cpu-baseline >> $OUT/bench.jsonl 2>> $OUT/bench.err
timeout 300 python bench.py --hook-feature-copy --no-cpu-baseline >> $OUT/bench.jsonl 2>> $OUT/bench.err
for w in headline_1m_1080p cfg3_400k_1080p stress_t_ras; do
  timeout 300 python bench.py --workload $w --forward-only --no-cpu-baseline >> $OUT/bench.jsonl 2>> $OUT/bench.err
  timeout 300 python bench.py --workload $w --forward-only --rgb-only --no-cpu-baseline >> $OUT/bench.jsonl 2>> $OUT/bench.err
done
bash tools/profile.sh r03 > $OUT/profile.log 2>&1
# SQ counters of the backward arms (stage_bench, binned lists as the operator uses them at this size)
cd /tmp && export TMPDIR=/tmp
for lib in r1 r2 r0 notrim; do
  GS_LIB_PATH=$ROOT/variants/libgsplat_hip_$lib.so GS_BIN_SHIFT=1 timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/sq_$lib -o sq -- python $ROOT/tools/stage_bench.py headline_1m_1080p 5 > $OUT/sq_$lib.log 2>&1
done
find $OUT $ROOT/gpurun_out/prof_r03 -name "*.db" -delete 2>/dev/null
cd $ROOT
python - <<'PY'
import glob, pandas as pd, re
for d in sorted(glob.glob("gpurun_out/exp3/sq_*/")):
    f = glob.glob(d + "*counter_collection.csv")
    if not f: print(d, "no counters"); continue
    t = pd.read_csv(f[0])
    t = t[t.Kernel_Name.str.contains("blend_backward")]
    t["dur_us"] = (t.End_Timestamp - t.Start_Timestamp) / 1e3
    p = t.pivot_table(index="Kernel_Name", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
    print(d, "dur_us", round(t.dur_us.mean(), 1)); print(p.round(0).to_string(header=True).replace("void (anonymous namespace)::", "")[:1500])
PY
timeout 1500 python tools/psnr_parity.py 2000 256 3 > $OUT/psnr_parity.log 2>&1
tail -40 $OUT/psnr_parity.log
grep -E "===|blend_|reduce|point_backward|identical|sum " $OUT/stage.log
python - <<'PY'
import json
for l in open("gpurun_out/exp3/bench.jsonl"):
    try: d=json.loads(l)
    except Exception: print("BAD", l[:200]); continue
    c=d["config"]; print(c["workload"], "fwd" if c["forward_only"] else "", "rgb" if c["rgb_only"] else "", "hook" if c["backward_hook"] else "nohook", "copy" if c["hook_feature_copy"] else "", "train" if c["training_like"] else "static", d["ms_per_step"], d["step_ms"], d["value"], d["roofline"]["stages_ms"] if d["roofline"] else None)
PY
