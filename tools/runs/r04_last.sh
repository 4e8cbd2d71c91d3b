#!/bin/bash
# round 4, last call (third session: 13 GPU-minutes were left): the GPU suite on the final tree (with the two new
# reference-run vectors i and j), smoke, the driver's bench line, a kernel trace of the same command.
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/last_r04
mkdir -p $OUT
cd $ROOT
timeout 480 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1
grep -E "passed|failed|error|\[record\]|reference_vector" $OUT/pytest.log | tail -20
grep -E "^FAILED|^ERROR" $OUT/pytest.log | head -20
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 240 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/last_r04/bench_default.json'))
r = d['roofline']
print('bench', d['ms_per_step'], d['value'], r['kernel'], r['frac'], 'traffic', r['traffic'], 'valu', r['valu'] and r['valu']['frac'],
      r['hbm_stage_furthest_from_bound'], d.get('variants'))
print(d['cpu_baseline'])
PY
(cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- \
    python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/trace.log 2>&1)
find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -name "*kernel_trace.csv" -delete 2>/dev/null
head -15 $OUT/trace/trace_kernel_stats.csv 2>/dev/null | cut -c1-160
