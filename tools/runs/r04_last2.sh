#!/bin/bash
# round 4, third session, after the correctly rounded scale activation (gs_exp_cr): the draw that found it, the GPU suite,
# the driver's bench line, and the profile passes retaken for the new kernel sources (tools/profile.sh)
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/last2_r04
mkdir -p $OUT
cd $ROOT
python tools/tile_count_probe.py 6102 2>&1 | grep -v amdgpu.ids | tail -8
GS_FUZZ_CASES=300 GS_FUZZ_FIRST=6000 timeout 120 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -k "6102 or 6100 or 6101 or 6103" 2>&1 | tail -2
timeout 480 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1
grep -E "passed|failed|error|\[record\]|preprocess.bits|preprocess.radius|preprocess.conic|preprocess.num_overlap" $OUT/pytest.log | tail -12
grep -E "^FAILED|^ERROR" $OUT/pytest.log | head -20
timeout 240 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/last2_r04/bench_default.json'))
r = d['roofline']
print('bench', d['ms_per_step'], d['value'], r['kernel'], r['frac'], 'traffic', r['traffic'], 'stages', d.get('stages_ms'))
PY
bash tools/profile.sh r04b > $OUT/profile.log 2>&1
find gpurun_out/prof_r04b -name "*kernel_trace.csv" -delete 2>/dev/null
head -8 gpurun_out/prof_r04b/trace/trace_kernel_stats.csv | cut -c1-120
du -sh gpurun_out/prof_r04b
