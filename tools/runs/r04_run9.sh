#!/bin/bash
# round 4, GPU call 9: fused pose inverse, sizes through pinned memory, cooperative SH reads -- full suite, benches, owner ranks
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run9; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
for w in cfg1_10k_256 headline_1m_1080p; do
  timeout 600 python bench.py --no-cpu-baseline --workload $w > $OUT/bench_$w.json 2> $OUT/bench_$w.err; python -c "
import json; d=json.load(open('$OUT/bench_$w.json')); r=d['roofline']; print('$w', d['ms_per_step'], d['value'], d['step_ms'], d['variants'], r and r['stages_ms'])"
done
GS_HOST_MIRROR=0 timeout 600 python bench.py --no-cpu-baseline --workload cfg1_10k_256 --no-stage-profile > $OUT/bench_cfg1_nomirror.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/bench_cfg1_nomirror.json')); print('cfg1 no host mirror', d['ms_per_step'], d['step_ms'])"
GS_SHARD_WORLDS=8 timeout 600 python tools/owner_shard_bench.py headline_1m_1080p 2>&1 | grep owner_shard
