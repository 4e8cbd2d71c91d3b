#!/bin/bash
# round 4, GPU call 5: one entry point per pass -- identity tests, full suite, host-bound step times with / without
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run5; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "one_entry_point or options_that_must or speculative" > $OUT/pytest_frame.log 2>&1; tail -8 $OUT/pytest_frame.log
for w in cfg1_10k_256 cfg2_100k_800 headline_1m_1080p; do
  for fe in 1 0; do GS_NO_CPROFILE=1 GS_FRAME_ENTRY_POINTS=$fe timeout 300 python tools/host_profile.py $w 200 2>&1 | grep host_profile; done
done
GS_FRAME_ENTRY_POINTS=1 timeout 300 python tools/host_profile.py cfg1_10k_256 300 > $OUT/host_profile_cfg1.log 2>&1; head -45 $OUT/host_profile_cfg1.log | tail -40
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench.json')); r=d['roofline']; print(d['ms_per_step'], d['value'], d['step_ms'], d['variants'])"
timeout 600 python bench.py --no-cpu-baseline --workload cfg1_10k_256 > $OUT/bench_cfg1.json 2> $OUT/bench_cfg1.err; python -c "
import json; d=json.load(open('$OUT/bench_cfg1.json')); print('cfg1', d['ms_per_step'], d['value'], d['step_ms'])"
