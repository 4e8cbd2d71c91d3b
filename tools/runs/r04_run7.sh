#!/bin/bash
# round 4, GPU call 7: list splitting (tests + what it buys), trained-scene test, full suite
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run7; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -s -k "split_backward or four_waves or blend_backward or one_entry_point or operator_end_to_end" > $OUT/pytest_split.log 2>&1; grep -E "passed|failed|split_backward|Error" $OUT/pytest_split.log | tail -12
for sp in 0 1; do
  for w in cfg1_10k_256 cfg2_100k_800; do
    GS_NO_CPROFILE=1 GS_SPLIT=$sp timeout 300 python tools/host_profile.py $w 200 2>&1 | grep host_profile
  done
done
timeout 600 python tools/owner_shard_bench.py headline_1m_1080p > $OUT/owner_headline.log 2>&1; grep owner_shard $OUT/owner_headline.log || tail -20 $OUT/owner_headline.log
timeout 1200 python -m pytest tests/test_trained_scene_gpu.py -m gpu -q -x -s > $OUT/pytest_trained.log 2>&1; grep -E "passed|failed|trained_scene|Error|assert" $OUT/pytest_trained.log | tail -14
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_trained_scene_gpu.py > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
