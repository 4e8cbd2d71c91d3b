#!/bin/bash
# round 4, GPU call 6: owner-sharded phases as single foreign calls + speculative band sizes; overflow fallback fix
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run6; mkdir -p $OUT
timeout 900 python -m pytest tests/test_owner_sharding_gpu.py tests/test_hip_parity.py -m gpu -q -x -k "owner or one_entry_point or speculative" > $OUT/pytest_a.log 2>&1; tail -12 $OUT/pytest_a.log
timeout 600 python tools/owner_shard_bench.py headline_1m_1080p > $OUT/owner_headline.log 2>&1; grep owner_shard $OUT/owner_headline.log || tail -20 $OUT/owner_headline.log
GS_SHARD_WORLDS=8 timeout 600 python tools/owner_shard_bench.py cfg4_2m_1080p > $OUT/owner_cfg4.log 2>&1; grep owner_shard $OUT/owner_cfg4.log || tail -5 $OUT/owner_cfg4.log
