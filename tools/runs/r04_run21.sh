#!/bin/bash
# round 4, GPU call 21: owner-sharded bands placed by the ranks' walk lengths
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run21; mkdir -p $OUT
timeout 900 python -m pytest tests/test_owner_sharding_gpu.py tests/test_multirank_gpu.py -m gpu -q -x > $OUT/pytest_owner.log 2>&1; grep -n "passed\|failed\|Error\|assert" $OUT/pytest_owner.log | tail -8
for w in trained_1080p headline_1m_1080p cfg4_2m_1080p; do
for bal in 0 1; do GS_SHARD_BALANCE=$bal GS_SHARD_WORLDS=8 timeout 300 python tools/owner_shard_bench.py $w 2>&1 | grep owner_shard | sed "s/^/balance=$bal /"; done; done | tee $OUT/owner_balance.txt | cut -c1-420
