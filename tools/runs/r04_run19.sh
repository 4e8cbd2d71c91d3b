#!/bin/bash
# round 4, GPU call 19: randomised whole-operator parity on the MSD-first sort (fresh seeds), owner-sharded ranks
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run19; mkdir -p $OUT
GS_FUZZ_CASES=160 GS_FUZZ_FIRST=2000 timeout 420 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -x > $OUT/fuzz.log 2>&1; grep -n "passed\|failed" $OUT/fuzz.log | tail -2
GS_SHARD_WORLDS=8 timeout 300 python tools/owner_shard_bench.py headline_1m_1080p 2>&1 | grep owner_shard | tee $OUT/owner_g8.txt
GS_SHARD_WORLDS=8 timeout 300 python tools/owner_shard_bench.py trained_1080p 2>&1 | grep owner_shard | tee -a $OUT/owner_g8.txt
