#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run10; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_multirank_gpu.py tests/test_owner_sharding_gpu.py -m gpu -q -x > $OUT/pytest_multi.log 2>&1; tail -15 $OUT/pytest_multi.log
