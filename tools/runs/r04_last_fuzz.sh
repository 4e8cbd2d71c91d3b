#!/bin/bash
# round 4, third session: the randomised parity suites on fresh seeds, on the final tree
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/last_r04
mkdir -p $OUT
cd $ROOT
GS_FUZZ_CASES=300 GS_FUZZ_FIRST=${GS_FIRST:-6000} timeout 420 python -m pytest tests/test_fuzz_gpu.py -m gpu -q > $OUT/fuzz_${GS_FIRST:-6000}.log 2>&1
grep -E "passed|failed" $OUT/fuzz_${GS_FIRST:-6000}.log | tail -2; grep -E "^FAILED" $OUT/fuzz_${GS_FIRST:-6000}.log | head
