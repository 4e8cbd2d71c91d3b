#!/bin/bash
# round 4, GPU call 11: longer randomised runs through the new paths (frame entry points, split backward, cooperative SH
# reads, owner sharding)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run11; mkdir -p $OUT
GS_FUZZ_CASES=240 timeout 1500 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -x > $OUT/fuzz.log 2>&1; tail -3 $OUT/fuzz.log
GS_FUZZ_CASES=120 GS_FUZZ_FIRST=1000 timeout 1500 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -x > $OUT/fuzz2.log 2>&1; tail -3 $OUT/fuzz2.log
GS_OWNER_FUZZ_CASES=80 timeout 1500 python -m pytest tests/test_owner_sharding_gpu.py -m gpu -q -x -k random > $OUT/owner_fuzz.log 2>&1; tail -3 $OUT/owner_fuzz.log
