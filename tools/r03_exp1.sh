#!/bin/bash
# Round 3, GPU call 1: atomics / LDS-sort micro-benchmark, GPU tests on the new default build, backward-kernel arms.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/exp1
mkdir -p $OUT
cd $ROOT
timeout 300 ./variants/bin_append_sort > $OUT/ubench.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
for lib in base notrim r2 r0 w6; do
  for bs in 0 1; do
    echo "=== lib=$lib bin_shift=$bs"
    GS_LIB_PATH=$ROOT/variants/libgsplat_hip_$lib.so GS_BIN_SHIFT=$bs GS_TILE_ORDER=1 timeout 300 python tools/stage_bench.py headline_1m_1080p 20
  done
done > $OUT/stage.log 2>&1
for w in cfg3_400k_1080p stress_t_ras; do
  for lib in base r2; do
    echo "=== lib=$lib workload=$w"
    GS_LIB_PATH=$ROOT/variants/libgsplat_hip_$lib.so GS_BIN_SHIFT=$([ $w = stress_t_ras ] && echo 2 || echo 1) GS_TILE_ORDER=1 timeout 300 python tools/stage_bench.py $w 10
  done
done >> $OUT/stage.log 2>&1
cat $OUT/ubench.log
grep -E "===|blend_backward|checksums|lpt" $OUT/stage.log
