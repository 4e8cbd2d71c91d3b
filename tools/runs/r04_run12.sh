#!/bin/bash
# round 4, GPU call 12: software-pipelined four-waves-per-tile kernels (LDS reads one group ahead, list batches two
# stages ahead in registers) against the round's earlier build, groups of 4 and 8: parity first, then small-grid timings
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run12; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_fuzz_gpu.py tests/test_owner_sharding_gpu.py tests/test_reference_operator.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for lib in base new g8; do
  if [ $lib = new ]; then unset GS_LIB_PATH; else export GS_LIB_PATH=$GRAFT_REPO_ROOT/variants/libgsplat_hip_$lib.so; fi
  echo "=== lib $lib" | tee -a $OUT/perf.log
  for wl in cfg1_10k_256 cfg2_100k_800; do
    GS_BIN_SHIFT=0 timeout 300 python tools/stage_bench.py $wl 30 2>&1 | grep -E "workload|blend_|sum" | tee -a $OUT/perf.log
  done
  GS_NO_CPROFILE=1 timeout 300 python tools/host_profile.py cfg1_10k_256 400 2>&1 | grep host_profile | tee -a $OUT/perf.log
  GS_NO_CPROFILE=1 timeout 300 python tools/host_profile.py cfg2_100k_800 300 2>&1 | grep host_profile | tee -a $OUT/perf.log
  for shift in "" 0; do
    GS_BIN_SHIFT=$shift GS_SHARD_WORLDS=8,4,2 timeout 600 python tools/owner_shard_bench.py headline_1m_1080p 2>&1 | grep owner_shard | cut -c1-330 | sed "s/^/[shift=$shift] /" | tee -a $OUT/perf.log
  done
done
