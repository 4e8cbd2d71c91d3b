#!/bin/bash
# Builds differently tuned variants of libgsplat_hip.so into variants/ (development tool for tuning sweeps:
# GS_LIB_PATH=variants/libgsplat_hip_<tag>.so python tools/stage_bench.py ...).  usage: tools/build_variants.sh "tag:-DFLAG=.. -DFLAG2=.." ...
# A variant that switches a MEASUREMENT ARM on (GS_STATS, GS_ABLATE_FWD, GS_MFMA_REDUCE, GS_BWD_REDUCE_ARM=0) must also pass
# -DGS_TUNING_BUILD=1 (the sources refuse to compile otherwise); the library then reports another ABI version and only loads
# with GS_ALLOW_TUNING_LIB=1 in the environment.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/taichi_3d_gaussian_splatting_amd/csrc
OUT=$ROOT/variants
mkdir -p $OUT
for spec in "$@"; do
    tag=${spec%%:*}; flags=${spec#*:}
    tmp=$(mktemp -d)
    for f in gs_api gs_frame gs_frontend gs_sort gs_blend gs_shard gs_point_backward gs_controller gs_loss gs_optim; do
        extra="$flags"   # (macros are file-specific: GS_GROUP_*, GS_RP_* in gs_blend, GS_SORT_* in gs_sort, ...)
        case $f in gs_frontend|gs_point_backward) extra="$extra -fno-slp-vectorize";; esac   # (as csrc/Makefile)
        /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -Wno-unused-function $extra -c $SRC/$f.hip -o $tmp/$f.o &
    done
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $tmp/*.o -o $OUT/libgsplat_hip_$tag.so
    rm -rf $tmp
    echo "built $tag ($flags)"
done
