#!/bin/bash
# Ablation builds of the two VALU-heavy front-end kernels (measurement only, never shipped): a patched COPY of
# csrc/gs_frontend.hip (tools/ubench/frontend_ablation.patch) compiled with -DGS_ABLATE_KEYS=1|2|3 (gs_make_keys: stop after
# the scans / after the record loads and box arithmetic / walk without stores) and -DGS_ABLATE_PRE=1|2|3 (gs_preprocess: no
# colour / no bin walk / neither), linked with the tree's other objects into variants/libgsplat_hip_abl_<tag>.so.
# Time them with tools/ablate_probe.py (GS_LIB_PATH=...).  The tree's sources (and their profile hash) stay untouched.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/taichi_3d_gaussian_splatting_amd/csrc
TMP=$(mktemp -d)
mkdir -p $TMP/pkg $ROOT/variants
cp -r $SRC $TMP/pkg/csrc && cp -r $ROOT/include $TMP/include
patch -s $TMP/pkg/csrc/gs_frontend.hip $ROOT/tools/ubench/frontend_ablation.patch
make -s -C $SRC > /dev/null
cd $TMP/pkg/csrc
for spec in "k1:-DGS_ABLATE_KEYS=1" "k2:-DGS_ABLATE_KEYS=2" "k3:-DGS_ABLATE_KEYS=3" "p1:-DGS_ABLATE_PRE=1" "p2:-DGS_ABLATE_PRE=2" "p3:-DGS_ABLATE_PRE=3"; do
    tag=${spec%%:*}; flags=${spec#*:}
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -Wno-unused-function $flags -c gs_frontend.hip -o $TMP/fe_$tag.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $SRC/gs_api.o $SRC/gs_frame.o $TMP/fe_$tag.o $SRC/gs_sort.o $SRC/gs_blend.o \
        $SRC/gs_shard.o $SRC/gs_point_backward.o $SRC/gs_controller.o $SRC/gs_loss.o $SRC/gs_optim.o -o $ROOT/variants/libgsplat_hip_abl_$tag.so
    echo "built $tag ($flags)"
done
rm -rf $TMP
