cd $GRAFT_REPO_ROOT/taichi_3d_gaussian_splatting_amd/csrc
for C in 0 8 30 120 480; do
  rm -f gs_blend.o
  make FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -Wno-unused-function -DGS_XCD_CHUNK=$C" > /dev/null 2>&1
  echo "CHUNK=$C"
  python $GRAFT_REPO_ROOT/tools/stage_bench.py headline_1m_1080p 20 2>&1 | grep -E "blend_forward|blend_backward"
  python $GRAFT_REPO_ROOT/tools/stage_bench.py cfg3_400k_1080p 20 2>&1 | grep -E "blend_forward|blend_backward"
done
