#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* out) {
  float x = 100.f + threadIdx.x, y = 1000.f + threadIdx.x;
  float a = x, b = y;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  out[threadIdx.x] = a; out[64 + threadIdx.x] = b;
  float c = x, d = y;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(c), "+v"(d));
  out[128 + threadIdx.x] = c; out[192 + threadIdx.x] = d;
}
int main() {
  float* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  const char* names[4] = {"swap32 vdst(x=100+l)", "swap32 src0(y=1000+l)", "swap16 vdst", "swap16 src0"};
  for (int r = 0; r < 4; ++r) { printf("%s:", names[r]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%g", l, h[r * 64 + l]); printf("\n"); }
  return 0;
}
