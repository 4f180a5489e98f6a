// Diagnostic: what v_permlane32_swap_b32 does on gfx950 (which halves of the two registers trade places).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
  unsigned a = threadIdx.x, b = threadIdx.x + 1000;
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);   // (the builtin: the compiler places the wait states)
  o[threadIdx.x] = r[0];
  o[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned *d, h[128];
  (void)hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("a: lane0 %u lane31 %u lane32 %u lane63 %u\n", h[0], h[31], h[32], h[63]);
  printf("b: lane0 %u lane31 %u lane32 %u lane63 %u\n", h[64], h[95], h[96], h[127]);
  return 0;
}
