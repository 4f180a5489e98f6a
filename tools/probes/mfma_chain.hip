// Probe (GPU box): cycles per v_mfma_f32_32x32x2_f32 when a wave issues them as ONE dependent accumulator chain, or
// as 2 / 4 independent chains, with 1 / 2 / 4 waves on a SIMD.  hipcc -O3 --offload-arch=gfx950 mfma_chain.hip -o mfma_chain.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ void k(float *out, long long *cyc, int n, float a0, float b0) {
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = f32x16{0};
  float a = a0 + threadIdx.x, b = b0;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i += CH) {
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0;
  for (int c = 0; c < CH; ++c)
    for (int q = 0; q < 16; ++q) s += acc[c][q];
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}
template <int CH>
void run(int waves) {
  float *o; long long *c; long long h[64];
  (void)hipMalloc(&o, 4 * 64 * 64 * 16); (void)hipMalloc(&c, sizeof h);
  const int n = 4096;
  for (int rep = 0; rep < 2; ++rep) k<CH><<<1, 64 * waves>>>(o, c, n, 1.0f, 0.5f);
  (void)hipMemcpy(h, c, sizeof(long long) * waves, hipMemcpyDeviceToHost);
  long long mx = 0;
  for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
  // s_memtime counts at 100 MHz on this chip; readcyclecounter = s_memtime -> report raw and let the reader scale
  printf("chains %d  waves/CU %2d (%d per SIMD): %lld counter ticks for %d MFMAs per wave -> %.3f ticks per MFMA per wave, %.3f per MFMA on a SIMD\n",
         CH, waves, (waves + 3) / 4, mx, n, (double)mx / n, (double)mx / n / ((waves + 3) / 4));
}
int main() {
  run<1>(1); run<2>(1); run<4>(1);
  run<1>(4); run<2>(4);
  run<1>(8); run<2>(8);
  run<1>(16); run<2>(16); run<4>(16);
  return 0;
}
