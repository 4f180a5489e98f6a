// Probe (GPU box): cycles per v_mfma_f32_16x16x4_f32 for CH independent accumulators, one wave per SIMD (256 threads per
// workgroup, 256 workgroups), bare and with one ds_read_b32 + counted s_waitcnt per instruction (the inner loop of
// k_dense_fwd_ring), reported in s_memtime ticks AND in wall-clock nanoseconds (hipEvent): their ratio is the clock.
// hipcc -O3 --offload-arch=gfx950 mfma16_rate.hip -o mfma16_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH, int MODE>
__global__ void __launch_bounds__(256) k(float *out, long long *cyc, int n, float a0) {
  __shared__ float lds[15360];
  for (int i = threadIdx.x; i < 15360; i += 256) lds[i] = 0.001f * i;
  __syncthreads();
  f32x4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = f32x4{0, 0, 0, 0};
  float a = a0 + threadIdx.x;
  float b[2][CH];
  for (int c = 0; c < CH; ++c) b[0][c] = b[1][c] = 0.5f + c;
  // MODE 1: consecutive lanes, consecutive dwords.  MODE 2: the fragment addresses of k_dense_fwd_ring (lane quarter q reads
  // row 4 q of a 204-float-wide image, 16 columns; waves 0 / 1 the first seven column sub-tiles, waves 2 / 3 the rest)
  const int l_ = threadIdx.x & 63, w_ = threadIdx.x >> 6;
  const unsigned addr = MODE == 2 ? (unsigned)((4 * (l_ >> 4) * 204 + 16 * 7 * (w_ >> 1) + (l_ & 15)) * 4) : (unsigned)(threadIdx.x * 4);
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; i += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {   // (unrolled by two: every register index is static)
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (MODE >= 1) {
          asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(b[h][c]) : "n"(CH - 1));
        }
        if (MODE == 3) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b[h][c]));      // accumulator in VGPRs
        else if (MODE == 4) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[c]) : "v"(a), "v"(b[h][c])); // ... in AGPRs
        else acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[h][c], acc[c], 0, 0, 0);
        if (MODE >= 1) {
          asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(b[h ^ 1][c]) : "v"(addr), "n"(MODE == 2 ? 64 * c + 816 * 2 * 1 : 64 * c + 1024));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int c = 0; c < CH; ++c)
    for (int q = 0; q < 4; ++q) s += acc[c][q] + b[0][c] + b[1][c];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CH>
__global__ void __launch_bounds__(256) k32(float *out, long long *cyc, int n, float a0) {
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = f32x16{0};
  float a = a0 + threadIdx.x, b = 0.5f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b + c, acc[c], 0, 0, 0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int c = 0; c < CH; ++c)
    for (int q = 0; q < 16; ++q) s += acc[c][q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class F>
void run(const char *name, int per_iter, double flop_per, F launch) {
  float *o; long long *c; long long h[256];
  (void)hipMalloc(&o, 4 * 256 * 256); (void)hipMalloc(&c, sizeof h);
  const int n = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(o, c, n);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) launch(o, c, n);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  (void)hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost);
  long long sum = 0;
  for (int w = 0; w < 256; ++w) sum += h[w];
  const double ticks = (double)sum / 256 / ((double)n * per_iter), ns = ms * 1e6 / 20 / ((double)n * per_iter);
  printf("%-44s %6.1f s_memtime ticks, %6.2f ns wall per instruction and wave  (ticks/ns %.2f)  chip %.1f TFLOP/s\n", name, ticks, ns,
         ticks / ns, flop_per * 1024 / ns / 1e3);
  hipFree(o); hipFree(c);
}

int main() {
#define R16(CH, MODE, NAME) run(NAME, CH, 2048.0, [](float *o, long long *c, int n) { k<CH, MODE><<<256, 256>>>(o, c, n, 1.0f); })
  R16(4, 0, "16x16x4, 4 accumulators, bare");
  R16(7, 0, "16x16x4, 7 accumulators, bare");
  R16(7, 1, "16x16x4, 7 accumulators, + ds_read + wait");
  R16(6, 1, "16x16x4, 6 accumulators, + ds_read + wait");
  R16(7, 3, "16x16x4, 7 acc in VGPRs (asm), + ds_read");
  R16(7, 4, "16x16x4, 7 acc in AGPRs (asm), + ds_read");
  R16(4, 3, "16x16x4, 4 acc in VGPRs (asm), + ds_read");
  R16(7, 2, "16x16x4, 7 acc, + ds_read (ring pattern)");
  R16(6, 2, "16x16x4, 6 acc, + ds_read (ring pattern)");
  run("32x32x2, 2 accumulators, bare", 2, 4096.0, [](float *o, long long *c, int n) { k32<2><<<256, 256>>>(o, c, n, 1.0f); });
  run("32x32x2, 4 accumulators, bare", 4, 4096.0, [](float *o, long long *c, int n) { k32<4><<<256, 256>>>(o, c, n, 1.0f); });
  return 0;
}
