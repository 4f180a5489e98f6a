// tools/probes/lds_dma.hip -- what an LDS-DMA instruction (16 bytes per lane) writes where, on gfx950:
//   global_load_lds_dwordx4 and buffer_load_dwordx4 ... lds with per-lane sources, out-of-range lanes, an M0 base,
//   and the lane maps of v_mfma_f32_16x16x4_f32.  Build: hipcc --offload-arch=gfx950 -O2 lds_dma.hip -o lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_dma(const float *g, float *out, int mode) {
  __shared__ __attribute__((aligned(16))) float a0[1024], a1[1024];
  const int l = threadIdx.x;
  for (int i = l; i < 1024; i += 64) { a0[i] = -1.0f; a1[i] = -2.0f; }
  __syncthreads();
  if (mode == 0) {
    __builtin_amdgcn_global_load_lds((glb_void *)(g + 12 * l), (lds_void *)(a1 + 64), 16, 0, 0);
  } else {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g), 0, 4096 * 4, 0x00020000);
    const unsigned voff = (l % 5 == 4) ? 0x80000000u : (unsigned)(12 * l * 4);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)(a1 + 64), 16, (int)voff, 0, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  for (int i = l; i < 1024; i += 64) { out[i] = a0[i]; out[1024 + i] = a1[i]; }
}

__global__ void k_mfma(float *out) {
  const int l = threadIdx.x;
  // A[i][k] = 100 i + k, B[k][j] = (k == 2) ? j + 1 : 0   ->  D[i][j] = (100 i + 2) (j + 1)
  const float a = 100.0f * (l & 15) + (l >> 4);
  const float b = ((l >> 4) == 2) ? (float)((l & 15) + 1) : 0.0f;
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[4 * l + r] = c[r];
}

int main() {
  std::vector<float> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = (float)i;
  float *g, *o;
  hipMalloc(&g, 4096 * 4);
  hipMalloc(&o, 2048 * 4);
  hipMemcpy(g, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  std::vector<float> r(2048);
  for (int mode = 0; mode < 2; ++mode) {
    k_dma<<<1, 64>>>(g, o, mode);
    hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
    int bad0 = 0;
    for (int i = 0; i < 1024; ++i) bad0 += r[i] != -1.0f;
    printf("mode %d: other array touched %d; a1[56..84]:", mode, bad0);
    for (int i = 56; i < 84; ++i) printf(" %g", r[1024 + i]);
    int ok = 0, zero = 0, untouched = 0;
    for (int l = 0; l < 64; ++l)
      for (int c = 0; c < 4; ++c) {
        const float v = r[1024 + 64 + 4 * l + c];
        ok += v == 12.0f * l + c;
        zero += v == 0.0f;
        untouched += v == -2.0f;
      }
    printf("\n   lane-linear 16-byte pieces: %d of 256 as expected, %d zero, %d untouched; after the piece range: %g %g\n", ok, zero, untouched,
           r[1024 + 64 + 256], r[1024 + 63]);
  }
  k_mfma<<<1, 64>>>(o);
  hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
  int good = 0;
  for (int l = 0; l < 64; ++l)
    for (int q = 0; q < 4; ++q) {
      const int row = 4 * (l >> 4) + q, col = l & 15;
      good += r[4 * l + q] == (100.0f * row + 2) * (col + 1);
    }
  printf("mfma 16x16x4: %d of 256 match D[r]: row 4 (lane / 16) + r, col lane %% 16 with A[l&15][l>>4], B[l>>4][l&15]\n", good);
  return 0;
}
