// pyz_rng.h -- counter-based Philox4x32-10 + Box-Muller for gfx950.
//
// Replaces the reference's unseeded TensorFlow draws (tf.random.normal at
// Pyesian/optimizers/SGLD.py:67 and HMC.py:171; tfp samplers at BBB.py:234-237,
// SVGD.py:154).  Stream definition (restated by oracle/philox.py for checking):
//   counter = (idx4_lo, idx4_hi, step, stream), key = (seed_lo, seed_hi)
//   element e uses idx4 = e / 4 and output word e % 4
//   u = ((word >> 9) + 0.5) * 2^-23   (exact in fp32, never 0 or 1)
//   words (0,1) -> (r cos t, r sin t), words (2,3) likewise,
//   r = sqrt(-2 ln u_a), t = 2 pi u_b.
#pragma once

#include "pyz_common.h"

#define PYZ_STREAM_SGLD 0u
#define PYZ_STREAM_BBB 1u
#define PYZ_STREAM_HMC 2u   // + chain index in the high bits: stream = 2 + 16 * chain
#define PYZ_STREAM_INIT 3u
#define PYZ_STREAM_PREDICT 4u

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
    uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += W0;
    k.y += W1;
  }
  return c;
}

__device__ __forceinline__ float pyz_unit(uint32_t w) {
  return ((float)(w >> 9) + 0.5f) * 1.1920928955078125e-07f;  // 2^-23
}

// four standard normals for elements 4*idx4 .. 4*idx4+3
__device__ __forceinline__ float4 pyz_normal4(uint64_t seed, uint32_t stream, uint32_t step,
                                              uint64_t idx4) {
  uint4 r = philox4x32_10(make_uint4((uint32_t)idx4, (uint32_t)(idx4 >> 32), step, stream),
                          make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
  float ra = sqrtf(-2.0f * logf(pyz_unit(r.x)));
  float rb = sqrtf(-2.0f * logf(pyz_unit(r.z)));
  float sa, ca, sb, cb;
  sincospif(2.0f * pyz_unit(r.y), &sa, &ca);
  sincospif(2.0f * pyz_unit(r.w), &sb, &cb);
  return make_float4(ra * ca, ra * sa, rb * cb, rb * sb);
}

// the standard normal of ONE element e (= component e % 4 of pyz_normal4 for idx4 = e / 4, bit for bit): the words of the
// element's pair are selected before the Box-Muller step, so there is one logarithm / square root / sincos instead of
// two and no branch on e % 4 (a divergent branch inside a latency-hiding hook makes the compiler wait for every load in flight)
__device__ __forceinline__ float pyz_normal1(uint64_t seed, uint32_t stream, uint32_t step, uint64_t e) {
  const uint64_t idx4 = e >> 2;
  const uint4 r = philox4x32_10(make_uint4((uint32_t)idx4, (uint32_t)(idx4 >> 32), step, stream),
                                make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
  const bool second = (e & 2) != 0, odd = (e & 1) != 0;
  const float rad = sqrtf(-2.0f * logf(pyz_unit(second ? r.z : r.x)));
  float sn, cs;
  sincospif(2.0f * pyz_unit(second ? r.w : r.y), &sn, &cs);
  return rad * (odd ? sn : cs);
}
