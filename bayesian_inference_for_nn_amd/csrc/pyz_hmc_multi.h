// pyz_hmc_multi.h -- one HMC proposal of a small 2-layer MLP spread over NW workgroups per chain.
//
// k_hmc_fused (pyz_hmc_fused.h) keeps a chain on ONE compute unit: the right shape when a GPU runs
// many chains, but a single chain (the reference's HMC: one chain, HMC.py:74-104; BASELINE config
// "8 chains on 8 GPUs" = one per GPU) then leaves 255 CUs idle and pays ~20 us of VALU work per
// gradient evaluation, 22 of them per sample.  Here the data rows are cut into NW slices:
//   launch t = 0 .. L   (grid NW x chains, 256 threads): every workgroup rebuilds the state q, p of
//       step t from the previous launch's buffers -- sum of the NW partial gradients in a fixed
//       order, then the kick / drift that follows gradient t-1 (HMC.py:82-87); workgroup 0 stores
//       it -- and evaluates the gradient at q over ITS rows (pyz_hf_loss_grad on 4 waves) into
//       slab[t & 1];
//   k_hmc_multi_final (grid chains): last kick, energies, Metropolis test, write-back (HMC.py:88-104).
// The kernel boundary is the cross-workgroup reduction (no in-kernel waits), state and slabs
// ping-pong between launches, and every sum has a fixed order: results do not depend on timing.
#pragma once

#include "pyz_hmc_fused.h"

#define PYZ_HM_THREADS 256
#define PYZ_HM_WAVES 4
#define PYZ_HM_MAXW 32

// what changes from proposal to proposal lives in device memory (uploaded with the uniforms), so
// that the launch sequence of a proposal can be captured once in a hipGraph and replayed
struct HmcCall {
  uint64_t seed;
  uint32_t step;
  int32_t burning;
};

struct HmcMultiArgs {
  HmcFusedArgs f;         // f.seed / f.step / f.burning are NOT read by the sliced kernels: see `call`
  const HmcCall *call;
  int NW;         // workgroups (row slices) per chain
  int t;          // gradient evaluation of this launch, 0 .. L
  int max_rows;   // rows of the largest slice
  float *qw;      // (2, P, D) state ping-pong: launch t reads [(t-1)&1], workgroup 0 writes [t&1]
  float *pw;      // (2, P, D)
  float *slab;    // (2, P, NW, D) partial gradients
  double *lpart;  // (2, P, NW) partial sums of the row losses
  float *scal;    // (P, 4): sum p^2 and sum log prior at the start, loss at the start
};

static inline size_t pyz_hmc_multi_lds_bytes(int max_rows, int MI, int MC, int C, int D, int loss) {
  const size_t fl = (size_t)(3 + PYZ_HM_WAVES) * D + (size_t)64 * (MI + MC + 2) + (size_t)max_rows * MI +
                    (size_t)max_rows * MC + (size_t)max_rows * (loss == PYZ_LOSS_MSE ? C : 1);
  return ((fl * 4 + 15) / 16) * 16 + 64 * sizeof(double);
}

// element e of the summed gradient: the NW slabs in slice order; the loads go out sixteen at a time (one
// load per round trip would put NW dependent latencies in front of every launch; the slabs were written by the
// previous launch on other XCDs, so a round trip goes to the Infinity Cache: with 16 slices per chain the sum
// is ONE round trip instead of two)
__device__ __forceinline__ float pyz_hm_slab_sum(const float *sl, const int NW, const int D, const int e) {
  float gs = 0.0f;
  for (int k0 = 0; k0 < NW; k0 += 16) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = sl[(long long)min(k0 + j, NW - 1) * D + e];
#pragma unroll
    for (int j = 0; j < 16; ++j) gs += k0 + j < NW ? v[j] : 0.0f;
  }
  return gs;
}

// momentum of element e (HMC.py:168-171: p = m z)
__device__ __forceinline__ float pyz_hm_momentum(const HmcFusedArgs &a, const HmcCall &c, const int chain, const int e) {
  float z;
  if (a.unit_p) {
    z = a.unit_p[(long long)chain * a.D + e];
  } else {
    const float4 v = pyz_normal4(c.seed, PYZ_STREAM_HMC + 16u * (uint32_t)chain, c.step, (uint64_t)(e >> 2));
    const int k = e & 3;
    z = k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w));
  }
  return a.m * z;
}

template <int MI, int MC, int ACT>
__global__ void __launch_bounds__(PYZ_HM_THREADS) k_hmc_multi(HmcMultiArgs m) {
  extern __shared__ float lds[];
  const HmcFusedArgs &a = m.f;
  const int D = a.D, N = a.N, I = a.I, C = a.C;
  const int t = threadIdx.x, wg = blockIdx.x, chain = blockIdx.y, P = gridDim.y, NW = m.NW;
  const int r0 = (int)(((long long)N * wg) / NW), r1 = (int)(((long long)N * (wg + 1)) / NW), nloc = r1 - r0;
  float *q = lds, *p = q + D, *g = p + D, *part = g + D;
  float *wj = part + PYZ_HM_WAVES * D, *xs = wj + 64 * (MI + MC + 2), *d2 = xs + m.max_rows * MI, *yf = d2 + m.max_rows * MC;
  const size_t fl = (size_t)(3 + PYZ_HM_WAVES) * D + (size_t)64 * (MI + MC + 2) + (size_t)m.max_rows * MI +
                    (size_t)m.max_rows * MC + (size_t)m.max_rows * (a.loss == PYZ_LOSS_MSE ? C : 1);
  double *sm = reinterpret_cast<double *>(lds + ((fl * 4 + 15) / 16) * 4);
  // ---- the state of step t (every workgroup computes the same values; workgroup 0 stores them).  Its operands were
  // written by the previous launch on other XCDs (a round trip to the Infinity Cache): this thread's first element
  // is requested BEFORE the data slice is staged, so that the two round trips overlap instead of following each other
  const long long so = (long long)chain * D;
  float *q_out = m.qw + ((long long)(m.t & 1) * P) * D + so, *p_out = m.pw + ((long long)(m.t & 1) * P) * D + so;
  const int pb0 = (m.t - 1) & 1;
  const float *sl0 = m.slab + (((long long)pb0 * P + chain) * NW) * D;
  float pre_q = 0.0f, pre_p = 0.0f, pre_v[16];
  const bool pre_on = m.t > 0 && NW <= 16;   // uniform
  if (pre_on) {
    const int e = min(t, D - 1);
    pre_q = m.qw[((long long)pb0 * P) * D + so + e];
    pre_p = m.pw[((long long)pb0 * P) * D + so + e];
#pragma unroll
    for (int j = 0; j < 16; ++j) pre_v[j] = sl0[(long long)min(j, NW - 1) * D + e];
  }
  // ---- this slice of the data set
  for (int e = t; e < nloc * MI; e += PYZ_HM_THREADS) {
    const int r = e / MI, i = e - r * MI;
    xs[e] = i < I ? a.x[(long long)(r0 + r) * I + i] : 0.0f;
  }
  if (a.loss == PYZ_LOSS_SCCE) {
    for (int e = t; e < nloc; e += PYZ_HM_THREADS) yf[e] = __int_as_float(reinterpret_cast<const int32_t *>(a.y)[r0 + e]);
  } else {
    for (int e = t; e < nloc * C; e += PYZ_HM_THREADS) yf[e] = reinterpret_cast<const float *>(a.y)[(long long)r0 * C + e];
  }
  if (m.t == 0) {
    double sp2 = 0.0, slp = 0.0;
    const float ls = logf(a.prior_sigma);
    const HmcCall call = *m.call;
    for (int e = t; e < D; e += PYZ_HM_THREADS) {
      const float qv = a.q[so + e], pv = pyz_hm_momentum(a, call, chain, e);
      q[e] = qv;
      p[e] = pv;
      if (wg == 0) {
        q_out[e] = qv;
        p_out[e] = pv;
        sp2 += (double)(pv * pv);
        const float u = (qv - a.prior_mean) / a.prior_sigma;
        slp += (double)(-0.5f * u * u - ls - PYZ_LOG_SQRT_2PI);
      }
    }
    if (wg == 0) {  // uniform per workgroup
      const double s0 = pyz_hf_block_sum<PYZ_HM_WAVES>(sp2, sm), s1 = pyz_hf_block_sum<PYZ_HM_WAVES>(slp, sm);
      if (t == 0) {
        m.scal[chain * 4 + 0] = (float)s0;
        m.scal[chain * 4 + 1] = (float)s1;
      }
    }
  } else {
    const int pb = (m.t - 1) & 1;
    const float *q_in = m.qw + ((long long)pb * P) * D + so, *p_in = m.pw + ((long long)pb * P) * D + so;
    const float *sl = m.slab + (((long long)pb * P + chain) * NW) * D;
    const float eps = a.epsilon, drift = eps / a.m, n_train = (float)N;
    const float isig2 = 1.0f / (a.prior_sigma * a.prior_sigma);
    for (int e = t; e < D; e += PYZ_HM_THREADS) {
      float gs, qv, pin;
      if (pre_on && e == t) {   // the element requested ahead (same values, same order of summation)
        gs = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) gs += j < NW ? pre_v[j] : 0.0f;
        qv = pre_q;
        pin = pre_p;
      } else {
        gs = pyz_hm_slab_sum(sl, NW, D, e);
        qv = q_in[e];
        pin = p_in[e];
      }
      const float dU = (qv - a.prior_mean) * isig2 + n_train * gs;
      float pv, qn;
      if (m.t == 1) {  // behind the first gradient: half kick (HMC.py:82), then the first drift
        pv = pin - (eps / 2) * dU;
        qn = qv + drift * pv;
      } else {         // kick + drift (HMC.py:84-86)
        pv = pin - eps * dU;
        qn = qv + drift * pv;
      }
      q[e] = qn;
      p[e] = pv;
      if (wg == 0) {
        q_out[e] = qn;
        p_out[e] = pv;
      }
    }
    if (m.t == 1 && wg == 0 && t == 0) {  // the loss at the starting point, for U0
      const double *lp = m.lpart + ((long long)pb * P + chain) * NW;
      double v = 0.0;
      for (int k = 0; k < NW; ++k) v += lp[k];
      m.scal[chain * 4 + 2] = (float)(v / (double)N);
    }
  }
  __syncthreads();
  // ---- gradient of the mean loss over this slice of the rows
  const double lsum = pyz_hf_loss_grad<MI, MC, ACT, PYZ_HM_WAVES>(a, q, g, part, wj, xs, d2, yf, sm, nloc);
  float *so_slab = m.slab + ((((long long)(m.t & 1) * P + chain) * NW) + wg) * D;
  for (int e = t; e < D; e += PYZ_HM_THREADS) so_slab[e] = g[e];
  if (t == 0) m.lpart[((long long)(m.t & 1) * P + chain) * NW + wg] = lsum;
}

// last kick, energies, Metropolis test and write-back; one workgroup per chain
__global__ void __launch_bounds__(PYZ_HM_THREADS) k_hmc_multi_final(HmcMultiArgs m) {
  __shared__ double sm[PYZ_HM_WAVES];
  const HmcFusedArgs &a = m.f;
  const int D = a.D, N = a.N, NW = m.NW, L = a.L;
  const int t = threadIdx.x, chain = blockIdx.x, P = gridDim.x;
  const int pb = L & 1;
  const long long so = (long long)chain * D;
  const float *q_in = m.qw + ((long long)pb * P) * D + so, *p_in = m.pw + ((long long)pb * P) * D + so;
  const float *sl = m.slab + (((long long)pb * P + chain) * NW) * D;
  const float eps = a.epsilon, n_train = (float)N;
  const float isig2 = 1.0f / (a.prior_sigma * a.prior_sigma), ls = logf(a.prior_sigma);
  double sp2 = 0.0, slp = 0.0;
  for (int e = t; e < D; e += PYZ_HM_THREADS) {
    const float gs = pyz_hm_slab_sum(sl, NW, D, e);
    const float qv = q_in[e];
    const float dU = (qv - a.prior_mean) * isig2 + n_train * gs;
    float pv;
    if (L == 0) {  // the only gradient: both half kicks (HMC.py:82, 87)
      pv = p_in[e] - (eps / 2) * dU;
      pv = pv - (eps / 2) * dU;
    } else {       // last kick and the closing half kick share the last gradient
      pv = p_in[e] - eps * dU;
      pv = pv - (eps / 2) * dU;
    }
    sp2 += (double)(pv * pv);
    const float u = (qv - a.prior_mean) / a.prior_sigma;
    slp += (double)(-0.5f * u * u - ls - PYZ_LOG_SQRT_2PI);
  }
  const float sp2_1 = (float)pyz_hf_block_sum<PYZ_HM_WAVES>(sp2, sm);
  const float slp_1 = (float)pyz_hf_block_sum<PYZ_HM_WAVES>(slp, sm);
  const double *lp = m.lpart + ((long long)pb * P + chain) * NW;
  double lv = 0.0;
  for (int k = 0; k < NW; ++k) lv += lp[k];
  const float loss1 = (float)(lv / (double)N);
  const float loss0 = L == 0 ? loss1 : m.scal[chain * 4 + 2];
  float U0 = 0.0f - m.scal[chain * 4 + 1];
  U0 = U0 + loss0 * n_train;
  const float K0 = (1.0f / (2.0f * a.m)) * m.scal[chain * 4 + 0];
  float U1 = 0.0f - slp_1;
  U1 = U1 + loss1 * n_train;
  const float K1 = (1.0f / (2.0f * a.m)) * sp2_1;
  const float lr = K0 + U0 - K1 - U1;
  const bool acc = m.call->burning || (a.uniform[chain] < expf(lr));
  if (acc)
    for (int e = t; e < D; e += PYZ_HM_THREADS) a.q[so + e] = q_in[e];
  if (t == 0) {
    float *s = a.stats + chain * 8;
    s[0] = acc ? 1.0f : 0.0f;
    s[1] = acc ? loss1 : loss0;
    s[2] = U0;
    s[3] = K0;
    s[4] = U1;
    s[5] = K1;
    s[6] = lr;
    s[7] = 0.0f;
  }
}
