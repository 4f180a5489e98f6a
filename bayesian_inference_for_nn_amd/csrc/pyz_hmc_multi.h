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

#ifndef PYZ_HM_WAVES
#define PYZ_HM_WAVES 4
#endif
#define PYZ_HM_THREADS (64 * PYZ_HM_WAVES)
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
  // k_hmc_resident only
  unsigned *epoch;               // (P) per chain: the tag below which every granule in `gran` is stale (advanced by L + 1 per proposal)
  unsigned long long *gran;      // (2, P, NW, Dp) {tag, value} granules: the partial gradients (and, in the last two, the loss sum) of a slice
  int Dp;                        // granules per slice: a multiple of 32, >= D + 2
  int spin_limit;                // sweeps without progress before a workgroup gives up (stats[7] = -1, q untouched)
  int diag;                      // PYZ_HMC_DIAG builds only (timing, wrong results): 1 = no exchange, 2 = no gradient evaluation
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

// The leapfrog arithmetic with its fused multiply-adds written out: k_hmc_multi (+ final) and k_hmc_resident must give the
// same bits, and left to the compiler the contraction of a * b + c depends on the code around it.
__device__ __forceinline__ float pyz_hm_dU(const float qv, const float pmean, const float isig2, const float n_train, const float gs) {
  return __fmaf_rn(n_train, gs, (qv - pmean) * isig2);
}
__device__ __forceinline__ float pyz_hm_kick(const float pv, const float k, const float dU) { return __fmaf_rn(-k, dU, pv); }
__device__ __forceinline__ float pyz_hm_drift(const float qv, const float drift, const float pv) { return __fmaf_rn(drift, pv, qv); }
__device__ __forceinline__ float pyz_hm_log_prior(const float qv, const float pmean, const float sigma, const float ls) {
  const float u = (qv - pmean) / sigma;
  return __fmaf_rn(-0.5f * u, u, -ls) - PYZ_LOG_SQRT_2PI;
}

__device__ __forceinline__ float pyz_hm_potential(const float sum_log_prior, const float loss, const float n_train) {
  return __fmaf_rn(loss, n_train, 0.0f - sum_log_prior);   // U = -sum log prior + N * mean loss
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
        slp += (double)pyz_hm_log_prior(qv, a.prior_mean, a.prior_sigma, ls);
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
      const float dU = pyz_hm_dU(qv, a.prior_mean, isig2, n_train, gs);
      // behind the first gradient: half kick (HMC.py:82); behind the others: kick (HMC.py:84-86); then the drift
      const float pv = pyz_hm_kick(pin, m.t == 1 ? eps / 2 : eps, dU);
      const float qn = pyz_hm_drift(qv, drift, pv);
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
    const float dU = pyz_hm_dU(qv, a.prior_mean, isig2, n_train, gs);
    // the last kick and the closing half kick share the last gradient (L == 0, the only gradient: both half kicks, HMC.py:82, 87)
    float pv = pyz_hm_kick(p_in[e], L == 0 ? eps / 2 : eps, dU);
    pv = pyz_hm_kick(pv, eps / 2, dU);
    sp2 += (double)(pv * pv);
    slp += (double)pyz_hm_log_prior(qv, a.prior_mean, a.prior_sigma, ls);
  }
  const float sp2_1 = (float)pyz_hf_block_sum<PYZ_HM_WAVES>(sp2, sm);
  const float slp_1 = (float)pyz_hf_block_sum<PYZ_HM_WAVES>(slp, sm);
  const double *lp = m.lpart + ((long long)pb * P + chain) * NW;
  double lv = 0.0;
  for (int k = 0; k < NW; ++k) lv += lp[k];
  const float loss1 = (float)(lv / (double)N);
  const float loss0 = L == 0 ? loss1 : m.scal[chain * 4 + 2];
  const float U0 = pyz_hm_potential(m.scal[chain * 4 + 1], loss0, n_train);
  const float K0 = (1.0f / (2.0f * a.m)) * m.scal[chain * 4 + 0];
  const float U1 = pyz_hm_potential(slp_1, loss1, n_train);
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

// ---------------------------------------------------------------- the sliced proposal as ONE launch
// k_hmc_multi pays a kernel boundary per gradient evaluation (22 launches x 9.4 us at C3).  Here the NW workgroups of a
// chain stay resident for the whole proposal: the data slice is staged once, q and p live in every workgroup's LDS
// (each applies the same kick / drift to its own copy), and between two gradient evaluations the workgroups exchange
// their partial gradients through memory as data-tagged granules (cdna_hip_programming.md, Guideline 16, R2: "the data
// IS the flag"): element e of slice k of evaluation ph is ONE aligned 8-byte write-through store {tag, value} with
// tag = the chain's epoch + ph + 1, and a consumer re-reads the granules it needs (sc1 loads) until every tag matches.
// No counter, no flag, no fence: a store is visible or it is not, and a stale granule has a smaller tag.
// (A first version followed the counter hand-off -- drain the stores, barrier, atomic add, poll, barrier, sc1 loads --
// and was no faster than one launch per evaluation: three dependent trips to the fabric per evaluation, 9.3 us.)
// Granule rows ping-pong by evaluation parity: a workgroup that has consumed evaluation ph + 1 knows that every other
// one has finished reading evaluation ph.  The chain's epoch lives in device memory and is advanced by workgroup 0 at
// the end of the proposal (the next launch reads it behind the kernel boundary): tags grow monotonically over the life
// of the buffer, which is zeroed when it is allocated.
// The sums keep k_hmc_multi's order (slices 0 .. NW-1), so both forms give the same bits.
// The grid must be resident at once (the launcher checks NW x chains against the device); a wave whose sweep makes
// `spin_limit` passes without completing gives up: the proposal is marked (stats[7] = -1), q stays, every workgroup
// returns -- the grid always drains.
typedef __attribute__((address_space(1))) unsigned long long pyz_gu64;

__device__ __forceinline__ void pyz_hm_store_granule(unsigned long long *g, const unsigned tag, const float v) {
  __hip_atomic_store((pyz_gu64 *)g, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long pyz_hm_load_granule(const unsigned long long *g) {
  return __hip_atomic_load((pyz_gu64 *)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MI, int MC, int ACT>
__global__ void __launch_bounds__(PYZ_HM_THREADS) k_hmc_resident(HmcMultiArgs m) {
  extern __shared__ float lds[];
  const HmcFusedArgs &a = m.f;
  const int D = a.D, N = a.N, I = a.I, C = a.C, L = a.L, Dp = m.Dp;
  const int t = threadIdx.x, wg = blockIdx.x, chain = blockIdx.y, P = gridDim.y, NW = m.NW;
  const int r0 = (int)(((long long)N * wg) / NW), r1 = (int)(((long long)N * (wg + 1)) / NW), nloc = r1 - r0;
  float *q = lds, *p = q + D, *g = p + D, *part = g + D;
  float *wj = part + PYZ_HM_WAVES * D, *xs = wj + 64 * (MI + MC + 2), *d2 = xs + m.max_rows * MI, *yf = d2 + m.max_rows * MC;
  const size_t fl = (size_t)(3 + PYZ_HM_WAVES) * D + (size_t)64 * (MI + MC + 2) + (size_t)m.max_rows * MI +
                    (size_t)m.max_rows * MC + (size_t)m.max_rows * (a.loss == PYZ_LOSS_MSE ? C : 1);
  double *sm = reinterpret_cast<double *>(lds + ((fl * 4 + 15) / 16) * 4);
  int *gave_up = reinterpret_cast<int *>(sm + 32);
  const long long so = (long long)chain * D;
  const unsigned epoch0 = m.epoch[chain];   // (written by the previous proposal's launch)
  // ---- this slice of the data set, once per proposal
  for (int e = t; e < nloc * MI; e += PYZ_HM_THREADS) {
    const int r = e / MI, i = e - r * MI;
    xs[e] = i < I ? a.x[(long long)(r0 + r) * I + i] : 0.0f;
  }
  if (a.loss == PYZ_LOSS_SCCE) {
    for (int e = t; e < nloc; e += PYZ_HM_THREADS) yf[e] = __int_as_float(reinterpret_cast<const int32_t *>(a.y)[r0 + e]);
  } else {
    for (int e = t; e < nloc * C; e += PYZ_HM_THREADS) yf[e] = reinterpret_cast<const float *>(a.y)[(long long)r0 * C + e];
  }
  // ---- the starting point: q, a fresh momentum, and (workgroup 0) K0 and the prior part of U0
  const float ls = logf(a.prior_sigma);
  float sp2_0 = 0.0f, slp_0 = 0.0f;
  {
    double sp2 = 0.0, slp = 0.0;
    const HmcCall call = *m.call;
    for (int e = t; e < D; e += PYZ_HM_THREADS) {
      const float qv = a.q[so + e], pv = pyz_hm_momentum(a, call, chain, e);
      q[e] = qv;
      p[e] = pv;
      sp2 += (double)(pv * pv);
      slp += (double)pyz_hm_log_prior(qv, a.prior_mean, a.prior_sigma, ls);
    }
    if (t == 0) *gave_up = 0;
    if (wg == 0) {  // uniform per workgroup
      sp2_0 = (float)pyz_hf_block_sum<PYZ_HM_WAVES>(sp2, sm);
      slp_0 = (float)pyz_hf_block_sum<PYZ_HM_WAVES>(slp, sm);
    }
  }
  const float eps = a.epsilon, drift = eps / a.m, n_train = (float)N;
  const float isig2 = 1.0f / (a.prior_sigma * a.prior_sigma);
  float loss0 = 0.0f, loss1 = 0.0f;
  double sp2_1 = 0.0, slp_1 = 0.0;
#ifdef PYZ_STAMPS
  unsigned long long lap[16] = {0};
  PYZ_LAP(lap, 8);
  lap[8] = 0;
#else
  unsigned long long *lap = nullptr;
#endif
  __syncthreads();
  for (int ph = 0; ph <= L; ++ph) {
    PYZ_LAP(lap, 7);
    const unsigned tag = epoch0 + (unsigned)ph + 1u;
    // ---- gradient of the mean loss over this slice of the rows, published as granule row (ph & 1, chain, wg)
    double lsum = 0.0;
#ifdef PYZ_HMC_DIAG
    if (!(m.diag & 2))
#endif
    lsum = pyz_hf_loss_grad<MI, MC, ACT, PYZ_HM_WAVES>(a, q, g, part, wj, xs, d2, yf, sm, nloc, lap, ph > 0);
    unsigned long long *row = m.gran + ((((long long)(ph & 1) * P + chain) * NW) + wg) * Dp;
    const unsigned long long lbits = __builtin_bit_cast(unsigned long long, lsum);
    for (int e = t; e < Dp; e += PYZ_HM_THREADS) {
      float v = e < D ? g[e] : 0.0f;
      if (e == Dp - 2) v = __uint_as_float((unsigned)(lbits & 0xffffffffull));
      if (e == Dp - 1) v = __uint_as_float((unsigned)(lbits >> 32));
      if (e < D || e >= Dp - 2) pyz_hm_store_granule(row + e, tag, v);
    }
    PYZ_LAP(lap, 5);
    if (ph == L && wg != 0) return;   // (uniform) only workgroup 0 closes the proposal
    // ---- element e of the summed gradient: the NW slices in order, sixteen granules per sweep
    const unsigned long long *rows = m.gran + (((long long)(ph & 1) * P + chain) * NW) * Dp;
    for (int e0 = 0; e0 < D; e0 += PYZ_HM_THREADS) {
      const int e = e0 + t;
      const bool mine = e < D;
      const int ec = mine ? e : 0;
      float gs = 0.0f;
      for (int k0 = 0; k0 < NW; k0 += 16) {
        float v[16];
        for (int spins = 0;;) {
          bool ok = true;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const unsigned long long x = pyz_hm_load_granule(rows + (long long)min(k0 + j, NW - 1) * Dp + ec);
            v[j] = __uint_as_float((unsigned)x);
            ok &= (unsigned)(x >> 32) == tag;
          }
#ifdef PYZ_HMC_DIAG
          if (m.diag & 1) ok = true;
#endif
          if (__all(ok || !mine)) break;
          if (++spins > m.spin_limit) {   // (wave-uniform)
            *gave_up = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) gs += k0 + j < NW ? v[j] : 0.0f;
      }
      if (mine) {
        const float qv = q[e];
        const float dU = pyz_hm_dU(qv, a.prior_mean, isig2, n_train, gs);
        if (ph == L) {   // the last kick and the closing half kick share the last gradient (L == 0: both half kicks)
          float pv = pyz_hm_kick(p[e], L == 0 ? eps / 2 : eps, dU);
          pv = pyz_hm_kick(pv, eps / 2, dU);
          sp2_1 += (double)(pv * pv);
          slp_1 += (double)pyz_hm_log_prior(qv, a.prior_mean, a.prior_sigma, ls);
        } else {         // half kick behind the first gradient (HMC.py:82), kick behind the others (HMC.py:84-86); then the drift
          const float pv = pyz_hm_kick(p[e], ph == 0 ? eps / 2 : eps, dU);
          const float qn = pyz_hm_drift(qv, drift, pv);
          p[e] = pv;
          q[e] = qn;
          const int slot = pyz_hf_wj_slot<MI, MC>(e, I, a.H, C);   // the next evaluation finds its weight records current
          if (slot >= 0) wj[slot] = qn;
        }
      }
    }
    PYZ_LAP(lap, 6);
    if (wg == 0 && (ph == 0 || ph == L) && t < 64) {   // the loss at the starting point (U0) / at the proposal (U1): wave 0
      // lane k holds slice k's sum (NW <= 32 <= 64 lanes); summed in slice order by lane 0
      const int k = min(t, NW - 1);
      unsigned long long lo = 0, hi = 0;
      for (int spins = 0;;) {
        lo = pyz_hm_load_granule(rows + (long long)k * Dp + Dp - 2);
        hi = pyz_hm_load_granule(rows + (long long)k * Dp + Dp - 1);
        bool ok = (unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag;
#ifdef PYZ_HMC_DIAG
        if (m.diag & 1) ok = true;
#endif
        if (__all(ok)) break;
        if (++spins > m.spin_limit) {
          *gave_up = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      const double mine_sum = __builtin_bit_cast(double, ((hi & 0xffffffffull) << 32) | (lo & 0xffffffffull));
      double v = 0.0;
      for (int kk = 0; kk < NW; ++kk) v += __shfl(mine_sum, kk, 64);
      const float lv = (float)(v / (double)N);
      if (ph == 0) loss0 = lv;
      if (ph == L) loss1 = lv;
    }
    __syncthreads();
    if (*gave_up) {   // (uniform) the others never showed up: not resident together, or one of them gave up
      if (t == 0 && wg == 0) {
        a.stats[chain * 8 + 7] = -1.0f;
        m.epoch[chain] = epoch0 + (unsigned)L + 1u;
      }
      return;
    }
  }
#ifdef PYZ_STAMPS
  if (t == 0 && chain == 0)
    for (int k = 0; k < 8; ++k) pyz_dbg_buf[3][0][0][k][0] = lap[k];
#endif
  // ---- workgroup 0: energies, Metropolis test, write-back (k_hmc_multi_final)
  loss0 = __shfl(loss0, 0, 64);   // (lane 0 of wave 0 holds them; thread 0 is the only reader below)
  loss1 = __shfl(loss1, 0, 64);
  const float s1 = (float)pyz_hf_block_sum<PYZ_HM_WAVES>(sp2_1, sm);
  const float l1 = (float)pyz_hf_block_sum<PYZ_HM_WAVES>(slp_1, sm);
  const float U0 = pyz_hm_potential(slp_0, loss0, n_train);
  const float K0 = (1.0f / (2.0f * a.m)) * sp2_0;
  const float U1 = pyz_hm_potential(l1, loss1, n_train);
  const float K1 = (1.0f / (2.0f * a.m)) * s1;
  const float lr = K0 + U0 - K1 - U1;
  // every thread needs the decision: thread 0 has the losses, the others get it through LDS
  if (t == 0) {
    const bool acc0 = m.call->burning || (a.uniform[chain] < expf(lr));
    *gave_up = acc0 ? 2 : 0;
  }
  __syncthreads();
  const bool acc = *gave_up == 2;
  if (acc)
    for (int e = t; e < D; e += PYZ_HM_THREADS) a.q[so + e] = q[e];
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
    m.epoch[chain] = epoch0 + (unsigned)L + 1u;
  }
}
