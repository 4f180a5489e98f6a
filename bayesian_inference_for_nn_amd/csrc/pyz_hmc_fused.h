// pyz_hmc_fused.h -- one HMC proposal of a small 2-layer MLP inside ONE workgroup.
//
// HMC.step (Pyesian/optimizers/HMC.py:74-104) evaluates the full-batch gradient L+2 times
// per proposal; for the reference's HMC models (make_moons 2->50->2, D = 252, N = 1600:
// HMC_classification.py:36-50) a gradient is ~1.6 MFLOP, far too small to spread over the
// chip, and a launch per kernel (5 per gradient, 22 gradients per sample) costs more than
// the arithmetic.  Here one 1024-thread workgroup owns one chain for the whole proposal:
// data, q, p and the gradient live in LDS, the leapfrog loop never leaves the CU, and a
// grid of P workgroups runs P independent chains (one per CU).
//
//   phase A  thread <-> data row (two rows per pass so each LDS weight read feeds two rows):
//            forward, softmax/MSE, per-row loss, delta_out -> LDS
//   phase B  lane <-> hidden unit j (its weights in registers), wave <-> slice of the rows:
//            recompute h_j, accumulate dW2[j][:], dW1[:][j], db1[j]; lane H+c accumulates db2[c];
//            the 16 per-wave partials are combined in a fixed order
// Supported: dims (I, H, C) with I <= 8, H <= 56, C <= 8 (H + C <= 64), hidden activation
// relu/tanh/sigmoid/linear, softmax+SCCE or MSE output.  Anything else uses the generic path.
#pragma once

#include <type_traits>

#include "pyz_common.h"
#include "pyz_gemm.h"
#include "pyz_kernels.h"
#include "pyz_rng.h"

#define PYZ_HF_MAXI 8
#define PYZ_HF_MAXC 8
#ifndef PYZ_HF_THREADS
#define PYZ_HF_THREADS 1024
#define PYZ_HF_WAVES 16
#endif

struct HmcFusedArgs {
  float *q;                // (P, D) in/out
  const float *x;          // (N, I)
  const void *y;           // int32 (N) or float (N, C)
  int N, I, H, C, D;
  int act_hidden, act_last, loss;
  int L;
  float epsilon, m, prior_mean, prior_sigma;
  int burning;
  const float *uniform;    // (P) device copy of the host uniforms
  uint64_t seed;
  uint32_t step;
  const float *unit_p;     // optional (P, D)
  float *stats;            // (P, 8)
};

// LDS carve (floats): q[D] p[D] g[D] qsave[D] part[16][D] wj[64*(MI+MC+2)] xs[N*MI] d2[N*MC]
// yf[N*C or N] + 64 doubles.  Rows of xs / d2 and the per-hidden-unit weight records wj are
// padded with zeros to the compile-time widths MI / MC, so the hot loops have no bounds
// checks (a runtime `i < I` guard becomes a branch around every LDS read).
static inline size_t pyz_hmc_fused_floats(int N, int MI, int MC, int C, int D, int loss) {
  return (size_t)(4 + PYZ_HF_WAVES) * D + (size_t)64 * (MI + MC + 2) + (size_t)N * MI + (size_t)N * MC +
         (size_t)N * (loss == PYZ_LOSS_MSE ? C : 1);
}
static inline size_t pyz_hmc_fused_lds_bytes(int N, int MI, int MC, int C, int D, int loss) {
  return ((pyz_hmc_fused_floats(N, MI, MC, C, D, loss) * 4 + 15) / 16) * 16 + 64 * sizeof(double);
}

template <int WAVES = PYZ_HF_WAVES>
__device__ __forceinline__ double pyz_hf_block_sum(double v, double *sm) {
  v = pyz_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < WAVES; ++i) s += sm[i];
  // The total is pinned HERE: left alone, the compiler keeps the WAVES partials of an early call in registers
  // and sinks their sum to the first use of the result -- for the energies of k_hmc_fused that is the
  // Metropolis test at the very end, 32 registers per call held across the whole leapfrog loop (and spilled:
  // .vgpr_spill_count 44-149 before this line, 0 after).
  asm volatile("" : "+v"(s));
  return s;  // every thread gets the same total
}

// loss term of one row and its delta_out (SCCE on softmax logits / MSE), from the logits z[]
template <int MC>
__device__ __forceinline__ double pyz_hf_row_tail(const HmcFusedArgs &a, const float (&z)[MC], float *d, const float *yf,
                                                  const int r) {
  const int N = a.N, C = a.C;
  if (a.loss == PYZ_LOSS_SCCE) {
    const int y = __float_as_int(yf[r]);
    float mx = z[0];
#pragma unroll
    for (int c = 1; c < MC; ++c)
      if (c < C) mx = fmaxf(mx, z[c]);
    float se = 0.0f;
#pragma unroll
    for (int c = 0; c < MC; ++c)
      if (c < C) se += expf(z[c] - mx);
    const float lse = mx + logf(se);
    float zy = __builtin_nanf("");
#pragma unroll
    for (int c = 0; c < MC; ++c)
      if (c < C && c == y) zy = z[c];
    const float inv = 1.0f / (float)N;
#pragma unroll
    for (int c = 0; c < MC; ++c) d[c] = c < C ? (expf(z[c] - lse) - (c == y ? 1.0f : 0.0f)) * inv : 0.0f;
    return (double)(lse - zy);
  }
  const float sc = 2.0f / ((float)N * (float)C);
  float acc = 0.0f;
#pragma unroll
  for (int c = 0; c < MC; ++c) {
    float dv = 0.0f;
    if (c < C) {
      const float o = pyz_act(z[c], a.act_last);
      const float e = o - yf[r * C + c];
      acc += e * e;
      dv = sc * e * pyz_act_grad(o, a.act_last);
    }
    d[c] = dv;
  }
  return (double)(acc / (float)C);
}

// where element e of the parameter vector sits in the weight records wj (-1: b2, which has no record)
template <int MI, int MC>
__device__ __forceinline__ int pyz_hf_wj_slot(const int e, const int I, const int H, const int C) {
  constexpr int SJ = MI + MC + 2;
  if (e < I * H) return (e % H) * SJ + e / H;
  const int e1 = e - I * H;
  if (e1 < H) return e1 * SJ + MI + MC;
  const int e2 = e1 - H;
  if (e2 < H * C) return (e2 / C) * SJ + MI + e2 % C;
  return -1;
}

// sum of the row losses over the nloc rows staged in xs / yf, and the gradient of (that sum / N) into
// g[] at the weights q[] (both LDS).  WAVES = waves of the workgroup (16: the whole data set in one
// workgroup, k_hmc_fused; 4: a row slice, k_hmc_multi).
// MI / MC = compile-time widths the rows are padded to; ACT = the hidden activation, a template
// parameter so that the inner loops carry exactly one activation's code.
// wj[j] = { W1[0..MI)[j], W2[j][0..MC), b1[j], 0 } is rebuilt from q at every call.
template <int MI, int MC, int ACT, int WAVES = PYZ_HF_WAVES>
__device__ double pyz_hf_loss_grad(const HmcFusedArgs &a, const float *q, float *g, float *part, float *wj,
                                   const float *xs, float *d2, const float *yf, double *sm, const int nloc,
                                   unsigned long long *lap = nullptr, const bool wj_ready = false) {
  constexpr int SJ = MI + MC + 2;
  constexpr int THREADS = 64 * WAVES;
  const int N = nloc, I = a.I, H = a.H, C = a.C;
  const int t = threadIdx.x, w = pyz_wave_id(), l = t & 63;  // w scalar: phase B's row slice lives in SGPRs
  const float *W1 = q, *b1 = q + I * H, *W2 = b1 + H, *b2 = W2 + H * C;
  // wj_ready (uniform): the caller keeps the weight records current itself (pyz_hf_wj_slot) and has had a barrier since
  if (!wj_ready) {
    for (int e = t; e < H * SJ; e += THREADS) {
      const int j = e / SJ, k = e - j * SJ;
      float v = 0.0f;
      if (k < MI) v = k < I ? W1[k * H + j] : 0.0f;
      else if (k < MI + MC) v = (k - MI) < C ? W2[j * C + (k - MI)] : 0.0f;
      else if (k == MI + MC) v = b1[j];
      wj[e] = v;
    }
    __syncthreads();
  }
  if (lap) PYZ_LAP(lap, 0);
  // ---------------- phase A
  double lsum = 0.0;
  if constexpr (WAVES <= 8) {
    // row slices (a few dozen rows on 256 / 512 threads): LPR lanes per row, each takes 1 / LPR of the hidden units, and the
    // partial logits are added across the lanes of a row (pairwise tree; a + b on one lane and b + a on its partner are
    // the same bits).  One lane per row leaves most of the workgroup idle behind a loop over all H units
    // (k_hmc_resident: 4 600 of 19 000 cycles per gradient evaluation).
    constexpr int LPR = WAVES <= 4 ? 2 : 4;
    const int sub = t & (LPR - 1), Hq = (H + LPR - 1) / LPR;
    const int jb = min(sub * Hq, H), je = min(jb + Hq, H);
    for (int rbase = 0; rbase < N; rbase += THREADS / LPR) {
      const int r = rbase + t / LPR;
      const bool has = r < N;
      const int rc = has ? r : N - 1;   // (spare lanes redo the last row: same values to the same places)
      float x0[MI], z0[MC];
#pragma unroll
      for (int i = 0; i < MI; ++i) x0[i] = xs[rc * MI + i];
#pragma unroll
      for (int c = 0; c < MC; ++c) z0[c] = (sub == 0 && c < C) ? b2[c] : 0.0f;
#pragma unroll 5
      for (int j = jb; j < je; ++j) {
        const float *rec = wj + j * SJ;
        float h0 = rec[MI + MC];
#pragma unroll
        for (int i = 0; i < MI; ++i) h0 = fmaf(x0[i], rec[i], h0);
        h0 = pyz_act(h0, ACT);
#pragma unroll
        for (int c = 0; c < MC; ++c) z0[c] = fmaf(h0, rec[MI + c], z0[c]);
      }
#pragma unroll
      for (int o = 1; o < LPR; o <<= 1) {
#pragma unroll
        for (int c = 0; c < MC; ++c) z0[c] = z0[c] + __shfl_xor(z0[c], o, 64);
      }
      const double tail = pyz_hf_row_tail<MC>(a, z0, d2 + rc * MC, yf, rc);
      if (has && sub == 0) lsum += tail;
    }
  } else {
  // rows (two per pass: every weight record read feeds two rows)
  for (int r0 = t; r0 < N; r0 += 2 * THREADS) {
    const int r1 = r0 + THREADS;
    const bool has1 = r1 < N;
    const int r1c = has1 ? r1 : r0;
    float x0[MI], x1[MI], z0[MC], z1[MC];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      x0[i] = xs[r0 * MI + i];
      x1[i] = xs[r1c * MI + i];
    }
#pragma unroll
    for (int c = 0; c < MC; ++c) z0[c] = z1[c] = c < C ? b2[c] : 0.0f;
    constexpr int UJ = (MI + MC > 8) ? 1 : 2;
#pragma unroll UJ
    for (int j = 0; j < H; ++j) {
      const float *rec = wj + j * SJ;
      float h0 = rec[MI + MC], h1 = h0;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        h0 = fmaf(x0[i], rec[i], h0);
        h1 = fmaf(x1[i], rec[i], h1);
      }
      h0 = pyz_act(h0, ACT);
      h1 = pyz_act(h1, ACT);
#pragma unroll
      for (int c = 0; c < MC; ++c) {
        z0[c] = fmaf(h0, rec[MI + c], z0[c]);
        z1[c] = fmaf(h1, rec[MI + c], z1[c]);
      }
    }
    lsum += pyz_hf_row_tail<MC>(a, z0, d2 + r0 * MC, yf, r0);
    if (has1) lsum += pyz_hf_row_tail<MC>(a, z1, d2 + r1 * MC, yf, r1);
  }
  }
  if (lap) PYZ_LAP(lap, 1);
  // the loss: per-wave sums now, the total behind the barrier that follows phase B (pyz_hf_block_sum's values and order, one
  // barrier less); this barrier also orders d2[] before phase B
  {
    const double wsum = pyz_wave_sum(lsum);
    if (l == 0) sm[w] = wsum;
  }
  __syncthreads();
  if (lap) PYZ_LAP(lap, 2);
  PYZ_STAMP(3, 5);
  // ---------------- phase B: lane <-> hidden unit, wave <-> row slice
  {
    const int j = l;
    const bool is_h = j < H;
    const int cb = j - H;  // lanes H .. H+C-1 own db2[cb]
    const bool is_b2 = cb >= 0 && cb < C;
    const int cbc = is_b2 ? cb : 0;
    float w1[MI], w2[MC], gw1[MI], gw2[MC];
    float gb1 = 0.0f, gb2 = 0.0f;
    const float *rec = wj + (is_h ? j : 0) * SJ;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      w1[i] = rec[i];
      gw1[i] = 0.0f;
    }
#pragma unroll
    for (int c = 0; c < MC; ++c) {
      w2[c] = rec[MI + c];
      gw2[c] = 0.0f;
    }
    const float bj = rec[MI + MC];
    const int rb = (int)(((long long)N * w) / WAVES), re = (int)(((long long)N * (w + 1)) / WAVES);
    // RB rows per trip: their LDS reads are issued together (one base address, constant offsets) and the dependent chains
    // (pre -> h -> dh -> dpre) interleave instead of serialising on LDS latency.  Whole trips carry no row masks or clamps;
    // the rows left over go one at a time.  (k_hmc_resident: a masked, clamped trip of four rows was ~200 instructions,
    // 6 500 of 17 500 cycles per gradient evaluation.)  Eight rows where the registers allow it (the 256-thread slice
    // kernels on a narrow model), two for the widest instantiation.
    // (the widest records in the sixteen-wave kernel, 128 registers per lane: one row per trip -- with two, k_hmc_fused<8, 8, relu>
    //  spilled two registers to scratch)
    constexpr int RB = (MI + MC > 12 && WAVES > 8) ? 1 : (MI + MC > 8) ? 2 : ((WAVES <= 8 && MI + MC <= 4) ? 8 : 4);
    auto row_terms = [&](const float (&xv)[MI], const float (&dv)[MC]) {
      float pre = bj, dh = 0.0f;
#pragma unroll
      for (int i = 0; i < MI; ++i) pre = fmaf(xv[i], w1[i], pre);
      const float h = pyz_act(pre, ACT);
#pragma unroll
      for (int c = 0; c < MC; ++c) {
        gw2[c] = fmaf(h, dv[c], gw2[c]);
        dh = fmaf(w2[c], dv[c], dh);
      }
      const float dpre = dh * pyz_act_grad(h, ACT);
#pragma unroll
      for (int i = 0; i < MI; ++i) gw1[i] = fmaf(xv[i], dpre, gw1[i]);
      gb1 += dpre;
      float dsel = dv[0];   // lanes H .. H+C-1: column cb of delta_out
#pragma unroll
      for (int c = 1; c < MC; ++c) dsel = cbc == c ? dv[c] : dsel;
      gb2 += dsel;
    };
    int r = rb;
    auto trips = [&](auto nrows) {   // whole trips of R rows, in row order
      constexpr int R = decltype(nrows)::value;
      for (; r + R <= re; r += R) {
        float xv[R][MI], dv[R][MC];
        const float *xr = xs + r * MI, *dr = d2 + r * MC;
#pragma unroll
        for (int u = 0; u < R; ++u) {
#pragma unroll
          for (int i = 0; i < MI; ++i) xv[u][i] = xr[u * MI + i];
#pragma unroll
          for (int c = 0; c < MC; ++c) dv[u][c] = dr[u * MC + c];
        }
#pragma unroll
        for (int u = 0; u < R; ++u) row_terms(xv[u], dv[u]);
      }
    };
    trips(std::integral_constant<int, RB>());
    if constexpr (RB > 4) trips(std::integral_constant<int, 4>());
    if constexpr (RB > 2) trips(std::integral_constant<int, 2>());
    trips(std::integral_constant<int, 1>());
    float *pw = part + w * a.D;
    if (is_h) {
      for (int i = 0; i < I; ++i) pw[i * H + j] = gw1[i];
      pw[I * H + j] = gb1;
      for (int c = 0; c < C; ++c) pw[I * H + H + j * C + c] = gw2[c];
    }
    if (is_b2) pw[I * H + H + H * C + cb] = gb2;
  }
  PYZ_STAMP(3, 6);
  if (lap) PYZ_LAP(lap, 3);
  __syncthreads();
  double loss = 0.0;
#pragma unroll
  for (int i = 0; i < WAVES; ++i) loss += sm[i];
  asm volatile("" : "+v"(loss));   // (pinned here: see pyz_hf_block_sum)
  for (int e = t; e < a.D; e += THREADS) {
    float s = part[e];
#pragma unroll
    for (int ww = 1; ww < WAVES; ++ww) s += part[ww * a.D + e];
    g[e] = s;
  }
  __syncthreads();
  if (lap) PYZ_LAP(lap, 4);
  return loss;
}

template <int MI, int MC, int ACT>
__global__ void __launch_bounds__(PYZ_HF_THREADS) k_hmc_fused(HmcFusedArgs a) {
  extern __shared__ float lds[];
  const int D = a.D, N = a.N, I = a.I, C = a.C;
  const int t = threadIdx.x, chain = blockIdx.x;
  float *q = lds, *p = q + D, *g = p + D, *qs = g + D, *part = qs + D;
  float *wj = part + PYZ_HF_WAVES * D, *xs = wj + 64 * (MI + MC + 2), *d2 = xs + N * MI, *yf = d2 + N * MC;
  const size_t fl = (size_t)(4 + PYZ_HF_WAVES) * D + (size_t)64 * (MI + MC + 2) + (size_t)N * MI + (size_t)N * MC +
                    (size_t)N * (a.loss == PYZ_LOSS_MSE ? C : 1);
  double *sm = reinterpret_cast<double *>(lds + ((fl * 4 + 15) / 16) * 4);
  float *qg = a.q + (long long)chain * D;
  PYZ_STAMP(3, 0);
  // ---- stage the data set, the labels and q; draw the momentum (HMC.py:168-171: p = m z)
  for (int e = t; e < N * MI; e += PYZ_HF_THREADS) {
    const int r = e / MI, i = e - r * MI;
    xs[e] = i < I ? a.x[r * I + i] : 0.0f;
  }
  if (a.loss == PYZ_LOSS_SCCE) {
    for (int e = t; e < N; e += PYZ_HF_THREADS) yf[e] = __int_as_float(reinterpret_cast<const int32_t *>(a.y)[e]);
  } else {
    for (int e = t; e < N * C; e += PYZ_HF_THREADS) yf[e] = reinterpret_cast<const float *>(a.y)[e];
  }
  double sp2 = 0.0, slp = 0.0;
  const float ls = logf(a.prior_sigma);
  for (int e = t; e < D; e += PYZ_HF_THREADS) {
    const float qv = qg[e];
    float z;
    if (a.unit_p) {
      z = a.unit_p[(long long)chain * D + e];
    } else {
      const float4 v = pyz_normal4(a.seed, PYZ_STREAM_HMC + 16u * (uint32_t)chain, a.step, (uint64_t)(e >> 2));
      const int k = e & 3;
      z = k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w));
    }
    const float pv = a.m * z;
    q[e] = qv;
    qs[e] = qv;
    p[e] = pv;
    sp2 += (double)(pv * pv);
    const float u = (qv - a.prior_mean) / a.prior_sigma;
    slp += (double)(-0.5f * u * u - ls - PYZ_LOG_SQRT_2PI);
  }
  const float sp2_0 = (float)pyz_hf_block_sum(sp2, sm);
  const float slp_0 = (float)pyz_hf_block_sum(slp, sm);  // (its barriers also publish xs / yf / q / p)
  const float n_train = (float)N;
  const float isig2 = 1.0f / (a.prior_sigma * a.prior_sigma);
  PYZ_STAMP(3, 1);
  // ---- U0, K0 and the first gradient (HMC.py:79-82)
  const float loss0 = (float)(pyz_hf_loss_grad<MI, MC, ACT>(a, q, g, part, wj, xs, d2, yf, sm, N) / (double)N);
  PYZ_STAMP(3, 2);
  float U0 = 0.0f - slp_0;
  U0 = U0 + loss0 * n_train;
  const float K0 = (1.0f / (2.0f * a.m)) * sp2_0;
  float loss1 = loss0;
  // ---- leapfrog (HMC.py:82-87): half kick, L x (drift, kick), half kick sharing the last gradient
  const float eps = a.epsilon, drift = eps / a.m;
  for (int e = t; e < D; e += PYZ_HF_THREADS) {
    const float dU = (q[e] - a.prior_mean) * isig2 + n_train * g[e];
    float pv = p[e] - (eps / 2) * dU;
    if (a.L == 0) pv = pv - (eps / 2) * dU;
    p[e] = pv;
    if (a.L > 0) q[e] = q[e] + drift * pv;
  }
  __syncthreads();
  for (int it = 1; it <= a.L; ++it) {
    loss1 = (float)(pyz_hf_loss_grad<MI, MC, ACT>(a, q, g, part, wj, xs, d2, yf, sm, N) / (double)N);
    for (int e = t; e < D; e += PYZ_HF_THREADS) {
      const float dU = (q[e] - a.prior_mean) * isig2 + n_train * g[e];
      float pv = p[e] - eps * dU;
      if (it == a.L) {
        pv = pv - (eps / 2) * dU;
        p[e] = pv;
      } else {
        p[e] = pv;
        q[e] = q[e] + drift * pv;
      }
    }
    __syncthreads();
  }
  PYZ_STAMP(3, 3);
  // ---- K1, U1 (the loss of the last gradient evaluation is the loss at the proposal)
  sp2 = 0.0;
  slp = 0.0;
  for (int e = t; e < D; e += PYZ_HF_THREADS) {
    const float pv = p[e];
    sp2 += (double)(pv * pv);
    const float u = (q[e] - a.prior_mean) / a.prior_sigma;
    slp += (double)(-0.5f * u * u - ls - PYZ_LOG_SQRT_2PI);
  }
  const float sp2_1 = (float)pyz_hf_block_sum(sp2, sm);
  const float slp_1 = (float)pyz_hf_block_sum(slp, sm);
  float U1 = 0.0f - slp_1;
  U1 = U1 + loss1 * n_train;
  const float K1 = (1.0f / (2.0f * a.m)) * sp2_1;
  // ---- Metropolis test (HMC.py:91) and write-back
  const float lr = K0 + U0 - K1 - U1;
  const bool acc = a.burning || (a.uniform[chain] < expf(lr));
  if (acc)
    for (int e = t; e < D; e += PYZ_HF_THREADS) qg[e] = q[e];
  PYZ_STAMP(3, 4);
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
