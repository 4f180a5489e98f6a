// pyz_fused.h -- fused kernels that cut a gradient step down to three launches.
//
// Every kernel of this path is latency-bound (the 784->200->10 step is
// 654 MFLOP / 7 MB, all L2 resident), so what counts is the number of launches and of
// dependent memory round trips, not FLOP/s.  A step of an L-layer MLP is
//     k_dense_fwd            x (L-1)   hidden layers
//     k_head                 x 1       last layer + loss + delta_L + delta_{L-1}
//     k_dense_bwd_data       x (L-2)   remaining data gradients
//     k_wgrad_all            x 1       every layer's [dW; db] + the optimizer update
// i.e. 3 launches for the 2-layer headline model instead of 7.
#pragma once

#include "pyz_common.h"
#include "pyz_gemm.h"
#include "pyz_kernels.h"

// ---------------------------------------------------------------- head
// One workgroup (4 waves) owns 32 batch rows of the LAST Dense layer (N <= 32):
//   1. z = h_in W + b           (MFMA, the 4 waves split K, LDS combine)
//   2. loss row terms, delta_L  (SparseCategoricalCrossentropy / MeanSquaredError as
//                                in k_loss_scce / k_loss_mse: Dataset.py:152-159)
//   3. delta_{L-1} = (delta_L W^T) * act'(h_in)   (MFMA with delta_L from LDS)
struct HeadArgs {
  const float *hin;          // (P, max_batch, K) input of the last layer, or the data x when L == 1
  long long hin_pstride;
  int lda;
  const int32_t *row_idx;    // gather of hin rows (only when hin is the data, L == 1) and of the labels
  int gather_hin;
  const float *theta;
  long long theta_pstride;
  long long w_off;
  int K, N;
  int loss, act_last, act_prev, vec;
  const void *y;
  float *out_last;           // optional (P, max_batch, N): logits (softmax layer) / outputs
  float *delta_last;         // (P, max_batch, N) or nullptr (loss only)
  float *delta_prev;         // (P, max_batch, K) or nullptr
  long long last_pstride, prev_pstride;
  double *part;              // (P, nblk) per-workgroup sums of the row losses
  int nblk;
  const StepCtl *ctl;
  StepCtl init;              // the step scalars by value when the head is the first kernel of an eager step (L == 1)
  int init_on;
  int wt;                    // write-through stores for the outputs (pyz_st)
  const StepCtl *gate;       // when set: nothing happens on steps with gate->n % gate_mod == 0 (see DenseArgs)
  int gate_mod;
  // k_head_rows behind a split-reduction forward (k_dense_fwd_ring, k_split): the hidden layer arrives as n_hparts <= 8 raw
  // partial sums (hparts + s * hpart_stride + row * K + unit); the head adds them in split order, then the hidden layer's
  // bias (bias_prev[unit]) and activation (act_prev), and stores the row of activations to h_store for the weight gradients
  const float *hparts;
  int n_hparts;
  long long hpart_stride;
  const float *bias_prev;
  float *h_store;
};

// lanes 8q..8q+7 of a wave cooperate on one row: reductions over the 8-lane group
__device__ __forceinline__ float pyz_grp8_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64));
  v = fmaxf(v, __shfl_xor(v, 2, 64));
  return fmaxf(v, __shfl_xor(v, 4, 64));
}
__device__ __forceinline__ float pyz_grp8_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  return v + __shfl_xor(v, 4, 64);
}

// blockDim.x = 64 * S with S in {4, 8}; dynamic LDS = S * 4096 + 2 * 32 * 33 * 4 + 64 bytes
__global__ void __launch_bounds__(512) k_head(HeadArgs g) {
  extern __shared__ float lds[];
  PYZ_STAMP(1, 0);
  const int S = blockDim.x >> 6, w = pyz_wave_id(), l = threadIdx.x & 63, r = l & 31, h = l >> 5;
  if (g.gate && g.gate->n % g.gate_mod == 0) return;
  float *red = lds, *zt = lds + S * 1024, *dt = zt + 32 * 33;
  double *lsum = reinterpret_cast<double *>(dt + 32 * 33);  // 4 doubles (8-byte aligned: S*4096 + 8448 bytes)
  const StepCtl ctl = pyz_ctl_first(g.ctl, g.init, g.init_on);
  const int batch = ctl.batch, p = blockIdx.y, m0 = blockIdx.x * 32;
  if (m0 >= batch) {
    if (threadIdx.x == 0) g.part[p * g.nblk + blockIdx.x] = 0.0;
    return;
  }
  const int K = g.K, N = g.N;
  const int32_t *idx = g.row_idx ? g.row_idx + ctl.row_off : nullptr;
  // loss-phase inputs (threads 0..255: row = t >> 3), fetched now so their latency hides behind phase 1
  const int lrow = threadIdx.x >> 3, lsub = threadIdx.x & 7;
  const int lmm = m0 + lrow;
  const bool lvalid = threadIdx.x < 256 && lmm < batch;
  long long yrow = min(lmm, batch - 1);
  if (idx) yrow = idx[yrow];
  int ylab = 0;
  if (g.loss == PYZ_LOSS_SCCE) ylab = reinterpret_cast<const int32_t *>(g.y)[yrow];
  // delta_{L-1} operands of this wave's tile (jt = w): the act' inputs and the W rows depend on nothing
  // computed here, so they are fetched now and their latency hides behind phases 1 and 2
  const int tiles_j = (K + 31) >> 5;
  const int nsteps = (N + 1) >> 1;  // <= 16 (N <= 32)
  const float *wl = g.theta + p * g.theta_pstride + g.w_off;
  const float *hp = g.hin + p * g.hin_pstride;
  const bool pre = g.delta_prev && w < tiles_j && !g.gather_hin;
  float hv0[16], bw0[16];
  if (pre) {
    const int j = min(w * 32 + r, K - 1);
    const float *wp = wl + (long long)j * N;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int mm = min(m0 + (i & 3) + 8 * (i >> 2) + 4 * h, batch - 1);
      hv0[i] = hp[(long long)mm * K + j];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int kk = 2 * q + h;
      bw0[q] = (q < nsteps) ? wp[kk < N ? kk : 0] : 0.0f;
    }
  }
  {
    const int m = min(m0 + r, batch - 1), n = min(r, N - 1);
    long long row = m;
    if (g.gather_hin && idx) row = idx[m];
    const float *ap = g.hin + p * g.hin_pstride + row * g.lda;
    f32x16 acc = {0};
    PYZ_STAMP(1, 1);
    pyz_fwd_accumulate(acc, ap, wl, n, K, N, g.vec, w, S, h);
    PYZ_STAMP(1, 2);
    float *my = red + w * 1024;
#pragma unroll
    for (int i = 0; i < 16; ++i) my[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 1024; e += blockDim.x) {
    float v = red[e];
    for (int ww = 1; ww < S; ++ww) v += red[ww * 1024 + e];
    zt[(e >> 5) * 33 + (e & 31)] = v;
  }
  __syncthreads();
  PYZ_STAMP(1, 3);
  // ---- loss rows: 8 lanes per row, classes c = lsub, lsub + 8, ... (N <= 32)
  if (threadIdx.x < 256) {
    const float *z = zt + lrow * 33;
    float *d = dt + lrow * 33;
    float *ol = g.out_last ? g.out_last + p * g.last_pstride + (long long)lmm * N : nullptr;
    float *dl = g.delta_last ? g.delta_last + p * g.last_pstride + (long long)lmm * N : nullptr;
    float zv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) zv[q] = (lsub + 8 * q < N) ? z[lsub + 8 * q] : 0.0f;
    float lm = 0.0f;
    if (g.loss == PYZ_LOSS_SCCE) {
      float mx = -3.0e38f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (lsub + 8 * q < N) mx = fmaxf(mx, zv[q]);
      mx = pyz_grp8_max(mx);
      float se = 0.0f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (lsub + 8 * q < N) se += expf(zv[q] - mx);
      se = pyz_grp8_sum(se);
      const float lse = mx + logf(se);
      const float zy = (ylab >= 0 && ylab < N) ? z[ylab] : __builtin_nanf("");
      lm = lse - zy;
      const float inv = 1.0f / (float)batch;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = lsub + 8 * q;
        if (c < N) {
          const float dv = lvalid ? (expf(zv[q] - lse) - (c == ylab ? 1.0f : 0.0f)) * inv : 0.0f;
          d[c] = dv;
          if (lvalid && dl) dl[c] = dv;
          if (lvalid && ol) ol[c] = zv[q];
        }
      }
    } else {
      const float *y = reinterpret_cast<const float *>(g.y) + yrow * N;
      const float sc = 2.0f / ((float)batch * (float)N);
      float a = 0.0f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = lsub + 8 * q;
        if (c < N) {
          const float o = pyz_act(zv[q], g.act_last);
          const float e = o - y[c];
          a += e * e;
          const float dv = lvalid ? sc * e * pyz_act_grad(o, g.act_last) : 0.0f;
          d[c] = dv;
          if (lvalid && dl) dl[c] = dv;
          if (lvalid && ol) ol[c] = o;
        }
      }
      lm = pyz_grp8_sum(a) / (float)N;
    }
    const double ws = pyz_wave_sum((lvalid && lsub == 0) ? (double)lm : 0.0);
    if (l == 0) lsum[w] = ws;
  }
  __syncthreads();
  if (threadIdx.x == 0) g.part[p * g.nblk + blockIdx.x] = (lsum[0] + lsum[1]) + (lsum[2] + lsum[3]);
  PYZ_STAMP(1, 4);
  if (!g.delta_prev) return;
  // ---- delta_{L-1} tile by tile: (delta_L [32 x N]) (W^T [N x 32]) * act'(h_in)
  float *op = g.delta_prev + p * g.prev_pstride;
  for (int jt = w; jt < tiles_j; jt += S) {
    const int j0 = jt * 32, jj = j0 + r, j = min(jj, K - 1);
    const float *wp = wl + (long long)j * N;
    float hv[16];
    f32x16 acc = {0};
    if (pre && jt == w) {  // operands already in registers
#pragma unroll
      for (int i = 0; i < 16; ++i) hv[i] = hv0[i];
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (q < nsteps) {
          const int kk = 2 * q + h;
          const bool vk = kk < N;
          const float a = vk ? dt[r * 33 + kk] : 0.0f;
          acc = pyz_mfma(a, vk ? bw0[q] : 0.0f, acc);
        }
      PYZ_STAMP(1, 6);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int mm = min(m0 + (i & 3) + 8 * (i >> 2) + 4 * h, batch - 1);
        long long row = mm;
        if (g.gather_hin && idx) row = idx[mm];
        hv[i] = hp[row * K + j];
      }
      pyz_steps1_all(
          0, nsteps, acc,
          [&](int s, float &a, float &b) {
            const int kk = 2 * s + h;
            const int kc = kk < N ? kk : 0;
            a = dt[r * 33 + kc];
            b = wp[kc];
          },
          [&](int s, float &a, float &b) {
            const bool vk = 2 * s + h < N;
            a = vk ? a : 0.0f;
            b = vk ? b : 0.0f;
          });
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int mm = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (mm < batch && jj < K) op[(long long)mm * K + jj] = acc[i] * pyz_act_grad(hv[i], g.act_prev);
    }
  }
  PYZ_STAMP(1, 5);
}

// ---------------------------------------------------------------- head, one wave per batch row
// The last layer of these models is narrow (N <= 32 classes / outputs) and its input a few hundred
// wide: 32 rows x K x N per workgroup is too little for the MFMA path above to amortise its LDS
// combines and barriers (k_head: 7.7 us in-kernel on 32 CUs for the 784->200->10 step).  Here every
// WAVE owns ONE batch row and the grid covers the whole chip (batch/4 workgroups of 4 waves):
// lane l holds hidden units l, l+64, ... (UT of them) and their W rows in registers, so
//   z      = sum over the lane's units, then a butterfly sum over the 64 lanes (fixed order),
//   loss row, delta_L           every lane redundantly (N <= 32 values),
//   delta_{L-1}[u] = act'(h[u]) * sum_c delta_L[c] W[u][c]   for the lane's own units,
// with no LDS, no barrier and coalesced row loads / stores.  NP = padded class count.
// value of lane `lane` (compile-time or wave-uniform) as a scalar
__device__ __forceinline__ float pyz_readlane(const float v, const int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// every lane of a 16-lane row gets the row's sum: lane ^ 1, lane ^ 2 (quad permutes), then the
// mirror inside each half row and the mirror of the row (data-parallel primitives: VALU speed)
template <int CTRL>
__device__ __forceinline__ float pyz_dpp(const float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float pyz_row16_allsum(float v) {
  v += pyz_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
  v += pyz_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
  v += pyz_dpp<0x141>(v);  // row_half_mirror
  v += pyz_dpp<0x140>(v);  // row_mirror
  return v;
}

__device__ __forceinline__ float pyz_row16_allmax(float v) {
  v = fmaxf(v, pyz_dpp<0xB1>(v));
  v = fmaxf(v, pyz_dpp<0x4E>(v));
  v = fmaxf(v, pyz_dpp<0x141>(v));
  v = fmaxf(v, pyz_dpp<0x140>(v));
  return v;
}

// The lane-resident part of the last layer's [W; b]: lane l keeps the rows of units l, l + 64, ... and (lane c) b[c].
// A wave loads it once and uses it for every batch row it takes.
template <int UT, int NP>
struct PyzHeadW {
  float wv[UT][NP];
  float bias_mine;
};

template <int UT, int NP>
__device__ __forceinline__ void pyz_head_load_w(const HeadArgs &g, const int p, const int l, PyzHeadW<UT, NP> &W) {
  const int K = g.K, N = g.N;
  const float *wl = g.theta + p * g.theta_pstride + g.w_off;
  // Operands through buffer descriptors (per-lane byte offsets, range checked against the byte count).
  // Only the per-lane offset is range checked by the hardware (not the scalar one): units past K get an
  // out-of-range voffset; a padded class c >= N reads a neighbouring in-range element (or 0 past the
  // end), which is harmless: z[c] is never used and delta_L[c] = 0 multiplies it
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wl), 0, (K + 1) * N * 4, 0x00020000);
  constexpr unsigned OOB = 0x7FFFFF00u;
#pragma unroll
  for (int t = 0; t < UT; ++t) {
    const int u = l + 64 * t;
    const unsigned wo = u < K ? 4u * (unsigned)u * (unsigned)N : OOB;
#pragma unroll
    for (int c = 0; c < NP; ++c) W.wv[t][c] = pyz_buf_load(rw, wo + 4u * (unsigned)c, 0u);
  }
  W.bias_mine = pyz_buf_load(rw, l < N ? 4u * ((unsigned)K * (unsigned)N + (unsigned)l) : OOB, 0u);  // lane c: b[c]
}

// The head of ONE batch row m by ONE wave (lane l).  PRELOADED: W already holds the wave's share of [W; b] (several
// rows per wave); else it is fetched here, behind the row's own loads (one row per wave: the order of round 1 --
// with the 48 strided [W; b] loads in front of them the row's input and label chain start 0.4 us later).
// HP: the hidden layer arrives as the partial sums of a split-reduction forward (g.hparts; only instantiated for the shapes
// that path takes -- as a run-time branch it cost every instantiation 35 registers: k_head_rows<4, 12, 4> fell from four to
// three waves per SIMD and from 60 to 76 us at C5)
template <int UT, int NP, bool PRELOADED, bool HP = false>
__device__ __forceinline__ void pyz_head_row(const HeadArgs &g, const int batch, const long long row_off, const int p,
                                             const int m, const int l, PyzHeadW<UT, NP> &W) {
  const int K = g.K, N = g.N;
  long long yrow = m;
  if (g.row_idx) yrow = g.row_idx[row_off + m];
  const float *hp = g.hin + p * g.hin_pstride + (g.gather_hin ? yrow : (long long)m) * g.lda;
  // units past K read as zero.  The class loops below are straight-line code over the NP padded
  // classes: no per-class branches (conditional writes into the register arrays would turn them
  // into whole-vector copies).
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(hp), 0, K * 4, 0x00020000);
  float hv[UT], z[NP];
  if constexpr (HP) {
    // the hidden layer as raw partial sums of a split reduction: eight loads per unit in flight together (splits past
    // n_hparts are out of range: zero), summed in split order; then bias and activation; the row goes to h_store
    const float *pp = g.hparts + (long long)m * K;
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(pp), 0,
        (int)(((long long)(g.n_hparts - 1) * g.hpart_stride + K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.bias_prev), 0, K * 4, 0x00020000);
    constexpr unsigned OOBH = 0x7FFFFF00u;
    float pv[UT][8], bv[UT];
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      const int u = l + 64 * t;
#pragma unroll
      for (int sp = 0; sp < 8; ++sp)
        pv[t][sp] = pyz_buf_load(rp, (u < K && sp < g.n_hparts) ? 4u * (unsigned)(sp * g.hpart_stride + u) : OOBH, 0u);
      bv[t] = pyz_buf_load(rb, 4u * (unsigned)u, 0u);
    }
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      float a = pv[t][0];
#pragma unroll
      for (int sp = 1; sp < 8; ++sp) a += pv[t][sp];
      hv[t] = pyz_act(a + bv[t], g.act_prev);
      const int u = l + 64 * t;
      if (u < K) pyz_st(g.h_store + (long long)m * K + u, hv[t], g.wt);
    }
  } else {
#pragma unroll
    for (int t = 0; t < UT; ++t) hv[t] = pyz_buf_load(rh, 4u * (unsigned)(l + 64 * t), 0u);
  }
  if (!PRELOADED) pyz_head_load_w<UT, NP>(g, p, l, W);
  const auto &wv = W.wv;
  const float bias_mine = W.bias_mine;
  int ylab = 0;
  if (g.loss == PYZ_LOSS_SCCE) ylab = reinterpret_cast<const int32_t *>(g.y)[yrow];
  PYZ_STAMP(1, 1);
  // ---- z[c] = sum_u h[u] W[u][c] + b[c]: the lane's units, then a butterfly over the lanes
#pragma unroll
  for (int c = 0; c < NP; ++c) {
    float a = 0.0f;
#pragma unroll
    for (int t = 0; t < UT; ++t) a = fmaf(hv[t], wv[t][c], a);
    z[c] = a;
  }
  // sum over the 64 lanes without LDS traffic: four DPP steps leave each 16-lane row's sum in all its
  // lanes, the four row sums are read as scalars and added in a fixed order (bitwise reproducible)
#pragma unroll
  for (int c = 0; c < NP; ++c) {
    const float v = pyz_row16_allsum(z[c]);
    z[c] = ((pyz_readlane(v, 0) + pyz_readlane(v, 16)) + (pyz_readlane(v, 32) + pyz_readlane(v, 48))) + pyz_readlane(bias_mine, c);
  }
  PYZ_STAMP(1, 2);
  // ---- loss row and delta_L (every lane holds all N values; padded classes carry zeros)
  float d2[NP], lm, my_d = 0.0f, my_o = 0.0f;
  if (g.loss == PYZ_LOSS_SCCE && NP <= 16) {
    // softmax with the classes spread over the lanes of each 16-lane row (lane c owns class c; the four rows
    // hold copies): one exp per lane instead of N in sequence on every lane, max and sum by DPP row steps
    const int lc = l & 15;
    float zc = z[0];
#pragma unroll
    for (int c = 1; c < NP; ++c) zc = (lc == c) ? z[c] : zc;
    const bool on = lc < N;
    const float mx = pyz_row16_allmax(on ? zc : -3.0e38f);
    const float exv = on ? expf(zc - mx) : 0.0f;
    const float se = pyz_row16_allsum(exv);
    const float lse = mx + logf(se);
    const bool mine = on && lc == ylab;
    float zy = pyz_row16_allsum(mine ? zc : 0.0f);
    zy = (ylab >= 0 && ylab < N) ? zy : __builtin_nanf("");
    const float inv = 1.0f / (float)batch, rse = 1.0f / se;
    my_d = (exv * rse - (mine ? 1.0f : 0.0f)) * inv;  // softmax - onehot, over the batch mean (0 past N)
    my_o = zc;
#pragma unroll
    for (int c = 0; c < NP; ++c) d2[c] = pyz_readlane(my_d, c < 16 ? c : 0);
    lm = lse - zy;
  } else {
    float outv[NP];
    if (g.loss == PYZ_LOSS_SCCE) {
      float mx = -3.0e38f;
#pragma unroll
      for (int c = 0; c < NP; ++c) mx = fmaxf(mx, c < N ? z[c] : -3.0e38f);
      float ex[NP], se = 0.0f;
#pragma unroll
      for (int c = 0; c < NP; ++c) {
        ex[c] = c < N ? expf(z[c] - mx) : 0.0f;
        se += ex[c];
      }
      const float lse = mx + logf(se);
      float zy = __builtin_nanf("");
      const float inv = 1.0f / (float)batch, rse = 1.0f / se;
#pragma unroll
      for (int c = 0; c < NP; ++c) {
        zy = (c == ylab && c < N) ? z[c] : zy;
        d2[c] = (ex[c] * rse - ((c == ylab && c < N) ? 1.0f : 0.0f)) * inv;  // softmax - onehot, over the batch mean
        outv[c] = z[c];
      }
      lm = lse - zy;
    } else {
      const float *y = reinterpret_cast<const float *>(g.y) + yrow * N;
      const float sc = 2.0f / ((float)batch * (float)N);
      const int act = g.act_last;
      float a = 0.0f;
#pragma unroll
      for (int c = 0; c < NP; ++c) {
        const float o = pyz_act(z[c], act);
        const float e = o - y[min(c, N - 1)];
        a += c < N ? e * e : 0.0f;
        d2[c] = c < N ? sc * e * pyz_act_grad(o, act) : 0.0f;
        outv[c] = o;
      }
      lm = a / (float)N;
    }
    // lane c stores column c of the row
#pragma unroll
    for (int c = 0; c < NP; ++c) {
      my_d = (l == c) ? d2[c] : my_d;
      my_o = (l == c) ? outv[c] : my_o;
    }
  }
  PYZ_STAMP(1, 3);
  if (l < N) {
    if (g.out_last) pyz_st(g.out_last + p * g.last_pstride + (long long)m * N + l, my_o, g.wt);
    if (g.delta_last) pyz_st(g.delta_last + p * g.last_pstride + (long long)m * N + l, my_d, g.wt);
  }
  if (l == 0) g.part[p * g.nblk + m] = (double)lm;
  PYZ_STAMP(1, 4);
  if (!g.delta_prev) return;
  float *op = g.delta_prev + p * g.prev_pstride + (long long)m * K;
  const int actp = g.act_prev;
#pragma unroll
  for (int t = 0; t < UT; ++t) {
    const int u = l + 64 * t;
    float a = 0.0f;
#pragma unroll
    for (int c = 0; c < NP; ++c) a = fmaf(d2[c], wv[t][c], a);
    if (u < K) pyz_st(op + u, a * pyz_act_grad(hv[t], actp), g.wt);
  }
  PYZ_STAMP(1, 5);
}

// RW = batch rows per wave: 1 when the launch has few rows (a single chain: every row its own wave, the chip is
// barely filled as it is), 4 when there are tens of thousands (64 particles x 1024 rows): the wave's share of
// [W; b] -- UT x NP strided loads, most of the kernel's memory instructions -- is then fetched once for four rows.
template <int UT, int NP, int RW = 1, bool HP = false>
__global__ void __launch_bounds__(256) k_head_rows(HeadArgs g) {
  PYZ_STAMP(1, 0);
  const int w = pyz_wave_id(), l = threadIdx.x & 63;
  if (g.gate && g.gate->n % g.gate_mod == 0) return;
  const StepCtl ctl = pyz_ctl_first(g.ctl, g.init, g.init_on);
  const int batch = ctl.batch, p = blockIdx.y;
  const int m0 = (blockIdx.x * 4 + w) * RW;  // this wave's first batch row (scalar)
  if (m0 >= batch) {
    if (l < RW && m0 + l < g.nblk) g.part[p * g.nblk + m0 + l] = 0.0;
    return;
  }
  PyzHeadW<UT, NP> W;
  if (RW == 1) {
    pyz_head_row<UT, NP, false, HP>(g, batch, ctl.row_off, p, m0, l, W);
    return;
  }
  pyz_head_load_w<UT, NP>(g, p, l, W);
#pragma unroll 1
  for (int rr = 0; rr < RW; ++rr) {
    const int m = m0 + rr;
    if (m >= batch) {   // uniform
      if (l == 0 && m < g.nblk) g.part[p * g.nblk + m] = 0.0;
      continue;
    }
    pyz_head_row<UT, NP, true>(g, batch, ctl.row_off, p, m, l, W);
  }
}

// ---------------------------------------------------------------- the next step's batch, assembled ahead of its step
// In a device-resident run the rows of step s + 1 are known while step s runs (row table + per-run tables).  The
// workgroups of k_wgrad_all that own no tile -- 73 of the 256 CUs idle at C2 -- copy those rows of the resident data
// set into the other one of two contiguous (max_batch, K) buffers.  The forward pass of step s + 1 then reads plain
// contiguous rows: no index load in front of its first operand load, no first touch of HBM rows on its critical
// path, no copy to store for the weight-gradient kernel.  A worker takes rows that the forward tiles of workgroups
// with its own blockIdx % 8 read (workgroups are dealt round-robin over the XCDs: the copy then waits in the L2 of
// the XCD that reads it; placement only changes speed).
struct PrepArgs {
  const float *src;           // (n_rows, lda) resident data set; nullptr: no preparation in this launch
  float *dst;                 // (max_batch, K) contiguous batch of the NEXT step
  const int32_t *row_idx;     // the run's row table
  const int32_t *tab_bs;      // per-run batch sizes
  long long row_stride;
  int K, lda;
  int fwd_tiles, fwd_tiles_n; // tile grid of the forward kernel that will read dst (pyz_xcd_remap over fwd_tiles ids)
};

// rows [ra, rb) of the batch whose row indices start at idx: dst[m][:] = src[idx[m]][:]
__device__ __forceinline__ void pyz_copy_rows(const PrepArgs &g, const int32_t *idx, const int ra, const int rb) {
  const int K = g.K, nt = blockDim.x, t = threadIdx.x;
  if (ra >= rb) return;
  if ((K & 3) == 0 && (g.lda & 3) == 0 && ((reinterpret_cast<uintptr_t>(g.src) | reinterpret_cast<uintptr_t>(g.dst)) & 15) == 0) {
    const int K4 = K >> 2, total = (rb - ra) * K4;
    for (int e0 = t; e0 < total; e0 += 4 * nt) {   // four independent 16-byte loads per thread and trip
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = min(e0 + u * nt, total - 1), m = ra + e / K4, c = e % K4;
        v[u] = *reinterpret_cast<const float4 *>(g.src + (long long)idx[m] * g.lda + 4 * c);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * nt;
        if (e < total) {
          float *d = g.dst + (long long)(ra + e / K4) * K + 4 * (e % K4);
          // write-through (sc1) stores: the copy leaves L2 while the tile workgroups still run.  With plain stores the
          // 3.2 MB stay dirty in L2 until the kernel ends and its end waits for their write-back (measured at C2:
          // k_wgrad_all 10.75 against 10.3 us; the stand-alone first-batch launch 28 against 4 us)
          const f32x4 vv = {v[u].x, v[u].y, v[u].z, v[u].w};
          asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(d), "v"(vv) : "memory");
        }
      }
    }
    return;
  }
  const int total = (rb - ra) * K;
  for (int e = t; e < total; e += nt) {
    const int m = ra + e / K, c = e % K;
    g.dst[(long long)m * K + c] = g.src[(long long)idx[m] * g.lda + c];
  }
}

// worker j of cnt (its workgroup id is bid): its share of the nb rows
__device__ __forceinline__ void pyz_prep_rows(const PrepArgs &g, const int32_t *idx, const int nb, const int j, const int cnt,
                                              const int bid) {
  int ra, rb;
  if (cnt >= 8 && g.fwd_tiles >= 8) {
    // the forward tiles of the workgroups with id % 8 == x: ids [t0, t1) after pyz_xcd_remap, i.e. row blocks from t0 / tiles_n
    const int x = bid & 7, q = g.fwd_tiles >> 3, r = g.fwd_tiles & 7;
    const int t0 = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q, t1 = t0 + (x < r ? q + 1 : q);
    // rows whose first tile lies in [t0, t1)
    const int b0 = (t0 + g.fwd_tiles_n - 1) / g.fwd_tiles_n, b1 = (t1 + g.fwd_tiles_n - 1) / g.fwd_tiles_n;
    const int r0 = min(32 * b0, nb), r1 = x == 7 ? nb : min(32 * b1, nb);
    // the workers with this residue: ids first, first + 8, ... (first = the smallest id >= bid - 8 j with id % 8 == x)
    const int id0 = bid - j;                       // id of worker 0
    const int first = id0 + ((x - (id0 & 7)) & 7); // first worker id with residue x
    const int mine = (bid - first) >> 3, all = (id0 + cnt - 1 - first) / 8 + 1;
    const int per = (r1 - r0 + all - 1) / all;
    ra = r0 + mine * per;
    rb = min(ra + per, r1);
  } else {
    const int per = (nb + cnt - 1) / cnt;
    ra = j * per;
    rb = min(ra + per, nb);
  }
  pyz_copy_rows(g, idx, ra, rb);
}

// the first batch of a run (nobody ran before it): the same copy as a launch of its own
__global__ void k_prep_batch(PrepArgs g, const StepCtl *ctl) {
  pyz_prep_rows(g, g.row_idx + ctl->row_off, ctl->batch, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.x);
}

// short runs (the tables travel as kernel arguments): k_set_ctl_tabs and k_prep_batch as ONE launch in front of the
// first step -- block 0 writes the tables and the first step's scalars, every block copies its share of the first batch
// (row offset and batch size of that step come as arguments: nothing here reads what block 0 writes)
__global__ void k_run_start(StepCtl *ctl, InlineTabs t, int32_t *tab_bs, float *tab_lr, long long n, long long row_off,
                            int slot0, PrepArgs g) {
  if (blockIdx.x == 0) {
    const int e = threadIdx.x;
    if (e < t.n) {
      tab_bs[e] = t.bs[e];
      tab_lr[e] = t.lr[e];
    }
    if (e == 0) {
      ctl->batch = t.bs[0];
      ctl->lr = t.lr[0];
      ctl->n = n;
      ctl->row_off = row_off;
      ctl->i = 0;
      ctl->slot0 = slot0;
      ctl->n_run = t.n - 1;
    }
  }
  pyz_prep_rows(g, g.row_idx + row_off, t.bs[0], (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.x);
}

// ---------------------------------------------------------------- all weight gradients + update
struct WgradLayer {
  const float *in;            // layer input: data x (layer 0) or act[l-1]
  long long in_pstride;
  const float *delta;         // (P, max_batch, N)
  long long delta_pstride;
  long long w_off;
  int lda, K, N;
  int tile0;                  // first tile of this layer in the launch
  int gather;                 // rows of `in` go through row_idx (layer 0)
};

#define PYZ_UPD_NONE 0  // write [dW; db] to grad
#define PYZ_UPD_SGD 1   // SGD.py:66-69
#define PYZ_UPD_SGLD 2  // SGLD.py:64-93
#define PYZ_UPD_BBB 3   // BBB.py:152-201 (theta = mu, mean = rho, sq_mean = the sampled w; read only)
#define PYZ_UPD_SWAG 4  // SWAG.py:61-92 (SGD update; gated moments; one deviation row)

struct WgradArgs {
  WgradLayer lay[PYZ_MAX_LAYERS];
  int L;
  int tiles;                  // tiles of all layers; the launch has one more workgroup, for the step duties
  const int32_t *row_idx;
  const StepCtl *ctl;
  int mode;
  float *grad;                // mode NONE: (P, D)
  long long grad_pstride;
  float *theta, *mean, *sq_mean;
  uint64_t seed;
  const float *unit_noise;    // SGLD: injected N(0,1); BBB: injected eps
  // SWAG only: dev_row = the (D) row of the deviation matrix that receives theta - mean, or nullptr
  float *dev_row;
  int swag_update;            // moments / deviation updated at this step (n % frequency == 0)
  // chained SWAG runs (pyz_swag_run): both follow from the step count n -- update when n % swag_freq == 0, into
  // row min(ceil(n / swag_freq), swag_k - 1) of the (swag_k, dev_stride) deviation matrix dev_base
  int swag_freq, swag_k;
  float *dev_base;
  long long dev_stride;
  // BBB only
  float alpha, prior_mean, prior_rho, bbb_lr;
  const float *pm_vec, *pr_vec;
  uint32_t bbb_step;
  int bbb_chained;            // device-resident run: learning rate and step count come from StepCtl, cost goes to slot (slot0 + i)
  const double *part_kl;
  int nblk_kl;
  float *cost;
  // chained BBB runs, sampling fused: the epilogue also draws the NEXT step's weights from the mu / rho it just wrote
  // (w_next, Philox step n + 1 -- k_bbb_sample's arithmetic) and leaves that step's log q - log p partial of its tile in
  // part_kl_next[workgroup]; nullptr = the step's own k_bbb_sample launch does both
  float *w_next;
  double *part_kl_next;
  // duties of workgroup 0 at the end of the step: loss and the next step's scalars
  const double *part;
  int nblk;
  float *loss;
  int loss_indexed;
  StepCtl *next;
  const int32_t *tab_bs;
  const float *tab_lr;
  long long row_stride;
  int *nonfinite;             // device counter of steps with a NaN / Inf loss (may be nullptr)
  float *ploss;               // particle-batched gradient passes: loss[p] = (sum of particle p's row partials) / batch
  int wt;                     // write-through stores for the updated state (pyz_st)
  PrepArgs prep;              // chained runs: the workgroups past the duties workgroup assemble the NEXT step's batch
};

// ---------------------------------------------------------------- the optimizer updates (shared by the
// epilogue of k_wgrad_all; kept apart from the tile code so that every mode is one readable block)
struct PyzUpdOut {
  float th, mu, sq, dev;
  bool wr_mu, wr_sq, wr_dev;
  float *dev_row;  // SWAG: the deviation row written at this step
};

// gv = d loss / d parameter e; th0 / mu0 / sq0 / zz = the state read before (see the modes above)
__device__ __forceinline__ PyzUpdOut pyz_update_math(const WgradArgs &g, const int mode, const long long e, const float gv,
                                                     const float th0, const float mu0, const float sq0, const float zz,
                                                     const float lr, const long long nstep) {
  PyzUpdOut o;
  o.th = o.mu = o.sq = o.dev = 0.0f;
  o.wr_mu = o.wr_sq = o.wr_dev = false;
  o.dev_row = nullptr;
  if (mode == PYZ_UPD_SGD) {
    o.th = th0 - lr * gv;
  } else if (mode == PYZ_UPD_SWAG) {
    const float th = th0 - lr * gv;
    o.th = th;
    const bool upd = g.swag_freq > 0 ? (nstep % g.swag_freq == 0) : (g.swag_update != 0);
    if (upd) {
      const float fn = (float)nstep, fn1 = fn + 1.0f;
      o.mu = (mu0 * fn + th) / fn1;
      o.sq = (sq0 * fn + th * th) / fn1;
      o.dev = th - o.mu;
      o.wr_mu = o.wr_sq = true;
      // columns are appended until there are k; afterwards the LAST one is replaced (SWAG.py:85-89, as written)
      o.dev_row = g.swag_freq > 0
                      ? g.dev_base + min((nstep + g.swag_freq - 1) / g.swag_freq, (long long)g.swag_k - 1) * g.dev_stride
                      : g.dev_row;
      o.wr_dev = o.dev_row != nullptr;
    }
  } else if (mode == PYZ_UPD_BBB) {
    // th0 = mu, mu0 = rho, sq0 = the sampled w, zz = eps, gv = d loss / d w   (k_bbb_update's arithmetic)
    const float pmean = g.pm_vec ? g.pm_vec[e] : g.prior_mean;
    const float sp = pyz_softplus(g.pr_vec ? g.pr_vec[e] : g.prior_rho), isp2 = 1.0f / (sp * sp);
    const float mu = th0, rho = mu0, wv = sq0;
    const float sg = pyz_softplus(rho), sig = pyz_sigmoid(rho);
    const float d = wv - mu, is2 = 1.0f / (sg * sg);
    const float g_mu = g.alpha * d * is2;
    const float g_rho = g.alpha * (-1.0f / sg + d * d * is2 / sg) * sig;
    const float g_w = gv + g.alpha * (-d * is2 + (wv - pmean) * isp2);
    const float blr = g.bbb_chained ? lr : g.bbb_lr;
    o.th = mu - blr * (g_mu + g_w);
    o.mu = rho - blr * (zz * sig * g_w + g_rho);
    o.wr_mu = true;
  } else if (mode == PYZ_UPD_SGLD) {
    const float fn = (float)nstep, fn1 = fn + 1.0f;
    const float noise = lr * zz;
    const float th = th0 + (-lr) * (gv + noise);
    o.th = th;
    o.mu = (mu0 * fn + th) / fn1;
    o.sq = (sq0 * fn + th * th) / fn1;
    o.wr_mu = o.wr_sq = true;
  }
  return o;
}

__device__ __forceinline__ void pyz_update_store(const WgradArgs &g, const int mode, const long long e, const int p,
                                                 const float gv, const PyzUpdOut &o) {
  if (mode == PYZ_UPD_NONE) {
    g.grad[p * g.grad_pstride + e] = gv;
    return;
  }
  pyz_st(g.theta + e, o.th, g.wt);
  if (o.wr_mu) pyz_st(g.mean + e, o.mu, g.wt);
  if (o.wr_sq) pyz_st(g.sq_mean + e, o.sq, g.wt);
  if (o.wr_dev) o.dev_row[e] = o.dev;
}

// duties of one spare wave per step that do not depend on the gradient: the loss of this step
// (partials written by k_head), BBB's cost, and the next step's scalars
__device__ __forceinline__ void pyz_step_duties(const WgradArgs &g, const int l) {
  const int batch = g.ctl->batch;
  if (g.loss) {
    const double v = pyz_sum_partials(g.part, g.nblk);
    if (l == 0) {
      float *lo = g.loss + (g.loss_indexed ? g.ctl->slot0 + g.ctl->i : 0);
      lo[0] = (float)(v / (double)batch);
      pyz_note_loss(g.nonfinite, lo[0]);
      if (g.next) pyz_prepare_next(g.ctl, g.next, g.tab_bs, g.tab_lr, g.row_stride);
    }
  }
  if (g.mode == PYZ_UPD_BBB) {   // cost = loss + alpha (log q - log p)
    const double v = pyz_sum_partials(g.part, g.nblk), k = pyz_sum_partials(g.part_kl, g.nblk_kl);
    if (l == 0) {
      const float loss = (float)(v / (double)batch), kl = (float)k;
      float *co = g.cost + (g.bbb_chained ? 4 * (g.ctl->slot0 + g.ctl->i) : 0);
      co[0] = loss + g.alpha * kl;
      pyz_note_loss(g.nonfinite, co[0]);
      co[1] = loss;
      co[2] = kl;
      if (g.bbb_chained && g.next) pyz_prepare_next(g.ctl, g.next, g.tab_bs, g.tab_lr, g.row_stride);
    }
  }
}

// contiguous (no gather) reduction over the batch rows, software pipelined.  Operands come through
// buffer descriptors built from wave-uniform bases: a step's address is a scalar row offset
// (soffset) plus a per-lane constant (voffset), so the loop spends no vector ALU on addressing.
// Requires batch * row bytes < 2^31 (checked when the plan is created).

template <class H>
__device__ __forceinline__ void pyz_wgrad_accumulate(f32x16 &acc, const float *abase, const int ic, const float *dbase,
                                                     const int n, const int lda, const int N, const int batch, int s,
                                                     const int se, const int h, const bool is_w, const bool is_b, H hook) {
  const float bconst = is_b ? 1.0f : 0.0f;
  const unsigned a_lane = ((unsigned)h * (unsigned)lda + (unsigned)ic) * 4u;
  const unsigned d_lane = ((unsigned)h * (unsigned)N + (unsigned)n) * 4u;
  const unsigned a_row2 = 8u * (unsigned)lda, d_row2 = 8u * (unsigned)N;  // bytes per MFMA step (two rows)
  const __amdgpu_buffer_rsrc_t ra =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(abase), 0, (int)((unsigned)batch * (unsigned)lda * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(dbase), 0, (int)((unsigned)batch * (unsigned)N * 4u), 0x00020000);
  // only whole steps here: the scalar offset is not range checked, so the odd last row is peeled
  const int full = batch >> 1;
  pyz_steps1_all(
      s, min(se, full), acc,
      [&](int st, float &a, float &d) {
        a = pyz_buf_load(ra, a_lane, (unsigned)PYZ_HOT(st, batch) * a_row2);
        d = pyz_buf_load(rd, d_lane, (unsigned)PYZ_HOT(st, batch) * d_row2);
      },
      [&](int, float &a, float &) { a = is_w ? a : bconst; }, hook);
  if (se > full) {  // this wave owns the odd last row: the second reduction slot of its MFMA stays empty
    const int b = batch - 1;
    float a = abase[(size_t)b * lda + ic], d = dbase[(size_t)b * N + n];
    a = h == 0 ? (is_w ? a : bconst) : 0.0f;
    d = h == 0 ? d : 0.0f;
    acc = pyz_mfma(a, d, acc);
  }
}

// S = waves per workgroup (compile time: the epilogue prefetches 16/S elements per thread).
// PLAIN: the launch only writes gradients (g.mode == PYZ_UPD_NONE: particle-batched passes of SVGD / HMC, loss_grad):
// no optimizer state is fetched, so the four prefetch arrays of the epilogue do not exist -- for S = 1 that is 64
// registers per lane, i.e. the difference between two and four resident waves per SIMD.
template <int S, bool PLAIN = false>
__global__ void __launch_bounds__(64 * S) k_wgrad_all(WgradArgs g) {
  extern __shared__ float red[];
  PYZ_STAMP(2, 0);
  constexpr int EPT = 16 / S;  // tile elements per thread in the epilogue
  const int w = pyz_wave_id(), l = threadIdx.x & 63;
  const int r = l & 31, h = l >> 5;
  // workgroup `tiles` (the last id) owns no tile: its first wave does the duties of the step that do
  // not depend on the gradient, off the critical path of the tile workgroups
  if (blockIdx.x >= (unsigned)g.tiles) {  // (without a batch to prepare, ids past `tiles` + 1 only pad the launch)
    if (blockIdx.x == (unsigned)g.tiles && blockIdx.y == 0 && w == 0) pyz_step_duties(g, l);
    if (blockIdx.x == (unsigned)g.tiles && w == 0 && g.ploss) {   // every particle's spare workgroup: its loss (k_loss_finalize)
      const double tot = pyz_sum_partials(g.part + (long long)blockIdx.y * g.nblk, g.nblk);
      if (l == 0) {
        g.ploss[blockIdx.y] = (float)(tot / (double)g.ctl->batch);
        pyz_note_loss(g.nonfinite, g.ploss[blockIdx.y]);
      }
    }
    if (!PLAIN && blockIdx.x > (unsigned)g.tiles && blockIdx.y == 0 && g.prep.src) {
      const int i2 = g.ctl->i + 1;
      if (i2 < g.ctl->n_run)
        pyz_prep_rows(g.prep, g.prep.row_idx + g.ctl->row_off + g.prep.row_stride, g.prep.tab_bs[i2],
                      (int)blockIdx.x - g.tiles - 1, (int)gridDim.x - g.tiles - 1, (int)blockIdx.x);
      PYZ_STAMP(2, 3);
    }
    return;
  }
  const int tile = pyz_xcd_remap(blockIdx.x, g.tiles);
  int li = 0;
  while (li + 1 < g.L && tile >= g.lay[li + 1].tile0) ++li;
  const WgradLayer &ly = g.lay[li];
  const int batch = g.ctl->batch;
  const int K = ly.K, N = ly.N;
  const int tiles_n = (N + 31) >> 5;
  const int t = tile - ly.tile0;
  const int i0 = (t / tiles_n) * 32, n0 = (t % tiles_n) * 32;
  const int p = blockIdx.y;
  const long long w_off = ly.w_off;
  const int mode = PLAIN ? PYZ_UPD_NONE : g.mode;
  const long long nstep = g.ctl->n;
  const float lr = g.ctl->lr;

  // -- epilogue operands of this thread's elements and the Philox noise: needed only after the
  //    reduction, so they are fetched / generated right behind the first group of operand loads
  //    (the hook of the pipelined loop) and their latency hides behind the loads and MFMAs
  long long ee[EPT];
  bool ev[EPT];
  float th0[PLAIN ? 1 : EPT], mu0[PLAIN ? 1 : EPT], sq0[PLAIN ? 1 : EPT], zz[PLAIN ? 1 : EPT];
  float zg[PLAIN ? 1 : EPT];   // generated noise: its own registers (writing a register a load may still be filling waits for EVERY load in flight)
  float zn[PLAIN ? 1 : EPT];   // BBB with fused sampling: the next step's eps
  const bool fuse_next = !PLAIN && mode == PYZ_UPD_BBB && g.w_next != nullptr;
  auto prefetch = [&]() {
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      int ro, co;
      if (S == 1) {
        ro = (q & 3) + 8 * (q >> 2) + 4 * h;
        co = r;
      } else {
        const int e = threadIdx.x + q * 64 * S;
        ro = e >> 5;
        co = e & 31;
      }
      const int ii = i0 + ro, nn = n0 + co;
      ev[q] = ii <= K && nn < N;
      ee[q] = w_off + (long long)min(ii, K) * N + min(nn, N - 1);
      if (PLAIN) continue;
      th0[q] = mu0[q] = sq0[q] = zz[q] = zg[q] = 0.0f;
      if (mode != PYZ_UPD_NONE) th0[q] = g.theta[ee[q]];
      if (mode == PYZ_UPD_SGLD || mode == PYZ_UPD_BBB || mode == PYZ_UPD_SWAG) {
        mu0[q] = g.mean[ee[q]];
        sq0[q] = g.sq_mean[ee[q]];
        if (g.unit_noise && mode != PYZ_UPD_SWAG) zz[q] = g.unit_noise[ee[q]];
      }
    }
    if (!PLAIN && (mode == PYZ_UPD_SGLD || mode == PYZ_UPD_BBB) && !g.unit_noise) {
#pragma unroll
      for (int q = 0; q < EPT; ++q) {
        zg[q] = mode == PYZ_UPD_SGLD ? pyz_normal1(g.seed, PYZ_STREAM_SGLD, (uint32_t)nstep, (uint64_t)ee[q])
                                     : pyz_normal1(g.seed, PYZ_STREAM_BBB, g.bbb_chained ? (uint32_t)nstep : g.bbb_step, (uint64_t)ee[q]);
      }
    }
    if (!PLAIN && fuse_next) {
#pragma unroll
      for (int q = 0; q < EPT; ++q) zn[q] = pyz_normal1(g.seed, PYZ_STREAM_BBB, (uint32_t)(nstep + 1), (uint64_t)ee[q]);
    }
  };

  const int i = i0 + r, n = min(n0 + r, N - 1);
  const int ic = min(i, K - 1);
  const bool is_w = i < K, is_b = i == K;
  const float *ap = ly.in + p * ly.in_pstride + ic;
  const float *dp = ly.delta + p * ly.delta_pstride + n;
  const int32_t *idx = (ly.gather && g.row_idx) ? g.row_idx + g.ctl->row_off : nullptr;
  f32x16 acc = {0};
  const int steps = (batch + 1) >> 1;
  int s = (steps * w) / S;
  const int se = (steps * (w + 1)) / S;
  PYZ_STAMP(2, 1);
  if (idx) {
    prefetch();
    // gathered rows: the 64 row indices of a 32-step chunk come from ONE coalesced load (lane j
    // holds the index of batch row 2*s0 + j) and reach the step that needs them by a lane
    // permute, so the pipelined operand loads never wait on an index load
    for (int s0 = s; s0 < se; s0 += 32) {
      const int idxv = idx[min(2 * s0 + l, batch - 1)];
      const int lda = ly.lda;
      pyz_steps1_all(
          s0, min(s0 + 32, se), acc,
          [&](int st, float &a, float &d) {
            const int bc = PYZ_HOT(min(2 * st + h, batch - 1), batch);
            const long long row = PYZ_HOT(__shfl(idxv, 2 * (st - s0) + h, 64), batch);
            a = ap[row * lda];
            d = dp[(long long)bc * N];
          },
          [&](int st, float &a, float &d) {
            const bool vb = 2 * st + h < batch;
            a = vb ? (is_w ? a : (is_b ? 1.0f : 0.0f)) : 0.0f;
            d = vb ? d : 0.0f;
          });
    }
  } else {
    pyz_wgrad_accumulate(acc, ly.in + p * ly.in_pstride, ic, ly.delta + p * ly.delta_pstride, n, ly.lda, N, batch, s, se, h,
                         is_w, is_b, prefetch);
  }
  PYZ_STAMP(2, 2);
  float gv[EPT];
  if (S == 1) {
#pragma unroll
    for (int q = 0; q < EPT; ++q) gv[q] = acc[q];
  } else {
    float *my = red + w * 1024;
#pragma unroll
    for (int q = 0; q < 16; ++q) my[((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + r] = acc[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int e = threadIdx.x + q * 64 * S;
      float v = red[e];
#pragma unroll
      for (int ww = 1; ww < S; ++ww) v += red[ww * 1024 + e];
      gv[q] = v;
    }
  }
  PYZ_STAMP(2, 4);
  double klp = 0.0;
  float sp_s = 1.0f, lsp_s = 0.0f;
  if (!PLAIN && fuse_next) {
    sp_s = pyz_softplus(g.prior_rho);
    lsp_s = logf(sp_s);
  }
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    if (!ev[q]) continue;
    if (PLAIN) {
      pyz_st(g.grad + p * g.grad_pstride + ee[q], gv[q], g.wt);
      continue;
    }
    const PyzUpdOut o = pyz_update_math(g, mode, ee[q], gv[q], th0[q], mu0[q], sq0[q],
                                        PLAIN ? 0.0f : (g.unit_noise ? zz[q] : zg[q]), lr, nstep);
    pyz_update_store(g, mode, ee[q], p, gv[q], o);
    if (fuse_next) {   // k_bbb_sample's element, from the mu (o.th) and rho (o.mu) of this update
      const long long e = ee[q];
      const float pmean = g.pm_vec ? g.pm_vec[e] : g.prior_mean;
      float sp = sp_s, lsp = lsp_s;
      if (g.pr_vec) {
        sp = pyz_softplus(g.pr_vec[e]);
        lsp = logf(sp);
      }
      const float mu = o.th, sg = pyz_softplus(o.mu);
      const float wn = zn[q] * sg + mu;
      pyz_st(g.w_next + e, wn, g.wt);
      const float a = (wn - mu) / sg, b = (wn - pmean) / sp;
      const float lq = -0.5f * a * a - logf(sg) - PYZ_LOG_SQRT_2PI;
      const float lp = -0.5f * b * b - lsp - PYZ_LOG_SQRT_2PI;
      klp += (double)lq - (double)lp;
    }
  }
  if (!PLAIN && fuse_next) {   // (uniform over the launch) this tile's share of the next step's log q - log p
    double tot = pyz_wave_sum(klp);
    if (S > 1) {
      double *sm = reinterpret_cast<double *>(red);
      __syncthreads();   // every wave is done reading the tile partials in `red`
      if (l == 0) sm[w] = tot;
      __syncthreads();
      tot = 0.0;
      if (threadIdx.x == 0)
        for (int ww = 0; ww < S; ++ww) tot += sm[ww];
    }
    if (threadIdx.x == 0) g.part_kl_next[blockIdx.x] = tot;
  }
  PYZ_STAMP(2, 3);
}
