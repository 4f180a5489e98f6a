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
};

__global__ void __launch_bounds__(256) k_head(HeadArgs g) {
  __shared__ float red[4 * 1024];
  __shared__ float zt[32 * 33];
  __shared__ float dt[32 * 33];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, r = l & 31, h = l >> 5;
  const int batch = g.ctl->batch, p = blockIdx.y, m0 = blockIdx.x * 32;
  if (m0 >= batch) {
    if (threadIdx.x == 0) g.part[p * g.nblk + blockIdx.x] = 0.0;
    return;
  }
  const int K = g.K, N = g.N;
  const int32_t *idx = g.row_idx ? g.row_idx + g.ctl->row_off : nullptr;
  {
    const int m = min(m0 + r, batch - 1), n = min(r, N - 1);
    long long row = m;
    if (g.gather_hin && idx) row = idx[m];
    const float *ap = g.hin + p * g.hin_pstride + row * g.lda;
    const float *wp = g.theta + p * g.theta_pstride + g.w_off + n;
    f32x16 acc = {0};
    pyz_fwd_accumulate(acc, ap, wp, K, N, g.vec, w, 4, h);
    float *my = red + w * 1024;
#pragma unroll
    for (int i = 0; i < 16; ++i) my[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 1024; e += 256)
    zt[(e >> 5) * 33 + (e & 31)] = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
  __syncthreads();
  if (threadIdx.x < 64) {  // wave 0; lanes 0..31 own one row each
    const int mm = m0 + r;
    const bool valid = (threadIdx.x < 32) && mm < batch;
    double lm = 0.0;
    if (threadIdx.x < 32) {
      const float *z = zt + r * 33;
      float *d = dt + r * 33;
      if (valid) {
        const long long row = idx ? (long long)idx[mm] : (long long)mm;
        float *ol = g.out_last ? g.out_last + p * g.last_pstride + (long long)mm * N : nullptr;
        float *dl = g.delta_last ? g.delta_last + p * g.last_pstride + (long long)mm * N : nullptr;
        if (g.loss == PYZ_LOSS_SCCE) {
          const int y = reinterpret_cast<const int32_t *>(g.y)[row];
          float mx = z[0];
          for (int c = 1; c < N; ++c) mx = fmaxf(mx, z[c]);
          float se = 0.0f;
          for (int c = 0; c < N; ++c) se += expf(z[c] - mx);
          const float lse = mx + logf(se);
          const float zy = (y >= 0 && y < N) ? z[y] : __builtin_nanf("");
          lm = (double)(lse - zy);
          const float inv = 1.0f / (float)batch;
          for (int c = 0; c < N; ++c) {
            const float dv = (expf(z[c] - lse) - (c == y ? 1.0f : 0.0f)) * inv;
            d[c] = dv;
            if (dl) dl[c] = dv;
            if (ol) ol[c] = z[c];
          }
        } else {
          const float *y = reinterpret_cast<const float *>(g.y) + row * N;
          const float sc = 2.0f / ((float)batch * (float)N);
          float a = 0.0f;
          for (int c = 0; c < N; ++c) {
            const float o = pyz_act(z[c], g.act_last);
            const float e = o - y[c];
            a += e * e;
            const float dv = sc * e * pyz_act_grad(o, g.act_last);
            d[c] = dv;
            if (dl) dl[c] = dv;
            if (ol) ol[c] = o;
          }
          lm = (double)(a / (float)N);
        }
      } else {
        for (int c = 0; c < N; ++c) d[c] = 0.0f;
      }
    }
    lm = pyz_wave_sum(lm);
    if (threadIdx.x == 0) g.part[p * g.nblk + blockIdx.x] = lm;
  }
  if (!g.delta_prev) return;
  __syncthreads();
  const int tiles_j = (K + 31) >> 5;
  const int nsteps = (N + 1) >> 1;
  const float *wl = g.theta + p * g.theta_pstride + g.w_off;
  const float *hp = g.hin + p * g.hin_pstride;
  float *op = g.delta_prev + p * g.prev_pstride;
  for (int jt = w; jt < tiles_j; jt += 4) {
    const int j0 = jt * 32, j = min(j0 + r, K - 1);
    const float *wp = wl + (long long)j * N;
    f32x16 acc = {0};
    pyz_steps1_all(0, nsteps, acc, [&](int s, float &a, float &b) {
      const int kk = 2 * s + h;
      const bool vk = kk < N;
      const int kc = vk ? kk : 0;
      const float av = dt[r * 33 + kc], bv = wp[kc];
      a = vk ? av : 0.0f;
      b = vk ? bv : 0.0f;
    });
    const int jj = j0 + r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int mm = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (mm < batch && jj < K) {
        const long long o = (long long)mm * K + jj;
        op[o] = acc[i] * pyz_act_grad(hp[o], g.act_prev);
      }
    }
  }
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

struct WgradArgs {
  WgradLayer lay[PYZ_MAX_LAYERS];
  int L;
  const int32_t *row_idx;
  const StepCtl *ctl;
  int mode;
  float *grad;                // mode NONE: (P, D)
  long long grad_pstride;
  float *theta, *mean, *sq_mean;
  uint64_t seed;
  const float *unit_noise;
  // duties of workgroup 0 at the end of the step: loss and the next step's scalars
  const double *part;
  int nblk;
  float *loss;
  int loss_indexed;
  StepCtl *next;
  const int32_t *tab_bs;
  const float *tab_lr;
  long long row_stride;
};

__global__ void k_wgrad_all(WgradArgs g) {
  extern __shared__ float red[];
  const int S = blockDim.x >> 6, w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int r = l & 31, h = l >> 5;
  int li = 0;
  while (li + 1 < g.L && (int)blockIdx.x >= g.lay[li + 1].tile0) ++li;
  const WgradLayer &ly = g.lay[li];
  const int batch = g.ctl->batch;
  const int K = ly.K, N = ly.N;
  const int tiles_n = (N + 31) >> 5;
  const int t = blockIdx.x - ly.tile0;
  const int i0 = (t / tiles_n) * 32, n0 = (t % tiles_n) * 32;
  const int p = blockIdx.y;
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
  if (idx) {
    pyz_wgrad_steps<16, true>(s, se, acc, ap, dp, idx, ly.lda, N, batch, h, r, is_w, is_b);
    pyz_wgrad_steps<4, true>(s, se, acc, ap, dp, idx, ly.lda, N, batch, h, r, is_w, is_b);
    pyz_wgrad_steps<1, true>(s, se, acc, ap, dp, idx, ly.lda, N, batch, h, r, is_w, is_b);
  } else {
    pyz_wgrad_steps<16, false>(s, se, acc, ap, dp, idx, ly.lda, N, batch, h, r, is_w, is_b);
    pyz_wgrad_steps<4, false>(s, se, acc, ap, dp, idx, ly.lda, N, batch, h, r, is_w, is_b);
    pyz_wgrad_steps<1, false>(s, se, acc, ap, dp, idx, ly.lda, N, batch, h, r, is_w, is_b);
  }
  const long long w_off = ly.w_off;
  const int mode = g.mode;
  const float lr = g.ctl->lr;
  const long long nstep = g.ctl->n;
  pyz_tile_epilogue(acc, red, [&](int ro, int co, float gv) {
    const int ii = i0 + ro, nn = n0 + co;
    if (ii > K || nn >= N) return;
    const long long e = w_off + (long long)ii * N + nn;
    if (mode == PYZ_UPD_NONE) {
      g.grad[p * g.grad_pstride + e] = gv;
    } else if (mode == PYZ_UPD_SGD) {
      g.theta[e] = g.theta[e] - lr * gv;
    } else {
      float z;
      if (g.unit_noise) {
        z = g.unit_noise[e];
      } else {
        const float4 q = pyz_normal4(g.seed, PYZ_STREAM_SGLD, (uint32_t)nstep, (uint64_t)(e >> 2));
        const int k = (int)(e & 3);
        z = k == 0 ? q.x : (k == 1 ? q.y : (k == 2 ? q.z : q.w));
      }
      const float fn = (float)nstep, fn1 = fn + 1.0f;
      const float noise = lr * z;
      const float th = g.theta[e] + (-lr) * (gv + noise);
      g.theta[e] = th;
      g.mean[e] = (g.mean[e] * fn + th) / fn1;
      g.sq_mean[e] = (g.sq_mean[e] * fn + th * th) / fn1;
    }
  });
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && g.loss) {
    float *lo = g.loss + (g.loss_indexed ? g.ctl->slot0 + g.ctl->i : 0);
    lo[0] = (float)(pyz_sum_partials(g.part, g.nblk) / (double)batch);
    if (g.next) pyz_prepare_next(g.ctl, g.next, g.tab_bs, g.tab_lr, g.row_stride);
  }
}
