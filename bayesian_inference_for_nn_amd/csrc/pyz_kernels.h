// pyz_kernels.h -- loss, update and reduction kernels of the optimizer steps.
// Each kernel cites the reference lines whose arithmetic it carries out.
#pragma once

#include "pyz_common.h"
#include "pyz_gemm.h"
#include "pyz_rng.h"

#define PYZ_LOG_SQRT_2PI 0.918938533204672741780329736406f

// ---------------------------------------------------------------- reductions
__device__ __forceinline__ double pyz_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// sum over the workgroup; result valid in thread 0.  `sm` holds >= 16 doubles.
__device__ __forceinline__ double pyz_block_sum(double v, double *sm) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
  v = pyz_wave_sum(v);
  __syncthreads();
  if (l == 0) sm[w] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < nw; ++i) s += sm[i];
  return s;
}

// sum `n` partials in a fixed order -- called by ALL 64 lanes of one wave (lane l takes
// part[l], part[l + 64], ...; then a shuffle tree); the total is valid in lane 0.
// (A single thread walking the partials serialises n dependent memory round trips.)
// The loads go out eight at a time: a lane's walk is a chain of dependent adds, and one load per
// round trip would make 1024 partials (one per batch row, k_head_rows) cost 16 round trips.
__device__ __forceinline__ double pyz_sum_partials(const double *part, int n) {
  double s = 0.0;
  const int l = threadIdx.x & 63;
  for (int i0 = 0; i0 < n; i0 += 512) {
    double t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = i0 + 64 * j + l;
      t[j] = i < n ? part[i] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s += t[j];
  }
  return pyz_wave_sum(s);
}

__device__ __forceinline__ float pyz_softplus(float x) {
  return x > 20.0f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float pyz_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// Non-finite sentinel: every kernel that finalises a step's loss counts the steps whose loss
// is NaN / Inf in a device counter the host reads through pyz_check_finite.
__device__ __forceinline__ void pyz_note_loss(int *counter, const float v) {
  if (counter && !(fabsf(v) <= 3.0e38f)) atomicAdd(counter, 1);
}

// ---------------------------------------------------------------- step control
__global__ void k_set_ctl(StepCtl *ctl, int batch, float lr, long long n, long long row_off, int i, int slot0, int n_run) {
  ctl->n_run = n_run;
  ctl->batch = batch;
  ctl->lr = lr;
  ctl->n = n;
  ctl->row_off = row_off;
  ctl->i = i;
  ctl->slot0 = slot0;
}

// Short device-resident runs (at most PYZ_INLINE_TAB steps): the per-run tables travel as kernel arguments and
// this one launch writes them and the first step's scalars -- no host-to-device copy in front of the first step.
#define PYZ_INLINE_TAB 32
struct InlineTabs {
  int32_t bs[PYZ_INLINE_TAB + 1];
  float lr[PYZ_INLINE_TAB + 1];
  int n;   // entries (steps + the padding entry)
};
__global__ void k_set_ctl_tabs(StepCtl *ctl, InlineTabs t, int32_t *tab_bs, float *tab_lr, long long n, long long row_off,
                               int slot0) {
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

// The last kernel of a step prepares the OTHER StepCtl slot for the next step
// (ping-pong: nobody reads that slot during this step).
__device__ __forceinline__ void pyz_prepare_next(const StepCtl *ctl, StepCtl *next, const int32_t *tab_bs,
                                                 const float *tab_lr, long long row_stride) {
  const int i2 = ctl->i + 1;
  next->i = i2;
  next->batch = tab_bs[i2];
  next->lr = tab_lr[i2];
  next->n = ctl->n + 1;
  next->row_off = ctl->row_off + row_stride;
  next->slot0 = ctl->slot0;
  next->n_run = ctl->n_run;
}

// ---------------------------------------------------------------- losses
struct LossArgs {
  const float *out_last;   // (P, max_batch, C): logits (softmax last layer) or outputs
  long long pstride;
  int C;
  const void *y;           // int32 labels (SCCE) / float targets (MSE), indexed like the data rows
  float *delta;            // (P, max_batch, C) or nullptr (loss only)
  double *part;            // (P, nblk) partial sums of the per-row loss
  int nblk;
  int act_last;
  const StepCtl *ctl;
  const int32_t *row_idx;
};

// SparseCategoricalCrossentropy on a softmax last layer, reduction 'auto' = mean
// over the batch (Dataset.py:152-159; used at SGLD.py:57, HMC.py:157, BBB.py:122,
// SVGD.py:107): loss_m = logsumexp(z_m) - z_m[y_m]; delta = (softmax - onehot)/B.
__global__ void k_loss_scce(LossArgs g) {
  __shared__ double sm[16];
  const int batch = g.ctl->batch, C = g.C, p = blockIdx.y;
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  double lm = 0.0;
  if (m < batch) {
    const float *z = g.out_last + p * g.pstride + (long long)m * C;
    const long long row = g.row_idx ? (long long)g.row_idx[g.ctl->row_off + m] : (long long)m;
    const int y = reinterpret_cast<const int32_t *>(g.y)[row];
    float mx = z[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, z[c]);
    float se = 0.0f;
    for (int c = 0; c < C; ++c) se += expf(z[c] - mx);
    const float lse = mx + logf(se);
    const float zy = (y >= 0 && y < C) ? z[y] : __builtin_nanf("");
    lm = (double)(lse - zy);
    if (g.delta) {
      float *d = g.delta + p * g.pstride + (long long)m * C;
      const float inv = 1.0f / (float)batch;
      for (int c = 0; c < C; ++c) d[c] = (expf(z[c] - lse) - (c == y ? 1.0f : 0.0f)) * inv;
    }
  }
  const double s = pyz_block_sum(lm, sm);
  if (threadIdx.x == 0) g.part[p * g.nblk + blockIdx.x] = s;
}

// MeanSquaredError: mean over the last axis, then over the batch;
// delta = 2 (yhat - y) / (B * C) * act'(yhat).
__global__ void k_loss_mse(LossArgs g) {
  __shared__ double sm[16];
  const int batch = g.ctl->batch, C = g.C, p = blockIdx.y;
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  double lm = 0.0;
  if (m < batch) {
    const float *o = g.out_last + p * g.pstride + (long long)m * C;
    const long long row = g.row_idx ? (long long)g.row_idx[g.ctl->row_off + m] : (long long)m;
    const float *y = reinterpret_cast<const float *>(g.y) + row * C;
    float acc = 0.0f;
    const float sc = 2.0f / ((float)batch * (float)C);
    float *d = g.delta ? g.delta + p * g.pstride + (long long)m * C : nullptr;
    for (int c = 0; c < C; ++c) {
      const float e = o[c] - y[c];
      acc += e * e;
      if (d) d[c] = sc * e * pyz_act_grad(o[c], g.act_last);
    }
    lm = (double)(acc / (float)C);
  }
  const double s = pyz_block_sum(lm, sm);
  if (threadIdx.x == 0) g.part[p * g.nblk + blockIdx.x] = s;
}

// loss[p] = (sum of the row-loss partials) / batch
// gate (optional): the launch does nothing on steps with gate->n % gate_mod == 0, and writes to loss[gate->slot0 + gate->i]
__global__ void k_loss_finalize_gated(const double *part, int nblk, const StepCtl *ctl, float *loss, int *nonfinite,
                                      const StepCtl *gate, int gate_mod) {
  if (gate->n % gate_mod == 0) return;
  const double tot = pyz_sum_partials(part, nblk);
  if (threadIdx.x == 0) {
    float *lo = loss + gate->slot0 + gate->i;
    lo[0] = (float)(tot / (double)ctl->batch);
    pyz_note_loss(nonfinite, lo[0]);
  }
}

__global__ void k_loss_finalize(const double *part, int nblk, const StepCtl *ctl, float *loss, int *nonfinite) {
  const int p = blockIdx.x;   // one 64-lane wave per particle
  const double tot = pyz_sum_partials(part + p * nblk, nblk);
  if (threadIdx.x == 0) {
    loss[p] = (float)(tot / (double)ctl->batch);
    pyz_note_loss(nonfinite, loss[p]);
  }
}

// ---------------------------------------------------------------- SGD / SGLD
// SGD.step update (SGD.py:66-69): theta -= lr * grad.
__global__ void k_sgd_update(float *theta, const float *grad, long long D, const StepCtl *ctl, const double *part,
                             int nblk, float *loss, int *nonfinite) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const float lr = ctl->lr;
  if (e < D) theta[e] = theta[e] - lr * grad[e];
  if (blockIdx.x == 0 && threadIdx.x < 64) {   // first wave of the launch
    const double tot = pyz_sum_partials(part, nblk);
    if (threadIdx.x == 0) {
      loss[0] = (float)(tot / (double)ctl->batch);
      pyz_note_loss(nonfinite, loss[0]);
    }
  }
}

// SWAG.step update for nets the fused path does not take (SWAG.py:61-92): SGD update, then (when
// `update`) the running moments with count n and one deviation row.
__global__ void k_swag_update(float *theta, float *mean, float *sq_mean, float *dev_row, const float *grad, long long D,
                              int update, const StepCtl *ctl, const double *part, int nblk, float *loss, int *nonfinite) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const float lr = ctl->lr;
  if (e < D) {
    const float th = theta[e] - lr * grad[e];
    theta[e] = th;
    if (update) {
      const float fn = (float)ctl->n, fn1 = fn + 1.0f;
      const float mn = (mean[e] * fn + th) / fn1;
      mean[e] = mn;
      sq_mean[e] = (sq_mean[e] * fn + th * th) / fn1;
      if (dev_row) dev_row[e] = th - mn;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    const double tot = pyz_sum_partials(part, nblk);
    if (threadIdx.x == 0) {
      loss[0] = (float)(tot / (double)ctl->batch);
      pyz_note_loss(nonfinite, loss[0]);
    }
  }
}

// SGLD.step (SGLD.py:64-93), fused over the flat vector:
//   noise = lr * z;  theta += -lr * (grad + noise)
//   mean <- (mean * n + theta) / (n + 1);  sq_mean <- (sq_mean * n + theta^2) / (n + 1)
// Each thread owns four consecutive elements (one Philox call).
struct SgldArgs {
  float *theta, *mean, *sq_mean;
  const float *grad;
  long long D;
  const StepCtl *ctl;
  StepCtl *next;            // may be nullptr (eager)
  const int32_t *tab_bs;
  const float *tab_lr;
  long long row_stride;
  uint64_t seed;
  const float *unit_noise;  // optional injected N(0,1)
  const double *part;
  int nblk;
  float *loss;              // eager: slot 0; run: indexed by ctl->i
  int loss_indexed;
  int *nonfinite;
};

__global__ void k_sgld_update(SgldArgs g) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long e0 = 4 * t;
  const float lr = g.ctl->lr;
  const long long n = g.ctl->n;
  if (e0 < g.D) {
    float z[4];
    if (g.unit_noise) {
#pragma unroll
      for (int j = 0; j < 4; ++j) z[j] = (e0 + j < g.D) ? g.unit_noise[e0 + j] : 0.0f;
    } else {
      const float4 q = pyz_normal4(g.seed, PYZ_STREAM_SGLD, (uint32_t)n, (uint64_t)t);
      z[0] = q.x; z[1] = q.y; z[2] = q.z; z[3] = q.w;
    }
    const float fn = (float)n, fn1 = fn + 1.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long e = e0 + j;
      if (e < g.D) {
        const float noise = lr * z[j];
        const float th = g.theta[e] + (-lr) * (g.grad[e] + noise);
        g.theta[e] = th;
        g.mean[e] = (g.mean[e] * fn + th) / fn1;
        g.sq_mean[e] = (g.sq_mean[e] * fn + th * th) / fn1;
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    const double tot = pyz_sum_partials(g.part, g.nblk);
    if (threadIdx.x == 0) {
      float *lo = g.loss + (g.loss_indexed ? g.ctl->slot0 + g.ctl->i : 0);
      lo[0] = (float)(tot / (double)g.ctl->batch);
      pyz_note_loss(g.nonfinite, lo[0]);
      if (g.next) pyz_prepare_next(g.ctl, g.next, g.tab_bs, g.tab_lr, g.row_stride);
    }
  }
}

// ---------------------------------------------------------------- BBB
struct BbbArgs {
  float *mu, *rho, *w;
  const float *grad;
  long long D;
  float lr, alpha, prior_mean, prior_rho;
  const float *pm_vec, *pr_vec;  // optional per-element prior mean / rho (list-valued GaussianPrior); else the scalars
  uint64_t seed;
  uint32_t step;
  const float *eps;      // optional injected N(0,1)
  double *part_kl;       // (nblk_kl)
  int nblk_kl;
  const double *part_loss;
  int nblk_loss;
  const StepCtl *ctl;
  float *cost;           // [0] = cost, [1] = data loss, [2] = log q - log p
  int chained;           // device-resident run: the Philox step is ctl->n
};

__device__ __forceinline__ void pyz_bbb_eps(const BbbArgs &g, long long t, float *z) {
  const long long e0 = 4 * t;
  if (g.eps) {
#pragma unroll
    for (int j = 0; j < 4; ++j) z[j] = (e0 + j < g.D) ? g.eps[e0 + j] : 0.0f;
  } else {
    const float4 q = pyz_normal4(g.seed, PYZ_STREAM_BBB, g.chained ? (uint32_t)g.ctl->n : g.step, (uint64_t)t);
    z[0] = q.x; z[1] = q.y; z[2] = q.z; z[3] = q.w;
  }
}

// BBB._update_weights (BBB.py:218-246): w = mu + softplus(rho) * eps, and the two
// Gaussian log-likelihood sums of _cost_function (BBB.py:51-124):
//   sum log N(w; mu, softplus rho) - sum log N(w; mu_p, softplus rho_p).
__global__ void k_bbb_sample(BbbArgs g) {
  __shared__ double sm[16];
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long e0 = 4 * t;
  double kl = 0.0;
  if (e0 < g.D) {
    float z[4], wv[4];
    pyz_bbb_eps(g, t, z);
    // a scalar prior: its softplus and logarithm once per thread, not per element (three of the seven transcendental
    // evaluations per element of this VALU-bound kernel)
    const float sp_s = pyz_softplus(g.prior_rho), lsp_s = logf(sp_s);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long e = e0 + j;
      wv[j] = 0.0f;
      if (e < g.D) {
        const float pmean = g.pm_vec ? g.pm_vec[e] : g.prior_mean;
        float sp = sp_s, lsp = lsp_s;
        if (g.pr_vec) {   // uniform
          sp = pyz_softplus(g.pr_vec[e]);
          lsp = logf(sp);
        }
        const float mu = g.mu[e], sg = pyz_softplus(g.rho[e]);
        const float w = z[j] * sg + mu;
        wv[j] = w;
        const float a = (w - mu) / sg, b = (w - pmean) / sp;
        const float lq = -0.5f * a * a - logf(sg) - PYZ_LOG_SQRT_2PI;
        const float lp = -0.5f * b * b - lsp - PYZ_LOG_SQRT_2PI;
        kl += (double)lq - (double)lp;
      }
    }
    // the sampled weights are the next kernel's operand: one 16-byte write-through store per thread (whole lines leave L2
    // while the kernel runs; plain stores would wait for their write-back at its end -- see pyz_st)
    if (e0 + 3 < g.D && (reinterpret_cast<uintptr_t>(g.w) & 15) == 0) {
      const f32x4 vv = {wv[0], wv[1], wv[2], wv[3]};
      float *d = g.w + e0;
      asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(d), "v"(vv) : "memory");
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (e0 + j < g.D) g.w[e0 + j] = wv[j];
    }
  }
  const double s = pyz_block_sum(kl, sm);
  if (threadIdx.x == 0) g.part_kl[blockIdx.x] = s;
}

// BBB.step gradients and update (BBB.py:152-201), closed forms of the three
// tape.gradient calls (derivation cross-checked in tests/test_oracle_kat.py):
//   d mu  = alpha (w - mu) / sigma^2
//   d rho = alpha (-1/sigma + (w - mu)^2 / sigma^3) sigmoid(rho)
//   d w   = d loss/d w + alpha (-(w - mu)/sigma^2 + (w - mu_p)/sigma_p^2)
//   mu  <- mu  - lr (d mu + d w);   rho <- rho - lr (eps sigmoid(rho) d w + d rho)
__global__ void k_bbb_update(BbbArgs g) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long e0 = 4 * t;
  if (e0 < g.D) {
    float z[4];
    pyz_bbb_eps(g, t, z);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long e = e0 + j;
      if (e < g.D) {
        const float pmean = g.pm_vec ? g.pm_vec[e] : g.prior_mean;
        const float sp = pyz_softplus(g.pr_vec ? g.pr_vec[e] : g.prior_rho), isp2 = 1.0f / (sp * sp);
        const float mu = g.mu[e], rho = g.rho[e], w = g.w[e];
        const float sg = pyz_softplus(rho), sig = pyz_sigmoid(rho);
        const float d = w - mu, is2 = 1.0f / (sg * sg);
        const float g_mu = g.alpha * d * is2;
        const float g_rho = g.alpha * (-1.0f / sg + d * d * is2 / sg) * sig;
        const float g_w = g.grad[e] + g.alpha * (-d * is2 + (w - pmean) * isp2);
        g.mu[e] = mu - g.lr * (g_mu + g_w);
        g.rho[e] = rho - g.lr * (z[j] * sig * g_w + g_rho);
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    const double sl = pyz_sum_partials(g.part_loss, g.nblk_loss);
    const double sk = pyz_sum_partials(g.part_kl, g.nblk_kl);
    if (threadIdx.x == 0) {
      const float loss = (float)(sl / (double)g.ctl->batch);
      const float kl = (float)sk;
      g.cost[0] = loss + g.alpha * kl;
      g.cost[1] = loss;
      g.cost[2] = kl;
    }
  }
}

// ---------------------------------------------------------------- HMC
struct HmcArgs {
  float *q;              // (P, D)
  float *p;              // (P, D) momentum
  float *qsave;          // (P, D)
  const float *grad;     // (P, D) d loss / d q (mean loss)
  long long D;
  float m, prior_mean, prior_sigma, n_train;
  const float *pm_vec, *ps_vec;  // optional per-element prior mean / sigma (D), shared by the chains
  float kick1, kick2;    // p -= kick1 * dU; p -= kick2 * dU  (kick2 = 0 when unused)
  float drift;           // q += drift * p   (0 when unused)
  uint64_t seed;
  uint32_t step;
  const float *unit_p;   // optional injected N(0,1), (P, D)
  double *part;          // (P, 2, nblk): [0] = sum log N(q), [1] = sum p^2
  int nblk;
};

// _sample_kinetic_energy (HMC.py:168-171): p = m * z;  snapshot q (HMC.py:81);
// partial sums for K0 (HMC.py:161-166) and the prior part of U0 (HMC.py:150-154).
__global__ void k_hmc_begin(HmcArgs g) {
  __shared__ double sm[16];
  const int c = blockIdx.y;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long e0 = 4 * t;
  double sp2 = 0.0, slp = 0.0;
  if (e0 < g.D) {
    float z[4];
    if (g.unit_p) {
#pragma unroll
      for (int j = 0; j < 4; ++j) z[j] = (e0 + j < g.D) ? g.unit_p[c * g.D + e0 + j] : 0.0f;
    } else {
      const float4 v = pyz_normal4(g.seed, PYZ_STREAM_HMC + 16u * (uint32_t)c, g.step, (uint64_t)t);
      z[0] = v.x; z[1] = v.y; z[2] = v.z; z[3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long e = e0 + j;
      if (e < g.D) {
        const float pmu = g.pm_vec ? g.pm_vec[e] : g.prior_mean, psg = g.ps_vec ? g.ps_vec[e] : g.prior_sigma;
        const float ls = logf(psg);
        const float q = g.q[c * g.D + e];
        const float pm = g.m * z[j];
        g.p[c * g.D + e] = pm;
        g.qsave[c * g.D + e] = q;
        sp2 += (double)(pm * pm);
        const float a = (q - pmu) / psg;
        slp += (double)(-0.5f * a * a - ls - PYZ_LOG_SQRT_2PI);
      }
    }
  }
  const double s0 = pyz_block_sum(slp, sm);
  const double s1 = pyz_block_sum(sp2, sm);
  if (threadIdx.x == 0) {
    g.part[(c * 2 + 0) * g.nblk + blockIdx.x] = s0;
    g.part[(c * 2 + 1) * g.nblk + blockIdx.x] = s1;
  }
}

// _step_p (HMC.py:128-136) with dU/dq = (q - mu_p)/sigma_p^2 + N * dloss/dq, optionally
// followed by _step_q (HMC.py:138-141) q += (eps/m) p.  Two sequential kicks keep the
// reference's rounding when a full and a half kick share one gradient (HMC.py:86-87).
__global__ void k_hmc_kick_drift(HmcArgs g) {
  const int c = blockIdx.y;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= g.D) return;
  const long long o = c * g.D + e;
  const float q = g.q[o];
  const float pmu = g.pm_vec ? g.pm_vec[e] : g.prior_mean, psg = g.ps_vec ? g.ps_vec[e] : g.prior_sigma;
  const float dU = (q - pmu) / (psg * psg) + g.n_train * g.grad[o];
  float p = g.p[o];
  p = p - g.kick1 * dU;
  if (g.kick2 != 0.0f) p = p - g.kick2 * dU;
  g.p[o] = p;
  if (g.drift != 0.0f) g.q[o] = q + g.drift * p;
}

// partial sums for K1 and the prior part of U1 at the end of the trajectory
__global__ void k_hmc_end_energy(HmcArgs g) {
  __shared__ double sm[16];
  const int c = blockIdx.y;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  double sp2 = 0.0, slp = 0.0;
  if (e < g.D) {
    const float q = g.q[c * g.D + e], p = g.p[c * g.D + e];
    sp2 = (double)(p * p);
    const float pmu = g.pm_vec ? g.pm_vec[e] : g.prior_mean, psg = g.ps_vec ? g.ps_vec[e] : g.prior_sigma;
    const float a = (q - pmu) / psg;
    slp = (double)(-0.5f * a * a - logf(psg) - PYZ_LOG_SQRT_2PI);
  }
  const double s0 = pyz_block_sum(slp, sm);
  const double s1 = pyz_block_sum(sp2, sm);
  if (threadIdx.x == 0) {
    g.part[(c * 2 + 0) * g.nblk + blockIdx.x] = s0;
    g.part[(c * 2 + 1) * g.nblk + blockIdx.x] = s1;
  }
}

// energies[c*4 + {0,1}] = {U, K} from the partials and the mean loss (HMC.py:149-166)
__global__ void k_hmc_energy_finalize(const double *part, int nblk, const float *loss, float n_train, float m,
                                      float *energies, int slot) {
  const int c = blockIdx.x;   // one 64-lane wave per chain
  const double d0 = pyz_sum_partials(part + (c * 2 + 0) * nblk, nblk);
  const double d1 = pyz_sum_partials(part + (c * 2 + 1) * nblk, nblk);
  if (threadIdx.x != 0) return;
  const float slp = (float)d0;
  const float sp2 = (float)d1;
  float U = 0.0f - slp;              // potential_energy -= reduce_sum(log_prob)
  U = U + loss[c] * n_train;         // += loss * cardinality
  const float K = (1.0f / (2.0f * m)) * sp2;
  energies[c * 8 + 2 * slot + 0] = U;
  energies[c * 8 + 2 * slot + 1] = K;
  energies[c * 8 + 4 + slot] = loss[c];
}

// Metropolis test (HMC.py:91): accept iff burning or u < exp(K0 + U0 - K1 - U1).
// stats (P, 8) = {accepted, loss, U0, K0, U1, K1, log_ratio, 0}
__global__ void k_hmc_accept(const float *energies, const float *uniform, int burning, float *stats, int P) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= P) return;
  const float U0 = energies[c * 8 + 0], K0 = energies[c * 8 + 1], U1 = energies[c * 8 + 2], K1 = energies[c * 8 + 3];
  const float lr = K0 + U0 - K1 - U1;
  const bool acc = burning || (uniform[c] < expf(lr));
  stats[c * 8 + 0] = acc ? 1.0f : 0.0f;
  stats[c * 8 + 1] = acc ? energies[c * 8 + 5] : energies[c * 8 + 4];
  stats[c * 8 + 2] = U0;
  stats[c * 8 + 3] = K0;
  stats[c * 8 + 4] = U1;
  stats[c * 8 + 5] = K1;
  stats[c * 8 + 6] = lr;
  stats[c * 8 + 7] = 0.0f;
}

// rejected chains get their snapshot back (HMC.py:97-101)
__global__ void k_hmc_restore(float *q, const float *qsave, const float *stats, long long D) {
  const int c = blockIdx.y;
  if (stats[c * 8] != 0.0f) return;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < D) q[c * D + e] = qsave[c * D + e];
}

// ---------------------------------------------------------------- SVGD
struct SvgdArgs {
  float *particles;       // (n_local, D) this rank's rows (updated)
  const float *all;       // (M, D) matrix the kernel row is evaluated against
  float *all_rw;          // == all when the sweep updates it in place (Gauss-Seidel, one GPU), else nullptr
  float *adam_m, *adam_v; // (n_local, D)
  const float *grad;      // (n_local, D) loss gradients
  long long D;
  int M, n_local, row0;
  int i_local;            // row handled by this launch (Gauss-Seidel) ; -1 = all rows (Jacobi)
  float lr_t;             // lr * sqrt(1 - b2^t) / (1 - b1^t)
  float gamma;
  double *part;           // (n_rows, nblk, M) partial squared distances
  int nblk;
};

// partial squared distances  sum_d (x_i[d] - x_j[d])^2  of row i against every j, in
// float64 (SVGD.py:198-201 evaluates them on the float64 particle matrix).
// grid = (nblk, ceil(M/8), rows), 256 threads.  A workgroup owns 2048 consecutive elements of D
// and EIGHT rows j: each wave keeps its 512-element slice of x_i in registers and streams the
// eight x_j slices with 16-B loads, all issued before the first use (one memory round trip per
// workgroup); lanes combine with shuffles, the 4 waves through LDS; one partial per workgroup.
// The (D-chunk x row-group) grid puts thousands of waves in flight: the pass is bandwidth-bound.
#define PYZ_SVGD_PASS 2
#define PYZ_SVGD_BLOCK_ELEMS (4 * PYZ_SVGD_PASS * 256)   // elements of D per workgroup

__global__ void __launch_bounds__(256) k_svgd_dist(SvgdArgs g) {
  __shared__ double wsum[4][8];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int il = g.i_local >= 0 ? g.i_local : blockIdx.z;
  const int i = g.row0 + il;
  const int j0 = blockIdx.y * 8;
  const long long base = (long long)blockIdx.x * PYZ_SVGD_BLOCK_ELEMS + (long long)w * PYZ_SVGD_PASS * 256;
  const float *xi_p = g.all + (long long)i * g.D;
  const bool vec_ok = (g.D % 4 == 0);   // rows 16-B aligned
  float xi[PYZ_SVGD_PASS][4], xj[8][PYZ_SVGD_PASS][4];
#pragma unroll
  for (int ps = 0; ps < PYZ_SVGD_PASS; ++ps) {
    const long long d = base + ps * 256 + 4 * l;
#pragma unroll
    for (int q = 0; q < 4; ++q) xi[ps][q] = (d + q < g.D) ? xi_p[d + q] : 0.0f;
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const float *xj_p = g.all + (long long)min(j0 + u, g.M - 1) * g.D;
#pragma unroll
    for (int ps = 0; ps < PYZ_SVGD_PASS; ++ps) {
      const long long d = base + ps * 256 + 4 * l;
      if (vec_ok && d + 3 < g.D) {
        const float4 v = *reinterpret_cast<const float4 *>(xj_p + d);
        xj[u][ps][0] = v.x; xj[u][ps][1] = v.y; xj[u][ps][2] = v.z; xj[u][ps][3] = v.w;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) xj[u][ps][q] = (d + q < g.D) ? xj_p[d + q] : 0.0f;
      }
    }
  }
  double s[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    double acc = 0.0;
#pragma unroll
    for (int ps = 0; ps < PYZ_SVGD_PASS; ++ps)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool in = base + ps * 256 + 4 * l + q < g.D;
        const double df = in ? (double)xi[ps][q] - (double)xj[u][ps][q] : 0.0;
        acc += df * df;
      }
    s[u] = pyz_wave_sum(acc);
  }
  if (l == 0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) wsum[w][u] = s[u];
  }
  __syncthreads();
  double *outp = g.part + ((long long)(g.i_local >= 0 ? 0 : blockIdx.z) * g.nblk + blockIdx.x) * g.M;
  if (threadIdx.x < 8 && j0 + (int)threadIdx.x < g.M)
    outp[j0 + threadIdx.x] = (wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + (wsum[2][threadIdx.x] + wsum[3][threadIdx.x]);
}

// phi_i = ( (sum_j K_ij) g_i + 2 gamma sum_j K_ij (x_i - x_j) ) / M   (SVGD.py:54-68)
// followed by the particle's Keras legacy Adam step (SVGD.py:120, Appendix A3):
//   m += (phi - m)(1 - b1); v += (phi^2 - v)(1 - b2); x -= lr_t m / (sqrt(v) + 1e-7)
// Rows j whose kernel value underflowed to exactly 0 (and j = i, whose difference is 0)
// contribute exactly nothing to the repulsion sum and are not read: bit-identical, and in
// the regime of far-apart particles the sweep touches one row instead of M.
// grid = (ceil(D/256), rows), 256 threads; dynamic LDS = (5 M) doubles.
__global__ void __launch_bounds__(256) k_svgd_update(SvgdArgs g) {
  extern __shared__ double sd[];  // [M] K_ij, then [4][M] partial sums
  const int il = g.i_local >= 0 ? g.i_local : blockIdx.y;
  const int i = g.row0 + il;
  const double *pp = g.part + (long long)(g.i_local >= 0 ? 0 : blockIdx.y) * g.nblk * g.M;
  double *ps4 = sd + g.M;
  {  // squared distances: the nblk partials of row j are summed by 4 threads (stride-4 slices), fixed order
    const int q = threadIdx.x >> 6;
    for (int j = threadIdx.x & 63; j < g.M; j += 64) {
      double s = 0.0;
      for (int b = q; b < g.nblk; b += 4) s += pp[(long long)b * g.M + j];
      ps4[q * g.M + j] = s;
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < g.M; j += blockDim.x)
    sd[j] = exp(-(double)g.gamma * ((ps4[j] + ps4[g.M + j]) + (ps4[2 * g.M + j] + ps4[3 * g.M + j])));
  __syncthreads();
  float ksum = 0.0f;
  for (int j = 0; j < g.M; ++j) ksum += (float)sd[j];
  const long long d = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= g.D) return;
  const float xi = g.all[(long long)i * g.D + d];
  double rep = 0.0;
  for (int j = 0; j < g.M; ++j) {
    const double kj = sd[j];
    if (kj == 0.0 || j == i) continue;   // wave-uniform
    rep += kj * ((double)xi - (double)g.all[(long long)j * g.D + d]);
  }
  rep *= 2.0 * (double)g.gamma;
  const long long o = (long long)il * g.D + d;
  const float phi = (ksum * g.grad[o] + (float)rep) / (float)g.M;
  float m = g.adam_m[o], v = g.adam_v[o];
  m = m + (phi - m) * (1.0f - 0.9f);
  v = v + (phi * phi - v) * (1.0f - 0.999f);
  g.adam_m[o] = m;
  g.adam_v[o] = v;
  const float xn = xi - g.lr_t * m / (sqrtf(v) + 1e-7f);   // (x_i as the matrix holds it: `particles` is only written)
  g.particles[o] = xn;
  if (g.all_rw && g.all_rw != g.particles) g.all_rw[(long long)i * g.D + d] = xn;
}

// ---------------------------------------------------------------- Jacobi sweep, all rows at once
// The per-row kernels above read the (M, D) particle matrix once per ROW: a Jacobi sweep of M = 64
// rows moves 64 x 40.7 MB twice (distances, update).  With every row updated from the same snapshot
// the matrix needs to be read once per pass:
//   k_svgd_dist_tile   a workgroup owns `range` consecutive elements of D (a multiple of 128 chosen so that the
//                      grid is one round of the 256 CUs), stages 128-element slabs
//                      of ALL particles in LDS ([element][particle]) and accumulates the float64
//                      squared-distance partials of its 4 x 4 pair blocks (thread (ti, tj): local rows
//                      4 ti .. + 3 against particles 4 tj .. + 3), the reference's arithmetic
//                      (SVGD.py:198-201: differences and squares in float64);
//   k_svgd_kmat        sums the partials of a row in block order, K_ij = exp(-gamma d_ij), sum_j K_ij;
//   k_svgd_update_tile a thread owns ONE element d: it keeps x_j[d] of all particles in registers and
//                      produces phi_i[d] + the Adam step for every local row i (the formulas and the
//                      summation order over j of k_svgd_update).
// Needs M <= 64 and local rows in multiples of 4 starting at a multiple of 4; else the per-row kernels run.
#define PYZ_SV_E 128   // elements staged per pass

struct SvgdTileArgs {
  float *particles;        // (n_local, D) this rank's rows (updated)
  const float *all;        // (M, D) snapshot the kernel matrix is evaluated on
  float *adam_m, *adam_v;  // (n_local, D)
  const float *grad;       // (n_local, D)
  long long D;
  int M, n_local, row0;
  float lr_t, gamma;
  double *part;            // (n_local, nblk, 64) partial squared distances
  int nblk;
  int range;               // elements of D per workgroup of k_svgd_dist_tile (multiple of PYZ_SV_E)
  double *kmat;            // (n_local, 64) kernel values (0 past M)
  float *ksum;             // (n_local) sum_j K_ij, float (the factor of the loss gradient)
  double *ksumd;           // (n_local) the same row sum in float64, in the update kernel's summation order
  double *diag;            // (unused since round 3: the Gram form leaves partial squared distances like the pairwise form)
  // median-heuristic bandwidth (SVGD.py:165-181) only: the squared distances of ALL pairs and the bandwidth they give
  double *dmat;            // (M, 64) squared distances (0 past M), or nullptr
  double *gamma_dev;       // [1] gamma = 1 / (2 h^2) = log(M + 1) / median(d), or nullptr: the fixed `gamma`
  // the step's loss rides in k_svgd_update_tile (its launch is 4 - 5 us for one thread's work as a kernel of its own; not in
  // k_svgd_kmat: the kernel matrix of a snapshot may be built on another stream WHILE the gradient pass produces the losses):
  // loss_out[0] = sum_i loss_in[i] / M over the local particles (k_svgd_loss), or nullptr
  const float *loss_in;
  float *loss_out;
  // the distance pass split over the ELEMENTS (ranks of a sharded run each take some of the PYZ_SVGD_GROUPS groups of
  // consecutive blocks): blk0 = global number of the launch's first block; groups = (PYZ_SVGD_GROUPS, 64, 64) group sums of
  // the partials gathered from all ranks (k_svgd_kmat then reads them instead of `part`), or nullptr
  int blk0;
  const double *groups;
};

// The partial squared distances of the nblk blocks are summed in PYZ_SVGD_GROUPS groups of consecutive blocks, each group
// as two interleaved halves in block order, the groups as a balanced tree: ONE order whether a device sums all groups
// itself (k_svgd_kmat) or receives some of them from other ranks (k_svgd_group_reduce there, then k_svgd_kmat).
// (PYZ_SVGD_GROUPS: include/pyz.h)
__device__ __forceinline__ double pyz_svgd_half_sum(const double *pp, const int b_lo, const int b_hi, const int h) {
  double s = 0.0;
  for (int b0 = b_lo + h; b0 < b_hi; b0 += 32) {   // all loads of up to 32 blocks per group in one round trip
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = b0 + 2 * u < b_hi ? pp[(long long)(b0 + 2 * u) * 64] : 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) s += v[u];
  }
  return s;
}

// group sums of rows [0, n_rows) for groups [g_lo, g_lo + gridDim.y): out[(g * 64 + i) * 64 + j]
__global__ void __launch_bounds__(128) k_svgd_group_reduce(const double *part, const int nblk, const int g_lo, double *out) {
  __shared__ double sh[64];
  const int i = blockIdx.x, g8 = g_lo + blockIdx.y, j = threadIdx.x & 63, h = threadIdx.x >> 6;
  const int nb8 = (nblk + PYZ_SVGD_GROUPS - 1) / PYZ_SVGD_GROUPS, b_lo = g8 * nb8, b_hi = min(b_lo + nb8, nblk);
  const double s = pyz_svgd_half_sum(part + (long long)i * nblk * 64 + j, b_lo, b_hi, h);
  if (h == 1) sh[j] = s;
  __syncthreads();
  if (h == 0) out[((long long)g8 * 64 + i) * 64 + j] = s + sh[j];
}

__global__ void __launch_bounds__(256) k_svgd_dist_tile(SvgdTileArgs g) {
  __shared__ float xs[PYZ_SV_E][64];
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  // pair blocks: rectangular (local rows x all particles) for a shard; for the whole matrix d_ij = d_ji
  // bit for bit, so only the 136 blocks with ti <= tj are computed (threads 0 .. 135) and mirrored
  const bool whole = g.n_local == g.M && g.row0 == 0;
  int ti = t >> 4, tj = t & 15;
  bool active = 4 * ti < g.n_local;
  if (whole) {
    int rem = t;
    ti = 0;
    while (ti < 16 && rem >= 16 - ti) {
      rem -= 16 - ti;
      ++ti;
    }
    tj = ti + rem;
    active = ti < 16 && 4 * tj < g.M;  // (ti <= tj: the row block is inside the matrix too)
    if (!active) ti = tj = 0;
  }
  const int blk = blockIdx.x + g.blk0;
  const long long base = (long long)blk * g.range;
  const bool vec_ok = (g.D % 4 == 0);
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  for (int ps = 0; ps < g.range / PYZ_SV_E; ++ps) {
    const long long e0 = base + (long long)ps * PYZ_SV_E;
    if (e0 >= g.D) break;  // uniform
    // lane = particle: each lane brings 8 x 16 bytes of its row (wave w: elements 32 w .. 32 w + 31 of the slab)
    {
      const float *row = g.all + (long long)min(l, g.M - 1) * g.D;
      float4 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const long long d = e0 + 32 * w + 4 * q;
        if (l < g.M && vec_ok && d + 3 < g.D) {
          v[q] = *reinterpret_cast<const float4 *>(row + d);
        } else {
          v[q].x = (l < g.M && d + 0 < g.D) ? row[d + 0] : 0.0f;
          v[q].y = (l < g.M && d + 1 < g.D) ? row[d + 1] : 0.0f;
          v[q].z = (l < g.M && d + 2 < g.D) ? row[d + 2] : 0.0f;
          v[q].w = (l < g.M && d + 3 < g.D) ? row[d + 3] : 0.0f;
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int e = 32 * w + 4 * q;
        xs[e + 0][l] = v[q].x;
        xs[e + 1][l] = v[q].y;
        xs[e + 2][l] = v[q].z;
        xs[e + 3][l] = v[q].w;
      }
    }
    __syncthreads();
    if (active) {
#pragma unroll 4
      for (int e = 0; e < PYZ_SV_E; ++e) {
        const float4 xi = *reinterpret_cast<const float4 *>(&xs[e][g.row0 + 4 * ti]);
        const float4 xj = *reinterpret_cast<const float4 *>(&xs[e][4 * tj]);
        const double di[4] = {(double)xi.x, (double)xi.y, (double)xi.z, (double)xi.w};
        const double dj[4] = {(double)xj.x, (double)xj.y, (double)xj.z, (double)xj.w};
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const double df = di[a] - dj[b];
            acc[a][b] += df * df;
          }
      }
    }
    __syncthreads();
  }
  if (active) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      double *o = g.part + ((long long)(4 * ti + a) * g.nblk + blk) * 64 + 4 * tj;
#pragma unroll
      for (int b = 0; b < 4; ++b) o[b] = acc[a][b];
    }
    if (whole && ti != tj) {  // the mirrored block: rows 4 tj .. 4 tj + 3 < M (M is a multiple of 4 here)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        double *o = g.part + ((long long)(4 * tj + b) * g.nblk + blk) * 64 + 4 * ti;
#pragma unroll
        for (int a = 0; a < 4; ++a) o[a] = acc[a][b];
      }
    }
  }
}

// one workgroup per local row: d_ij = sum of the row's partials (four interleaved slices of the blocks, each
// in block order, combined in a fixed order), K_ij, sum_j K_ij (float, j ascending)
// Gram form of the distance pass on the float64 matrix cores: G = X X^T over the workgroup's range of D with
// v_mfma_f64_16x16x4_f64 (float32 inputs converted exactly; their products are exact in float64, only the
// sums round), the squared distances follow in k_svgd_kmat.  Same staging as k_svgd_dist_tile; wave w takes
// elements 32 w .. 32 w + 31 of each slab (8 reduction steps of 4), the four partial Grams are added through
// LDS in wave order.  The pairwise float64 VALU form above is 128 us at C5; this one is bound by the read of
// the matrix.  Lane layout of the instruction (checked once per process by k_probe_mfma_f64):
//   A: row = lane % 16, k = lane / 16;  B: k = lane / 16, col = lane % 16;  D[r]: row = lane / 16 + 4 r, col = lane % 16
//   (not the 4 (lane / 16) + r of the float32 16x16x4 instruction).
typedef double pyz_f64x4 __attribute__((ext_vector_type(4)));

__global__ void k_probe_mfma_f64(int *out) {  // out[0..255]: row of D[r] per lane, out[256..511]: column
  const int l = threadIdx.x & 63;
  pyz_f64x4 c = {0.0, 0.0, 0.0, 0.0};
  const pyz_f64x4 dr = __builtin_amdgcn_mfma_f64_16x16x4f64((l >> 4) == 0 ? (double)(l & 15) : 0.0, (l >> 4) == 0 ? 1.0 : 0.0, c, 0, 0, 0);
  const pyz_f64x4 dc = __builtin_amdgcn_mfma_f64_16x16x4f64((l >> 4) == 0 ? 1.0 : 0.0, (l >> 4) == 0 ? (double)(l & 15) : 0.0, c, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    out[4 * l + r] = (int)dr[r];
    out[256 + 4 * l + r] = (int)dc[r];
  }
}

// LDS-only barrier: __syncthreads() also waits for every global load in flight (the fence in it is not address-space
// aware); the kernels below keep loads in flight across their barriers on purpose.
__device__ __forceinline__ void pyz_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Eight waves; a slab is 128 elements of all 64 particles.  Lanes run along the ELEMENTS when fetching (two consecutive ones
// per lane: 512 contiguous bytes of one row per load instruction, wave w brings particles w, w + 8, ...; lane = particle,
// 64 rows 636 KB apart per instruction, read at 1.4 TB/s), the slab is transposed into LDS ([element][particle]), wave w
// multiplies elements 16 w .. 16 w + 15 of it (four reduction steps of 4).  Two slabs are in flight in registers and the
// LDS slab is double-buffered: ONE LDS-only barrier per slab.  (Round 2's form -- four waves, one slab in flight, two
// __syncthreads per slab, always the 10 blocks of the whole matrix -- took 31 us at C5 whatever the number of local rows:
// one wave per SIMD waiting for its own loads.)
//   SYM       the 10 blocks of 16 x 16 on and above the diagonal, mirrored (G is symmetric bit for bit): any row range
//   NRB 1 / 2 a shard whose local rows lie in one / two row blocks of 16: only those blocks x all four column blocks
// Every G_ij is the same sequence of matrix instructions and the same fixed-order sum over the waves in all forms, so a
// shard's rows equal the whole matrix's bit for bit.  The squared norms are summed on the vector ALU (the diagonal
// blocks are not computed by a shard).
#define PYZ_GRAM_RS 65   // row stride (floats) of a staged slab: odd, so that a wave's transposing writes spread over the banks
#define PYZ_GRAM_SLAB (PYZ_SV_E * PYZ_GRAM_RS)
static inline size_t pyz_svgd_gram_lds_bytes() { return 65536 + 16384 + 512; }   // epilogue: 8 waves x 4 blocks x 256 doubles + the norms' partials + the norms
static_assert(2 * PYZ_GRAM_SLAB * sizeof(float) <= 65536 + 16384, "slabs fit");

template <int NRB, bool SYM>
__global__ void __launch_bounds__(512) k_svgd_gram_tile(SvgdTileArgs g) {
  extern __shared__ double gram_lds[];
  float *const xs0 = reinterpret_cast<float *>(gram_lds);
  float *const xs1 = xs0 + PYZ_GRAM_SLAB;
  constexpr int RS = PYZ_GRAM_RS;
  constexpr int NR = SYM ? 4 : NRB;
  const int t = threadIdx.x, w = pyz_wave_id(), l = t & 63;
  const int blk = blockIdx.x + g.blk0;
  const long long base = (long long)blk * g.range;
  const int rb_lo = SYM ? 0 : g.row0 >> 4;   // uniform
  pyz_f64x4 acc[NR][4];
#pragma unroll
  for (int a = 0; a < NR; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = pyz_f64x4{0.0, 0.0, 0.0, 0.0};
  double nrm[4] = {0.0, 0.0, 0.0, 0.0};
  const bool pair_ok = (g.D % 2 == 0) && ((reinterpret_cast<uintptr_t>(g.all) & 7) == 0);   // every row 8-byte aligned
  auto fetch = [&](const long long e0, float2 (&v)[8]) {
    const long long d = e0 + 2 * l;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = w + 8 * u;   // wave-uniform
      const float *p = g.all + (long long)min(j, g.M - 1) * g.D + d;
      if (j < g.M && pair_ok && d + 1 < g.D) {
        v[u] = *reinterpret_cast<const float2 *>(p);
      } else {
        v[u].x = (j < g.M && d < g.D) ? p[0] : 0.0f;
        v[u].y = (j < g.M && d + 1 < g.D) ? p[1] : 0.0f;
      }
    }
  };
  int n_slabs = 0;
  if (base < g.D) {
    const long long left = (g.D - base + PYZ_SV_E - 1) / PYZ_SV_E;
    n_slabs = (int)min((long long)(g.range / PYZ_SV_E), left);
  }
  auto body = [&](float2 (&v)[8], float *xs, const int ps) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      xs[(2 * l) * RS + w + 8 * u] = v[u].x;
      xs[(2 * l + 1) * RS + w + 8 * u] = v[u].y;
    }
    pyz_lds_barrier();
    if (ps + 2 < n_slabs) fetch(base + (long long)(ps + 2) * PYZ_SV_E, v);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const float *xe = xs + (16 * w + 4 * ks + (l >> 4)) * RS;
      double a[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) a[b] = (double)xe[(l & 15) + 16 * b];
#pragma unroll
      for (int b = 0; b < 4; ++b) nrm[b] = fma(a[b], a[b], nrm[b]);
      if constexpr (SYM) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
          for (int cb = rb; cb < 4; ++cb)
            acc[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rb], a[cb], acc[rb][cb], 0, 0, 0);
      } else {
#pragma unroll
        for (int r = 0; r < NRB; ++r) {
          const int rb = rb_lo + r;   // uniform
          const double ar = rb == 0 ? a[0] : rb == 1 ? a[1] : rb == 2 ? a[2] : a[3];
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) acc[r][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, a[cb], acc[r][cb], 0, 0, 0);
        }
      }
    }
  };
  float2 v0[8], v1[8];
  PYZ_STAMP(4, 0);
  if (n_slabs > 0) fetch(base, v0);
  if (n_slabs > 1) fetch(base + PYZ_SV_E, v1);
  for (int ps = 0; ps < n_slabs; ps += 2) {
    body(v0, xs0, ps);
    if (ps == 0) PYZ_STAMP(4, 1);
    if (ps + 1 < n_slabs) body(v1, xs1, ps + 1);
  }
  PYZ_STAMP(4, 2);
  // ---- the eight waves' partial Grams, four blocks per round through LDS, summed in wave order by the thread that stores
  //      the element; then the norms (32 partials per particle: wave-major, reduction lane minor)
  double *const red = gram_lds;                 // [8 waves][4 blocks][256]
  double *const nred = gram_lds + 8 * 4 * 256;  // [8 waves * 4 lanes][64]
  double *const nsum = nred + 32 * 64;          // [64] squared norms over the workgroup's elements
  constexpr int NBLK = SYM ? 10 : 4 * NRB;
  pyz_lds_barrier();   // the last slab's fragment reads are done
#pragma unroll
  for (int b = 0; b < 4; ++b) nred[(4 * w + (l >> 4)) * 64 + 16 * b + (l & 15)] = nrm[b];
  pyz_lds_barrier();
  if (t < 64) {
    double s = nred[t];
#pragma unroll
    for (int q = 1; q < 32; ++q) s += nred[q * 64 + t];
    nsum[t] = s;
  }
#pragma unroll
  for (int r0 = 0; r0 < NBLK; r0 += 4) {
    if (r0) pyz_lds_barrier();   // the previous round's sums have been read
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int bi = r0 + q;   // compile-time block number: (rb, cb) of the 10 upper blocks in row-major order, or (r, cb)
      if (bi >= NBLK) break;
      int rb, cb;
      if constexpr (SYM) {
        rb = bi < 4 ? 0 : bi < 7 ? 1 : bi < 9 ? 2 : 3;
        cb = bi < 4 ? bi : bi < 7 ? bi - 3 : bi < 9 ? bi - 5 : 3;
      } else {
        rb = bi >> 2;
        cb = bi & 3;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(w * 4 + q) * 256 + r * 64 + l] = acc[rb][cb][r];
    }
    pyz_lds_barrier();   // (the first round's also publishes nsum)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = t + 512 * h, q = e >> 8, r = (e >> 6) & 3, ll = e & 63;
      const int bi = r0 + q;
      if (bi < NBLK) {
        int rb, cb;
        if (SYM) {
          rb = bi < 4 ? 0 : bi < 7 ? 1 : bi < 9 ? 2 : 3;
          cb = bi < 4 ? bi : bi < 7 ? bi - 3 : bi < 9 ? bi - 5 : 3;
        } else {
          rb = rb_lo + (bi >> 2);
          cb = bi & 3;
        }
        double s = red[q * 256 + r * 64 + ll];
#pragma unroll
        for (int ww = 1; ww < 8; ++ww) s += red[(ww * 4 + q) * 256 + r * 64 + ll];
        const int gi = 16 * rb + (ll >> 4) + 4 * r, gj = 16 * cb + (ll & 15);
        // this block's share of the squared distance, |x_i|^2 + |x_j|^2 - 2 x_i . x_j (symmetric bit for bit): what the
        // pairwise form leaves in `part` too, so k_svgd_kmat sums one kind of partial
        s = (nsum[gi] + nsum[gj]) - 2.0 * s;
        const int il = gi - g.row0, jl = gj - g.row0;
        if (il >= 0 && il < g.n_local) g.part[((long long)il * g.nblk + blk) * 64 + gj] = s;
        if (SYM && cb != rb && jl >= 0 && jl < g.n_local) g.part[((long long)jl * g.nblk + blk) * 64 + gi] = s;
      }
    }
  }
  PYZ_STAMP(4, 3);
}

// dist_only != 0: the squared distances of the row go to g.dmat and nothing else happens (first half of the
// median-heuristic path); with g.dmat set and dist_only == 0 the distances are read from there.
__global__ void __launch_bounds__(1024) k_svgd_kmat(SvgdTileArgs g, const int dist_only) {
  // sixteen waves per row: wave q sums one half of group q / 2 (all of a wave's loads in ONE round trip; four waves with
  // eight loads in flight each took eight dependent trips, 12 us for a 64-value row); groups gathered from other ranks
  // (g.groups) replace the sums over `part`
  __shared__ double sl[16][64];
  const int il = blockIdx.x, j = threadIdx.x & 63, q = threadIdx.x >> 6;
  double d;
  if (g.dmat && !dist_only) {
    if (q != 0) return;
    d = g.dmat[(g.row0 + il) * 64 + j];
  } else {
    if (g.groups) {
      if (q < PYZ_SVGD_GROUPS) sl[q][j] = g.groups[((long long)q * 64 + g.row0 + il) * 64 + j];
    } else {
      const int nb8 = (g.nblk + PYZ_SVGD_GROUPS - 1) / PYZ_SVGD_GROUPS, b_lo = (q >> 1) * nb8, b_hi = min(b_lo + nb8, g.nblk);
      sl[q][j] = pyz_svgd_half_sum(g.part + (long long)il * g.nblk * 64 + j, b_lo, b_hi, q & 1);
    }
    __syncthreads();
    if (q != 0) return;
    double sg[PYZ_SVGD_GROUPS];
#pragma unroll
    for (int u = 0; u < PYZ_SVGD_GROUPS; ++u) sg[u] = g.groups ? sl[u][j] : sl[2 * u][j] + sl[2 * u + 1][j];
    const double s0 = sg[0] + sg[1], s1 = sg[2] + sg[3], s2 = sg[4] + sg[5], s3 = sg[6] + sg[7];
    // (the Gram form's partials are |x_i|^2 + |x_j|^2 - 2 x_i . x_j over a block's elements, by cancellation: the diagonal
    //  is set to its exact value and the sum kept non-negative; the pairwise form's partials are sums of squares, its
    //  diagonal exact zeros -- both lines leave them unchanged)
    d = (j == g.row0 + il) ? 0.0 : fmax((s0 + s1) + (s2 + s3), 0.0);
    if (dist_only) {
      g.dmat[(g.row0 + il) * 64 + j] = j < g.M ? d : 0.0;
      return;
    }
  }
  const double gam = g.gamma_dev ? g.gamma_dev[0] : (double)g.gamma;
  const double k = j < g.M ? exp(-gam * d) : 0.0;
  g.kmat[il * 64 + j] = k;
  float ks = 0.0f;
  double k4[4] = {0.0, 0.0, 0.0, 0.0};   // four interleaved partial sums, combined in a fixed order (k_svgd_update_tile's
  for (int u = 0; u < g.M; ++u) {        //  arithmetic for x_i sum_j K_ij: the sum is row-constant, taken once here)
    const double ku = __shfl(k, u, 64);
    ks += (float)ku;
    k4[u & 3] += ku;
  }
  if (j == 0) {
    g.ksum[il] = ks;
    g.ksumd[il] = (k4[0] + k4[1]) + (k4[2] + k4[3]);
  }
}

// The median heuristic of SVGD.baseline__kernel (SVGD.py:165-181, dead code in the reference; opt-in here):
//   h = sqrt(0.5 median(sqdist) / log(M + 1)),  K = exp(-sqdist / (2 h^2))   =>   gamma = log(M + 1) / median(sqdist),
// the median over ALL M^2 entries of the squared-distance matrix (numpy.median: the mean of the two middle
// values for an even count).  One workgroup of 1024 threads sorts the M^2 <= 4096 values in LDS (bitonic
// network on the next power of two, padded with +inf) -- a fixed sequence of compare-exchanges: bitwise
// reproducible.  The repulsion term (-K X + X rowsum K) / h^2 is 2 gamma sum_j K_ij (x_i - x_j) with this gamma.
__global__ void __launch_bounds__(1024) k_svgd_median(SvgdTileArgs g) {
  __shared__ double v[4096];
  const int t = threadIdx.x, M = g.M, n = M * M;
  int n2 = 2;
  while (n2 < n) n2 *= 2;
  for (int e = t; e < n2; e += 1024) v[e] = e < n ? g.dmat[(e / M) * 64 + (e % M)] : __builtin_inf();
  __syncthreads();
  for (int k = 2; k <= n2; k *= 2)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int e = t; e < n2; e += 1024) {
        const int p = e ^ j;
        if (p > e) {
          const double a = v[e], b = v[p];
          const bool up = (e & k) == 0;
          if ((a > b) == up) {
            v[e] = b;
            v[p] = a;
          }
        }
      }
      __syncthreads();
    }
  if (t == 0) {
    const double med = (n & 1) ? v[n / 2] : 0.5 * (v[n / 2 - 1] + v[n / 2]);
    g.gamma_dev[0] = log((double)M + 1.0) / med;
  }
}

// phi_i and the legacy Adam step (the arithmetic of k_svgd_update) for every local row, one element per thread.
// The kernel values are wave-uniform: they come through the scalar cache (restrict-qualified kernel
// arguments, so the compiler may use s_load) and feed the float64 FMAs as SGPR operands; reading them
// from LDS cost one 512-byte broadcast per (row, j) and made the kernel LDS-bound (222 us).
__global__ void __launch_bounds__(256) k_svgd_update_tile(SvgdTileArgs g, const double *__restrict__ kmat,
                                                          const float *__restrict__ ksum,
                                                          const double *__restrict__ ksumd,
                                                          const double *__restrict__ gamma_dev) {
  if (g.loss_out && blockIdx.x == gridDim.x - 1 && threadIdx.x >= 192) {   // the last wave of the last (ragged) block:
    const int j = threadIdx.x & 63;                                         // k_svgd_loss's arithmetic, same order; the loads
    const float mine = j < g.n_local ? g.loss_in[j] / (float)g.M : 0.0f;   // in one round trip
    float s = 0.0f;
    for (int i = 0; i < g.n_local; ++i) s += __shfl(mine, i, 64);
    if (j == 0) g.loss_out[0] = s;
  }
  const long long d = (long long)blockIdx.x * 256 + threadIdx.x;
  if (d >= g.D) return;
  // x_j[d] of every particle: all 64 loads are issued before the first use (rows past M repeat the last one
  // and are zeroed; a load under `if (j < M)` would wait for its own round trip, 64 times in a row)
  float xf[64];
#pragma unroll
  for (int j = 0; j < 64; ++j) xf[j] = g.all[(long long)min(j, g.M - 1) * g.D + d];
  // (kept as float32 and widened at every use: 64 registers instead of 192 -- twice the waves per SIMD for a kernel that
  // waits on its five per-row operands)
#pragma unroll
  for (int j = 0; j < 64; ++j) xf[j] = j < g.M ? xf[j] : 0.0f;
  const double two_gamma = 2.0 * (gamma_dev ? gamma_dev[0] : (double)g.gamma);
  // (Requesting the five per-row operands of four rows together, or of the next row ahead of the FMAs, made the kernel
  // SLOWER: 134 and 149 us against 92 -- the row loop lives on its 64 kernel values arriving as SGPR operands, and more
  // live vector state pushes them out.)
  for (int il = 0; il < g.n_local; ++il) {
    const int i = g.row0 + il;
    const double *kr = kmat + il * 64;
    const float xi = g.all[(long long)i * g.D + d];
    // sum_j K_ij (x_i - x_j) = x_i sum_j K_ij - sum_j K_ij x_j: one float64 FMA per (i, j) instead of a
    // subtraction and an FMA (the kernel is float64-VALU-bound); float64 leaves ~1e-16 |x_i| sum_j K_ij of
    // cancellation error, far below float32 phi.  Four interleaved partial sums (j mod 4), combined in a fixed
    // order: a single chain of 64 dependent float64 FMAs would run at their latency, not their rate
    const double xid = (double)xi;
    double r4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < 64; ++j) r4[j & 3] += kr[j] * (double)xf[j];
    double rep = xid * ksumd[il] - ((r4[0] + r4[1]) + (r4[2] + r4[3]));
    rep *= two_gamma;
    const long long o = (long long)il * g.D + d;
    const float phi = (ksum[il] * g.grad[o] + (float)rep) / (float)g.M;
    float m = g.adam_m[o], v = g.adam_v[o];
    m = m + (phi - m) * (1.0f - 0.9f);
    v = v + (phi * phi - v) * (1.0f - 0.999f);
    g.adam_m[o] = m;
    g.adam_v[o] = v;
    // (the row's current value is the snapshot's: `particles` is only written, so a caller may alternate two buffers
    // instead of copying the matrix for every Jacobi step)
    g.particles[o] = xi - g.lr_t * m / (sqrtf(v) + 1e-7f);
  }
}

// phi of one element and the legacy Adam step on it (SVGD.py:66-67,113; Appendix A3), with every rounding spelled out:
// k_svgd_gs and k_svgd_gs_resident must give the same bits, and left to itself the compiler fuses multiply-adds
// differently in the two kernels.
__device__ __forceinline__ void pyz_svgd_gs_adam(const float ksum, const float gi, const double rep, const float gamma, const int M,
                                                 const float xi, const float lr_t, float &m, float &v, float &xn) {
#pragma clang fp contract(off)
  const float phi = (ksum * gi + (float)(rep * (2.0 * (double)gamma))) / (float)M;
  m = m + (phi - m) * (1.0f - 0.9f);
  v = v + (phi * phi - v) * (1.0f - 0.999f);
  xn = xi - lr_t * m / (sqrtf(v) + 1e-7f);
}

// ---------------------------------------------------------------- Gauss-Seidel sweep, one launch per particle
// The reference updates the particles one after the other, each against the rows already updated
// (SVGD.py:110-120).  With the per-row kernels above that is two launches per particle, each reading the whole
// (M, D) matrix: 2 x 40.7 MB at C5.  k_svgd_gs does one particle per launch and reads the matrix once:
//   a workgroup owns 256 * PYZ_GS_E consecutive elements of D and keeps x_j[d] of ALL particles in registers;
//   launch i  (a) sums the squared-distance partials of row i that launch i - 1 left (float64, block order),
//                 K_ij = exp(-gamma d_ij), sum_j K_ij;
//             (b) phi_i and the Adam step on its elements (the formulas and the order over j of k_svgd_update),
//                 stores x_i and patches its register copy;
//             (c) leaves the partials of row i + 1 against the matrix as it now stands.
//   The launch with i = -1 only does (c) for row 0.  Partials ping-pong between two buffers (a workgroup writes
//   its new partials while others still read the old ones).  Needs M <= 64 and D <= 256 * 256 * PYZ_GS_E
//   (every workgroup reads all partials: one round of workgroups); else the per-row kernels run.
#define PYZ_GS_E 3
#define PYZ_GS_AHEAD 32  // rows requested ahead of the arithmetic
#define PYZ_GS_PAD 257   // row stride (doubles) of the reduction scratch: conflict-free column sums
typedef float pyz_gs_vec __attribute__((ext_vector_type(PYZ_GS_E), aligned(4)));  // a thread's elements of one row
typedef double pyz_gs_d2 __attribute__((ext_vector_type(2)));

struct SvgdGsArgs {
  float *all;              // (M, D) particle matrix, updated in place
  float *adam_m, *adam_v;  // (M, D)
  const float *grad;       // (M, D)
  long long D;
  int M;
  int i;                   // particle of this launch; -1: only the distances of row 0
  float lr_t, gamma;
  const double *part_in;   // (nblk, 64) partials of row i
  double *part_out;        // (nblk, 64) partials of row i + 1
  int nblk;
};

static inline size_t pyz_svgd_gs_lds_bytes() { return sizeof(double) * (64 * PYZ_GS_PAD + 9 * 64); }

// LDS-only workgroup barrier: __syncthreads() also waits for the global loads in flight (one counter for loads
// and stores on this target), which would serialise the phases below

#ifdef PYZ_STAMPS
#define PYZ_GS_STAMP(slot) do { if (g.i == 32) PYZ_STAMP(3, slot); } while (0)   // one mid-sweep launch
#else
#define PYZ_GS_STAMP(slot)
#endif

// REV: the rows are requested and consumed in DESCENDING order.  Consecutive launches alternate the direction: a launch
// then starts on the rows its predecessor touched last, i.e. on what is still in the XCD's L2 (each XCD streams a
// 5.1 MB share of the matrix through a 4 MiB L2: always in the same direction nothing survives from launch to launch).
// The distance partials are per row and the float64 repulsion sum only changes its order of summation.
template <bool REV>
__global__ void __launch_bounds__(256) k_svgd_gs(SvgdGsArgs g) {
  extern __shared__ double gs_lds[];
  PYZ_GS_STAMP(0);
  double *red = gs_lds;                  // [64][PYZ_GS_PAD]
  double *ps8 = red + 64 * PYZ_GS_PAD;   // [8][64] slices of the partial sums
  double *ps4 = ps8;                     // [4][64] (second use: the block reduction at the end)
  double *sd = ps8 + 8 * 64;             // [64]
  const int tid = threadIdx.x, M = g.M, i = g.i;
  const long long D = g.D, base = (long long)blockIdx.x * (256 * PYZ_GS_E);
  // -- requests first, in the order they are consumed: the partials of row i (at most 256 blocks: 64 per thread,
  //    4 threads per row j), then the matrix; the K row is built while the matrix is still on its way
  const int pj = tid & 63, pq = tid >> 6;
  // (partials: a thread takes the row PAIR 2 pj2, 2 pj2 + 1 of every 8th block with 16-byte loads: 32 requests)
  const int pj2 = tid & 31, pq8 = tid >> 5;
  pyz_gs_d2 pv[32];
  if (i >= 0) {
#pragma unroll
    for (int u = 0; u < 32; ++u)
      pv[u] = *reinterpret_cast<const pyz_gs_d2 *>(g.part_in + (long long)min(pq8 + 8 * u, g.nblk - 1) * 64 + 2 * pj2);
  }
  // a thread owns PYZ_GS_E CONSECUTIVE elements: one 12-byte load per particle row
  long long e[PYZ_GS_E];
  bool in[PYZ_GS_E];
#pragma unroll
  for (int q = 0; q < PYZ_GS_E; ++q) {
    e[q] = base + PYZ_GS_E * tid + q;
    in[q] = e[q] < D;
    e[q] = in[q] ? e[q] : D - 1;
  }
  const bool whole = base + (long long)PYZ_GS_E * 64 * (pyz_wave_id() + 1) <= D;  // all elements of this WAVE exist (scalar)
  const int inext = i + 1;
  float xnext[PYZ_GS_E];
#pragma unroll
  for (int q = 0; q < PYZ_GS_E; ++q) xnext[q] = inext < M ? g.all[(long long)inext * D + e[q]] : 0.0f;
  float xi[PYZ_GS_E], gi[PYZ_GS_E], am[PYZ_GS_E], av[PYZ_GS_E];
  if (i >= 0) {
#pragma unroll
    for (int q = 0; q < PYZ_GS_E; ++q) {
      const long long o = (long long)i * D + e[q];
      xi[q] = g.all[o];
      gi[q] = g.grad[o];
      am[q] = g.adam_m[o];
      av[q] = g.adam_v[o];
    }
  }
  // -- the matrix, PYZ_GS_AHEAD rows ahead of the arithmetic on them: the partial squared distances of row i + 1
  //    against every row (differences and squares in float64: SVGD.py:198-201) are taken as the rows arrive and
  //    hide behind the loads.  (Row i is about to change: its term is recomputed below.)  The empty asm pins each
  //    row's arithmetic between the loads around it: left alone, the compiler sinks it to the LDS writes at the end.
  float x[64][PYZ_GS_E];
  double acc[64];
  auto load_row = [&](const int j) {
    if (whole) {
      const pyz_gs_vec v = *reinterpret_cast<const pyz_gs_vec *>(g.all + (long long)min(j, M - 1) * D + e[0]);
#pragma unroll
      for (int q = 0; q < PYZ_GS_E; ++q) x[j][q] = v[q];
    } else {
      const float *row = g.all + (long long)min(j, M - 1) * D;
#pragma unroll
      for (int q = 0; q < PYZ_GS_E; ++q) x[j][q] = row[e[q]];
    }
  };
#pragma unroll
  for (int j = 0; j < PYZ_GS_AHEAD; ++j) load_row(REV ? 63 - j : j);
  PYZ_GS_STAMP(1);
  // -- the K row of particle i, while the first rows are on their way (its partials were requested first)
  double kd[64];
  float ksum = 0.0f;
  double ksd = 0.0;   // sum_j K_ij in float64, j ascending: sum_j K_ij (x_i - x_j) = x_i ksd - sum_j K_ij x_j (one FMA per term)
  if (i >= 0) {
    {  // squared distances of row i: the partials of a row are summed by 8 threads (stride-8 slices), fixed order
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        const bool on = pq8 + 8 * u < g.nblk;
        s0 += on ? pv[u][0] : 0.0;
        s1 += on ? pv[u][1] : 0.0;
      }
      *reinterpret_cast<pyz_gs_d2 *>(ps8 + pq8 * 64 + 2 * pj2) = pyz_gs_d2{s0, s1};
    }
    PYZ_GS_STAMP(2);
    pyz_lds_barrier();
    if (tid < 64) {
      const double dsq = ((ps8[tid] + ps8[64 + tid]) + (ps8[128 + tid] + ps8[192 + tid])) +
                         ((ps8[256 + tid] + ps8[320 + tid]) + (ps8[384 + tid] + ps8[448 + tid]));
      sd[tid] = tid < M ? exp(-(double)g.gamma * dsq) : 0.0;
    }
    pyz_lds_barrier();
    // the K row into registers in one batch of LDS reads (a loop that reads sd[j] and branches on it pays the LDS
    // latency 64 times); rows past M hold 0
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      kd[j] = sd[j];
      ksum += (float)kd[j];   // (+0.0f past M: the sum over j < M, same order)
      ksd += kd[j];
    }
  }
  PYZ_GS_STAMP(3);
  double rep[PYZ_GS_E];
#pragma unroll
  for (int q = 0; q < PYZ_GS_E; ++q) rep[q] = 0.0;
#pragma unroll
  for (int jo = 0; jo < 64; ++jo) {
    constexpr int dummy_ = 0; (void)dummy_;
    const int j = REV ? 63 - jo : jo;   // (compile-time after unrolling)
    if (jo + PYZ_GS_AHEAD < 64) load_row(REV ? 63 - (jo + PYZ_GS_AHEAD) : jo + PYZ_GS_AHEAD);
    double a = 0.0;
#pragma unroll
    for (int q = 0; q < PYZ_GS_E; ++q) {
      const double df = in[q] ? (double)xnext[q] - (double)x[j][q] : 0.0;
      a = fma(df, df, a);
    }
    asm volatile("" : "+v"(a)::"memory");
    acc[j] = a;
    if (i >= 0) {  // repulsion term of this row, in float64 like the kernel values (see below)
#pragma unroll
      for (int q = 0; q < PYZ_GS_E; ++q) {
        rep[q] = fma(kd[j], (double)x[j][q], rep[q]);
        asm volatile("" : "+v"(rep[q]));
      }
    }
  }
  PYZ_GS_STAMP(4);
  double acc_i = 0.0;
  if (i >= 0) {
    // (the repulsion sum was taken in float64 as the rows arrived -- the reference differentiates the float64
    // kernel sum and casts the result once, SVGD.py:59-63 -- with k_svgd_update's arithmetic: where the 64 terms
    // K_ij (x_i - x_j) cancel, a float32 sum loses the SIGN of phi, and Adam's first steps move by lr sign(phi);
    // row i -- difference 0 -- and underflowed rows add exactly 0)
#pragma unroll
    for (int q = 0; q < PYZ_GS_E; ++q) {
      float m = am[q], v = av[q], xn;
      pyz_svgd_gs_adam(ksum, gi[q], fma((double)xi[q], ksd, -rep[q]), g.gamma, M, xi[q], g.lr_t, m, v, xn);
      if (in[q]) {
        const long long o = (long long)i * D + e[q];
        g.adam_m[o] = m;
        g.adam_v[o] = v;
        g.all[o] = xn;
        const double df = (double)xnext[q] - (double)xn;
        acc_i = fma(df, df, acc_i);
      }
    }
  }
  PYZ_GS_STAMP(5);
  if (inext >= M) return;
#pragma unroll
  for (int j = 0; j < 64; ++j) red[j * PYZ_GS_PAD + tid] = acc[j];
  if (i >= 0) red[i * PYZ_GS_PAD + tid] = acc_i;   // the updated row i (same thread, after its first write)
  PYZ_GS_STAMP(6);
  pyz_lds_barrier();  // (also orders the ps4 reads of the K row before the writes below)
  {
    const double *rp = red + pj * PYZ_GS_PAD + 64 * pq;
    double s = 0.0;
    for (int t = 0; t < 64; ++t) s += rp[t];
    ps4[pq * 64 + pj] = s;
  }
  pyz_lds_barrier();
  if (tid < 64) g.part_out[(long long)blockIdx.x * 64 + tid] = tid < M ? (ps4[tid] + ps4[64 + tid]) + (ps4[128 + tid] + ps4[192 + tid]) : 0.0;
  PYZ_GS_STAMP(7);
}

// ---------------------------------------------------------------- Gauss-Seidel sweep, ONE launch for all particles
// k_svgd_gs reads the whole matrix once per particle (64 x 40.7 MB at C5, 14.4 us per launch).  Here the workgroups stay
// resident for the sweep: each loads its 768 elements of all rows into registers ONCE and keeps them current (row i is
// patched when it is updated); per particle the workgroups exchange 64 partial squared distances through memory as
// data-tagged granules (k_hmc_resident's hand-off: one aligned 8-byte write-through store {tag, half of a double}; the
// consumer re-reads until the tags are the step's), in two hops:
//   every workgroup publishes its block's partials of row i + 1 (after updating row i);
//   the reducer of column j (one of cdiv(M, 4) extra workgroups that own no elements, four columns each, sixteen lanes per
//   column so that every lane has ONE batch of loads: a reducer wave that also carried a slice's arithmetic held every step
//   back by its own, and eight lanes per column needed two batches in turn -- a second round trip on the chain every workgroup
//   waits on, 2 200 cycles per particle) sums the nblk partials of d_{i+1,j} in k_svgd_gs's
//   order, takes K_{i+1,j} = exp(-gamma d) and publishes it; every workgroup's first wave polls the 64 kernel values.
// Everything else is k_svgd_gs<false>'s arithmetic in its order, so both forms give the same bits (with the per-launch
// form's rows in ascending order: PYZ_SVGD_GS_ZIGZAG=0).  The partials of row i + 1 against the rows that do not change
// are taken BEFORE the wait for K_i.  Granule rows ping-pong by particle parity (a workgroup that holds K_i knows every
// reducer has read the partials of row i; one that publishes partials of row i + 2 knows everyone has read K_i).
// The grid must be resident at once (the launcher checks); a wave that polls `spin_limit` times without progress
// gives up: *fail is set (the step's loss becomes NaN, k_svgd_loss), every workgroup returns -- the grid always drains.
struct SvgdGsResArgs {
  float *all;              // (M, D) particle matrix, updated in place
  float *adam_m, *adam_v;  // (M, D)
  const float *grad;       // (M, D)
  long long D;
  int M;
  float lr_t, gamma;
  int nblk;
  unsigned *epoch;             // [1] tags <= *epoch are stale; advanced by M per sweep (workgroup 0)
  unsigned long long *pgran;   // (2, nblk, 64, 2) granules {tag, low half} {tag, high half} of the partials
  unsigned long long *kgran;   // (2, 64, 2) granules of the kernel row
  unsigned long long *cgran;   // (2, 256, 2) granules of the blocks' partials of the one critical distance (k_svgd_gs_resident2)
  int *fail;
  int spin_limit;
};

static inline size_t pyz_svgd_gs_res_lds_bytes() { return pyz_svgd_gs_lds_bytes() + 64; }   // + the row sum of K, the give-up flag

typedef __attribute__((address_space(1))) unsigned long long pyz_gs_gu64;
__device__ __forceinline__ void pyz_gs_store_f64(unsigned long long *g, const unsigned tag, const double v) {
  const unsigned long long bits = __builtin_bit_cast(unsigned long long, v), t = (unsigned long long)tag << 32;
  __hip_atomic_store((pyz_gs_gu64 *)g, t | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store((pyz_gs_gu64 *)(g + 1), t | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long pyz_gs_load(const unsigned long long *g) {
  return __hip_atomic_load((pyz_gs_gu64 *)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double pyz_gs_join(const unsigned long long lo, const unsigned long long hi) {
  return __builtin_bit_cast(double, ((hi & 0xffffffffull) << 32) | (lo & 0xffffffffull));
}

#ifndef PYZ_GS_POLL_SLEEP
#define PYZ_GS_POLL_SLEEP 2
#endif
__global__ void __launch_bounds__(256) k_svgd_gs_resident(SvgdGsResArgs g) {
  extern __shared__ double gs_lds[];
  double *red = gs_lds;                  // [64][PYZ_GS_PAD]
  double *ps4 = red + 64 * PYZ_GS_PAD;   // [4][64]
  double *sd = ps4 + 8 * 64;             // [64]
  int *gave_up = reinterpret_cast<int *>(sd + 66);
  const int tid = threadIdx.x, M = g.M, b = blockIdx.x, w = pyz_wave_id(), lane = tid & 63;
  const long long D = g.D, base = (long long)b * (256 * PYZ_GS_E);
  const int pj = tid & 63, pq = tid >> 6;
  if (tid == 0) *gave_up = 0;
  const unsigned epoch0 = *g.epoch;   // (left by the previous sweep's launch)
  if (b >= g.nblk) {
    // ---- a reducer workgroup (no elements of its own): columns 4 r .. 4 r + 3 of every row's squared distances, one wave
    if (w != 0) return;
    // sixteen lanes per column: slice sl = the blocks sl, sl + 8, ... (k_svgd_gs's eight stride-8 slices), its first sixteen
    // blocks on lane `half` = 0 and the rest on half = 1 -- every lane has ONE batch of loads (one round trip per particle on
    // the chain every workgroup waits on; eight lanes per column needed two batches in turn); the sum stays in block order:
    // the second half continues from the first half's sum
    const int r = b - g.nblk, c = lane >> 4, sl = lane & 7, half = (lane >> 3) & 1, j = 4 * r + c;
    for (int inext = 0; inext < M; ++inext) {
      const unsigned tagn = epoch0 + 1u + (unsigned)inext;
      const unsigned long long *pp = g.pgran + ((long long)(inext & 1) * g.nblk * 64 + min(j, M - 1)) * 2;
      unsigned long long lo[16], hi[16];
      bool lost = false;
      for (int spins = 0;;) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const unsigned long long *p = pp + (long long)min(sl + 8 * (16 * half + u), g.nblk - 1) * 128;
          lo[u] = pyz_gs_load(p);
          hi[u] = pyz_gs_load(p + 1);
        }
        unsigned bad = 0;
#pragma unroll
        for (int u = 0; u < 16; ++u) bad |= ((unsigned)(lo[u] >> 32) ^ tagn) | ((unsigned)(hi[u] >> 32) ^ tagn);
        if (__all(bad == 0)) break;
        if (++spins > g.spin_limit) {   // (wave-uniform)
          lost = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (lost) {   // the compute workgroups never showed up: they give up on their own polls
        if (lane == 0) *g.fail = 1;
        return;
      }
      double sum = 0.0;
#pragma unroll
      for (int u = 0; u < 16; ++u) sum += sl + 8 * u < g.nblk ? pyz_gs_join(lo[u], hi[u]) : 0.0;   // (first half's blocks; used by half 0)
      const double first = __shfl(sum, lane & ~8, 64);   // the first half's sum of this slice
      if (half) {
        sum = first;
#pragma unroll
        for (int u = 0; u < 16; ++u) sum += sl + 8 * (16 + u) < g.nblk ? pyz_gs_join(lo[u], hi[u]) : 0.0;
      }
      const int l0 = (lane & ~15) + 8;   // the lanes that hold the slices' whole sums
      const double s0 = __shfl(sum, l0, 64), s1 = __shfl(sum, l0 + 1, 64), s2 = __shfl(sum, l0 + 2, 64), s3 = __shfl(sum, l0 + 3, 64);
      const double s4 = __shfl(sum, l0 + 4, 64), s5 = __shfl(sum, l0 + 5, 64), s6 = __shfl(sum, l0 + 6, 64), s7 = __shfl(sum, l0 + 7, 64);
      if ((lane & 15) == 0 && j < M) {
        const double dsq = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
        pyz_gs_store_f64(g.kgran + ((long long)(inext & 1) * 64 + j) * 2, tagn, exp(-(double)g.gamma * dsq));
      }
    }
    return;
  }
  long long e[PYZ_GS_E];
  bool in[PYZ_GS_E];
#pragma unroll
  for (int q = 0; q < PYZ_GS_E; ++q) {
    e[q] = base + PYZ_GS_E * tid + q;
    in[q] = e[q] < D;
    e[q] = in[q] ? e[q] : D - 1;
  }
  const bool whole = base + (long long)PYZ_GS_E * 64 * (w + 1) <= D;  // all elements of this WAVE exist (scalar)
  // the workgroup's elements of every row, once
  float x[64][PYZ_GS_E];
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    if (whole) {
      const pyz_gs_vec v = *reinterpret_cast<const pyz_gs_vec *>(g.all + (long long)min(j, M - 1) * D + e[0]);
#pragma unroll
      for (int q = 0; q < PYZ_GS_E; ++q) x[j][q] = v[q];
    } else {
      const float *row = g.all + (long long)min(j, M - 1) * D;
#pragma unroll
      for (int q = 0; q < PYZ_GS_E; ++q) x[j][q] = row[e[q]];
    }
  }
  float xnext[PYZ_GS_E], xi[PYZ_GS_E], gi[PYZ_GS_E], am[PYZ_GS_E], av[PYZ_GS_E];
#pragma unroll
  for (int q = 0; q < PYZ_GS_E; ++q) {
    xnext[q] = x[0][q];
    xi[q] = gi[q] = am[q] = av[q] = 0.0f;
  }
  __syncthreads();   // gave_up is initialised
#ifdef PYZ_STAMPS
  unsigned long long lap[16] = {0};
  PYZ_LAP(lap, 8);
  lap[8] = 0;
#else
  unsigned long long *lap = nullptr;
#endif
  for (int i = -1; i < M; ++i) {
    const int inext = i + 1;
    PYZ_LAP(lap, 5);
    // -- partial squared distances of row i + 1 against every row as it stands (row i is about to change: its term is
    //    recomputed below); float64 differences and squares (SVGD.py:198-201)
    if (inext < M) {
#pragma unroll
      for (int j = 0; j < 64; ++j) {
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < PYZ_GS_E; ++q) {
          const double df = in[q] ? (double)xnext[q] - (double)x[j][q] : 0.0;
          a = fma(df, df, a);
        }
        red[j * PYZ_GS_PAD + tid] = a;
      }
    }
    PYZ_LAP(lap, 0);
    if (i >= 0) {
      // -- the kernel row of particle i: the first wave polls the reducers' granules
      const unsigned tag = epoch0 + 1u + (unsigned)i;
      if (w == 0) {
        const unsigned long long *kg = g.kgran + ((long long)(i & 1) * 64 + lane) * 2;
        unsigned long long lo = 0, hi = 0;
        for (int spins = 0;;) {
          lo = pyz_gs_load(kg);
          hi = pyz_gs_load(kg + 1);
          const bool ok = lane >= M || ((unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag);
          if (__all(ok)) break;
          if (++spins > g.spin_limit) {   // (wave-uniform)
            *gave_up = 1;
            *g.fail = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(PYZ_GS_POLL_SLEEP);
        }
        const double kv = lane < M ? pyz_gs_join(lo, hi) : 0.0;
        sd[lane] = kv;
        // sum_j (float) K_ij, j ascending (+0.0f past M), once per workgroup: lane j's value as a SCALAR operand (v_readlane
        // with a constant lane; a shuffle by a loop counter is an LDS permute per term on the path every workgroup waits on)
        float ks = 0.0f;
        const int kfb = __builtin_bit_cast(int, (float)kv);
#pragma unroll
        for (int j = 0; j < 64; ++j) ks += __builtin_bit_cast(float, __builtin_amdgcn_readlane(kfb, j));
        if (lane == 0) sd[64] = (double)ks;
      }
      pyz_lds_barrier();
      if (*gave_up) {   // (uniform) the others never showed up: not resident together, or one of them gave up
        if (b == 0 && tid == 0) *g.epoch = epoch0 + (unsigned)M;
        return;
      }
      PYZ_LAP(lap, 1);
      const float ksum = (float)sd[64];
      double ksd = 0.0;   // sum_j K_ij in float64, j ascending (every thread: 64 adds beside its FMAs)
      double rep[PYZ_GS_E];
#pragma unroll
      for (int q = 0; q < PYZ_GS_E; ++q) rep[q] = 0.0;
#pragma unroll
      for (int j = 0; j < 64; ++j) {
        const double kdj = sd[j];
        ksd += kdj;
#pragma unroll
        for (int q = 0; q < PYZ_GS_E; ++q) rep[q] = fma(kdj, (double)x[j][q], rep[q]);
      }
      double acc_i = 0.0;
      float xn[PYZ_GS_E];
#pragma unroll
      for (int q = 0; q < PYZ_GS_E; ++q) {
        float m = am[q], v = av[q];
        pyz_svgd_gs_adam(ksum, gi[q], fma((double)xi[q], ksd, -rep[q]), g.gamma, M, xi[q], g.lr_t, m, v, xn[q]);
        if (in[q]) {
          const long long o = (long long)i * D + e[q];
          g.adam_m[o] = m;
          g.adam_v[o] = v;
          g.all[o] = xn[q];
          const double df = (double)xnext[q] - (double)xn[q];
          acc_i = fma(df, df, acc_i);
        }
      }
      if (inext < M) red[i * PYZ_GS_PAD + tid] = acc_i;   // the updated row i (same thread, after its first write)
#pragma unroll
      for (int q = 0; q < PYZ_GS_E; ++q) xi[q] = xn[q];   // (kept for the register copy, patched behind the publish)
    }
    if (inext >= M) break;
    PYZ_LAP(lap, 2);
    pyz_lds_barrier();
    {
      const double *rp = red + pj * PYZ_GS_PAD + 64 * pq;
      double s = 0.0;
#pragma unroll
      for (int t0 = 0; t0 < 64; t0 += 16) {   // (the reads of a batch in flight together; the sum stays in order)
        double v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = rp[t0 + t];
#pragma unroll
        for (int t = 0; t < 16; ++t) s += v[t];
      }
      ps4[pq * 64 + pj] = s;
    }
    pyz_lds_barrier();
    PYZ_LAP(lap, 3);
    const unsigned tagn = epoch0 + 1u + (unsigned)inext;
    if (w == 0) {   // this block's partials of row i + 1
      const double tot = lane < M ? (ps4[lane] + ps4[64 + lane]) + (ps4[128 + lane] + ps4[192 + lane]) : 0.0;
      pyz_gs_store_f64(g.pgran + (((long long)(inext & 1) * g.nblk + b) * 64 + lane) * 2, tagn, tot);
    }
    // the register copy of row i: i is uniform, so this is a scalar jump into one of 64 three-move cases (192 selects --
    // with part of x living in accumulation registers, three instructions each -- took 4 000 cycles per particle)
#define PYZ_GS_PATCH(J) case J: _Pragma("unroll") for (int q = 0; q < PYZ_GS_E; ++q) x[J][q] = xi[q]; break;
#define PYZ_GS_PATCH8(J) PYZ_GS_PATCH(J) PYZ_GS_PATCH(J + 1) PYZ_GS_PATCH(J + 2) PYZ_GS_PATCH(J + 3) PYZ_GS_PATCH(J + 4) PYZ_GS_PATCH(J + 5) PYZ_GS_PATCH(J + 6) PYZ_GS_PATCH(J + 7)
    switch (i) {
      PYZ_GS_PATCH8(0) PYZ_GS_PATCH8(8) PYZ_GS_PATCH8(16) PYZ_GS_PATCH8(24) PYZ_GS_PATCH8(32) PYZ_GS_PATCH8(40) PYZ_GS_PATCH8(48) PYZ_GS_PATCH8(56)
      default: break;
    }
#undef PYZ_GS_PATCH8
#undef PYZ_GS_PATCH
    PYZ_LAP(lap, 4);
    // -- the next particle's operands (rows >= i + 1 are still the values the sweep started with)
#pragma unroll
    for (int q = 0; q < PYZ_GS_E; ++q) {
      const long long o = (long long)inext * D + e[q];
      xi[q] = xnext[q];
      gi[q] = g.grad[o];
      am[q] = g.adam_m[o];
      av[q] = g.adam_v[o];
      xnext[q] = inext + 1 < M ? g.all[(long long)(inext + 1) * D + e[q]] : 0.0f;
    }
  }
  if (b == 0 && tid == 0) *g.epoch = epoch0 + (unsigned)M;
#ifdef PYZ_STAMPS
  if (lane == 0 && b < PYZ_STAMP_BLOCKS)
    for (int k = 0; k < 8; ++k) pyz_dbg_buf[3][b][w][k][0] = lap[k];
#endif
}

// ---------------------------------------------------------------- resident Gauss-Seidel sweep, distances one step early
// k_svgd_gs_resident's step i is a chain: K_i -> update of row i -> partials of row i + 1 -> reducers -> K_{i+1}: two cross-XCD
// hops plus the reducer wave per particle.  But only ONE of the 64 squared distances of row i + 1 depends on the update of row i:
// d(i + 1, i).  Here the partials of row i + 2 against every row are taken and sent through the reducers during step i (the
// column of row i + 1, not yet updated, is ignored by its reducer), and the one critical distance goes in ONE hop: after updating
// row i every workgroup publishes its block's partial of d(i + 1, i) (`cgran`), and every workgroup's first wave sums the nblk
// partials itself -- in the reducers' order, eight stride-8 slices in block order -- and takes K_{i+1,i} = exp(-gamma d).  The
// arithmetic is k_svgd_gs_resident's term by term (same partials, same orders of summation), so both give the same bits.
// Slots by row parity as before: a workgroup that publishes partials of row i + 2 holds K_i, so every reducer is done with row
// i's slot; the reducers publish K_{i+2} after every workgroup's partials of row i + 2, which each sent after reading K_i; a
// workgroup that publishes its critical partial for row i + 1 has summed everyone's for row i, which each sent after summing
// those for row i - 1.
__global__ void __launch_bounds__(256) k_svgd_gs_resident2(SvgdGsResArgs g) {
  extern __shared__ double gs_lds[];
  double *red = gs_lds;                  // [64][PYZ_GS_PAD]
  double *ps4 = red + 64 * PYZ_GS_PAD;   // [4][64]
  double *cv = ps4 + 4 * 64;             // [8][32] the critical distance's partials, slice-major (first wave only)
  double *sd = ps4 + 8 * 64;             // [64]
  int *gave_up = reinterpret_cast<int *>(sd + 66);
  const int tid = threadIdx.x, M = g.M, b = blockIdx.x, w = pyz_wave_id(), lane = tid & 63;
  const long long D = g.D, base = (long long)b * (256 * PYZ_GS_E);
  const int pj = tid & 63, pq = tid >> 6;
  if (tid == 0) *gave_up = 0;
  const unsigned epoch0 = *g.epoch;   // (left by the previous sweep's launch)
  if (b >= g.nblk) {
    // ---- a reducer workgroup: k_svgd_gs_resident's, except that row r's column r - 1 is not published
    if (w != 0) return;
    const int r = b - g.nblk, c = lane >> 3, sl = lane & 7, j = 8 * r + c;
    for (int inext = 0; inext < M; ++inext) {
      const unsigned tagn = epoch0 + 1u + (unsigned)inext;
      const unsigned long long *pp = g.pgran + ((long long)(inext & 1) * g.nblk * 64 + min(j, M - 1)) * 2;
      double sum = 0.0;
      bool lost = false;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        unsigned long long lo[16], hi[16];
        for (int spins = 0;;) {
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const unsigned long long *p = pp + (long long)min(sl + 8 * (16 * h + u), g.nblk - 1) * 128;
            lo[u] = pyz_gs_load(p);
            hi[u] = pyz_gs_load(p + 1);
          }
          unsigned bad = 0;
#pragma unroll
          for (int u = 0; u < 16; ++u) bad |= ((unsigned)(lo[u] >> 32) ^ tagn) | ((unsigned)(hi[u] >> 32) ^ tagn);
          if (__all(bad == 0)) break;
          if (++spins > g.spin_limit) {   // (wave-uniform)
            lost = true;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (lost) break;
#pragma unroll
        for (int u = 0; u < 16; ++u) sum += sl + 8 * (16 * h + u) < g.nblk ? pyz_gs_join(lo[u], hi[u]) : 0.0;
      }
      if (lost) {   // the compute workgroups never showed up: they give up on their own polls
        if (lane == 0) *g.fail = 1;
        return;
      }
      const int l0 = lane & ~7;
      const double s0 = __shfl(sum, l0, 64), s1 = __shfl(sum, l0 + 1, 64), s2 = __shfl(sum, l0 + 2, 64), s3 = __shfl(sum, l0 + 3, 64);
      const double s4 = __shfl(sum, l0 + 4, 64), s5 = __shfl(sum, l0 + 5, 64), s6 = __shfl(sum, l0 + 6, 64), s7 = __shfl(sum, l0 + 7, 64);
      if (sl == 0 && j < M && j != inext - 1) {
        const double dsq = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
        pyz_gs_store_f64(g.kgran + ((long long)(inext & 1) * 64 + j) * 2, tagn, exp(-(double)g.gamma * dsq));
      }
    }
    return;
  }
  long long e[PYZ_GS_E];
  bool in[PYZ_GS_E];
#pragma unroll
  for (int q = 0; q < PYZ_GS_E; ++q) {
    e[q] = base + PYZ_GS_E * tid + q;
    in[q] = e[q] < D;
    e[q] = in[q] ? e[q] : D - 1;
  }
  const bool whole = base + (long long)PYZ_GS_E * 64 * (w + 1) <= D;  // all elements of this WAVE exist (scalar)
  float x[64][PYZ_GS_E];
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    if (whole) {
      const pyz_gs_vec v = *reinterpret_cast<const pyz_gs_vec *>(g.all + (long long)min(j, M - 1) * D + e[0]);
#pragma unroll
      for (int q = 0; q < PYZ_GS_E; ++q) x[j][q] = v[q];
    } else {
      const float *row = g.all + (long long)min(j, M - 1) * D;
#pragma unroll
      for (int q = 0; q < PYZ_GS_E; ++q) x[j][q] = row[e[q]];
    }
  }
  __syncthreads();   // gave_up is initialised
#ifdef PYZ_STAMPS
  unsigned long long lap[16] = {0};
  PYZ_LAP(lap, 8);
  lap[8] = 0;
#else
  unsigned long long *lap = nullptr;
  (void)lap;
#endif
  // the first wave's granules of the NEXT step's kernel row, requested behind the distance arithmetic of send_row (a coherent
  // load is a round trip across the XCDs: asked for at the top of the step it is paid in full even when the values are there)
  unsigned long long plo = 0, phi = 0, pclo[4] = {0, 0, 0, 0}, pchi[4] = {0, 0, 0, 0};
  bool pre = false;
  // partial squared distances of a row (values xr) against every row as it stands -> this block's 64 sums -> the reducers
  // (float64 differences and squares, SVGD.py:198-201; k_svgd_gs's sums in its order)
  auto send_row = [&](const float (&xr)[PYZ_GS_E], const int row) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      double a = 0.0;
#pragma unroll
      for (int q = 0; q < PYZ_GS_E; ++q) {
        const double df = in[q] ? (double)xr[q] - (double)x[j][q] : 0.0;
        a = fma(df, df, a);
      }
      red[j * PYZ_GS_PAD + tid] = a;
    }
    PYZ_LAP(lap, 4);
    pre = row >= 2;   // (step row - 1 >= 1 comes next: it has a critical distance, published by every block a pass ago)
    if (w == 0 && pre) {
      const int nx = row - 1;
      const unsigned long long *kg = g.kgran + ((long long)(nx & 1) * 64 + lane) * 2;
      const unsigned long long *cg = g.cgran + (long long)(nx & 1) * 512;
      plo = pyz_gs_load(kg);
      phi = pyz_gs_load(kg + 1);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int blk = min((lane & 7) + 8 * ((lane >> 3) + 8 * u), g.nblk - 1);
        pclo[u] = pyz_gs_load(cg + 2 * blk);
        pchi[u] = pyz_gs_load(cg + 2 * blk + 1);
      }
    }
    pyz_lds_barrier();
    {
      const double *rp = red + pj * PYZ_GS_PAD + 64 * pq;
      double s = 0.0;
#pragma unroll
      for (int t0 = 0; t0 < 64; t0 += 16) {   // (the reads of a batch in flight together; the sum stays in order)
        double v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = rp[t0 + t];
#pragma unroll
        for (int t = 0; t < 16; ++t) s += v[t];
      }
      ps4[pq * 64 + pj] = s;
    }
    pyz_lds_barrier();
    if (w == 0) {
      const double tot = lane < M ? (ps4[lane] + ps4[64 + lane]) + (ps4[128 + lane] + ps4[192 + lane]) : 0.0;
      pyz_gs_store_f64(g.pgran + (((long long)(row & 1) * g.nblk + b) * 64 + lane) * 2, epoch0 + 1u + (unsigned)row, tot);
    }
    pyz_lds_barrier();   // (ps4 and red are free again)
    PYZ_LAP(lap, 5);
  };
  // steps -2 and -1 only send rows 0 and 1 (row 1's column 0 waits for the update of row 0: the reducers leave it out)
  float xi[PYZ_GS_E], xnext[PYZ_GS_E], xnn[PYZ_GS_E], gi[PYZ_GS_E], am[PYZ_GS_E], av[PYZ_GS_E];
#pragma unroll
  for (int q = 0; q < PYZ_GS_E; ++q) {
    xnn[q] = x[0][q];
    xi[q] = xnext[q] = gi[q] = am[q] = av[q] = 0.0f;
  }
  for (int i = -2; i < M; ++i) {
    const int inext = i + 1;
    float xn[PYZ_GS_E] = {0.0f, 0.0f, 0.0f};
    if (i >= 0) {
    // -- the kernel row of particle i: the reducers' granules, and the critical distance d(i, i - 1) from every block's partial
    const unsigned tag = epoch0 + 1u + (unsigned)i;
    if (w == 0) {
      const unsigned long long *kg = g.kgran + ((long long)(i & 1) * 64 + lane) * 2;
      const unsigned long long *cg = g.cgran + (long long)(i & 1) * 512;
      const bool need_k = lane < M && lane != i - 1;
      const int sl = lane & 7, c = lane >> 3;   // lane -> blocks sl + 8 (c + 8 u), u < 4
      unsigned long long lo = plo, hi = phi, clo[4] = {pclo[0], pclo[1], pclo[2], pclo[3]}, chi[4] = {pchi[0], pchi[1], pchi[2], pchi[3]};
      bool have = pre;   // (uniform) the first look is at what was requested a pass ago
      for (int spins = 0;;) {
        if (!have) {
          lo = pyz_gs_load(kg);
          hi = pyz_gs_load(kg + 1);
          if (i >= 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int blk = min(sl + 8 * (c + 8 * u), g.nblk - 1);
              clo[u] = pyz_gs_load(cg + 2 * blk);
              chi[u] = pyz_gs_load(cg + 2 * blk + 1);
            }
          }
        }
        have = false;
        bool ok = !need_k || ((unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag);
        if (i >= 1) {
#pragma unroll
          for (int u = 0; u < 4; ++u) ok = ok && (unsigned)(clo[u] >> 32) == tag && (unsigned)(chi[u] >> 32) == tag;
        }
        if (__all(ok)) break;
        if (++spins > g.spin_limit) {   // (wave-uniform)
          *gave_up = 1;
          *g.fail = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(PYZ_GS_POLL_SLEEP);
      }
      double kv = need_k ? pyz_gs_join(lo, hi) : 0.0;
      if (i >= 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = c + 8 * u;
          cv[sl * 32 + k] = sl + 8 * k < g.nblk ? pyz_gs_join(clo[u], chi[u]) : 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double s = 0.0;   // slice sl in block order (every lane: lanes 8 c + sl repeat lane sl)
#pragma unroll
        for (int k0 = 0; k0 < 32; k0 += 16) {
          double v[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) v[k] = cv[sl * 32 + k0 + k];
#pragma unroll
          for (int k = 0; k < 16; ++k) s += v[k];
        }
        const double s0 = __shfl(s, 0, 64), s1 = __shfl(s, 1, 64), s2 = __shfl(s, 2, 64), s3 = __shfl(s, 3, 64);
        const double s4 = __shfl(s, 4, 64), s5 = __shfl(s, 5, 64), s6 = __shfl(s, 6, 64), s7 = __shfl(s, 7, 64);
        const double dsq = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
        if (lane == i - 1) kv = exp(-(double)g.gamma * dsq);
      }
      sd[lane] = kv;
      float ks = 0.0f;
      const int kfb = __builtin_bit_cast(int, (float)kv);
#pragma unroll
      for (int j = 0; j < 64; ++j) ks += __builtin_bit_cast(float, __builtin_amdgcn_readlane(kfb, j));
      if (lane == 0) sd[64] = (double)ks;
    }
    pyz_lds_barrier();
    if (*gave_up) {   // (uniform) the others never showed up: not resident together, or one of them gave up
      if (b == 0 && tid == 0) *g.epoch = epoch0 + (unsigned)M;
      return;
    }
    PYZ_LAP(lap, 0);
    const float ksum = (float)sd[64];
    double ksd = 0.0;   // sum_j K_ij in float64, j ascending
    double rep[PYZ_GS_E];
#pragma unroll
    for (int q = 0; q < PYZ_GS_E; ++q) rep[q] = 0.0;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      const double kdj = sd[j];
      ksd += kdj;
#pragma unroll
      for (int q = 0; q < PYZ_GS_E; ++q) rep[q] = fma(kdj, (double)x[j][q], rep[q]);
    }
    double acc_i = 0.0;
#pragma unroll
    for (int q = 0; q < PYZ_GS_E; ++q) {
      float m = am[q], v = av[q];
      pyz_svgd_gs_adam(ksum, gi[q], fma((double)xi[q], ksd, -rep[q]), g.gamma, M, xi[q], g.lr_t, m, v, xn[q]);
      if (in[q]) {
        const long long o = (long long)i * D + e[q];
        g.adam_m[o] = m;
        g.adam_v[o] = v;
        g.all[o] = xn[q];
        const double df = (double)xnext[q] - (double)xn[q];
        acc_i = fma(df, df, acc_i);
      }
    }
    if (inext >= M) break;
    PYZ_LAP(lap, 1);
    // -- the critical partial: this block's sum of d(i + 1, i), the block reduction's column i alone
    red[i * PYZ_GS_PAD + tid] = acc_i;
    pyz_lds_barrier();
    if (pj == i) {
      const double *rp = red + i * PYZ_GS_PAD + 64 * pq;
      double s = 0.0;
#pragma unroll
      for (int t0 = 0; t0 < 64; t0 += 16) {
        double v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = rp[t0 + t];
#pragma unroll
        for (int t = 0; t < 16; ++t) s += v[t];
      }
      ps4[pq * 64 + i] = s;
    }
    pyz_lds_barrier();
    if (tid == 0) {
      const double tot = (ps4[i] + ps4[64 + i]) + (ps4[128 + i] + ps4[192 + i]);
      pyz_gs_store_f64(g.cgran + (long long)(inext & 1) * 512 + 2 * b, epoch0 + 1u + (unsigned)inext, tot);
    }
    PYZ_LAP(lap, 2);
    }   // (i >= 0)
    // -- the next particle's operands (rows >= i + 1 are still the values the sweep started with): back behind the next wait
    if (inext >= 0) {
#pragma unroll
      for (int q = 0; q < PYZ_GS_E; ++q) {
        const long long o = (long long)inext * D + e[q];
        gi[q] = g.grad[o];
        am[q] = g.adam_m[o];
        av[q] = g.adam_v[o];
      }
    }
    // the register copy of row i (uniform i: a scalar jump into one of 64 three-move cases)
#define PYZ_GS_PATCH(J) case J: _Pragma("unroll") for (int q = 0; q < PYZ_GS_E; ++q) x[J][q] = xn[q]; break;
#define PYZ_GS_PATCH8(J) PYZ_GS_PATCH(J) PYZ_GS_PATCH(J + 1) PYZ_GS_PATCH(J + 2) PYZ_GS_PATCH(J + 3) PYZ_GS_PATCH(J + 4) PYZ_GS_PATCH(J + 5) PYZ_GS_PATCH(J + 6) PYZ_GS_PATCH(J + 7)
    switch (i) {
      PYZ_GS_PATCH8(0) PYZ_GS_PATCH8(8) PYZ_GS_PATCH8(16) PYZ_GS_PATCH8(24) PYZ_GS_PATCH8(32) PYZ_GS_PATCH8(40) PYZ_GS_PATCH8(48) PYZ_GS_PATCH8(56)
      default: break;
    }
#undef PYZ_GS_PATCH8
#undef PYZ_GS_PATCH
    // -- partials of row i + 2 against the matrix as it now stands (its column i + 1 is the next step's critical distance)
    PYZ_LAP(lap, 3);
    pre = false;
    if (i + 2 < M) send_row(xnn, i + 2);
#pragma unroll
    for (int q = 0; q < PYZ_GS_E; ++q) {
      xi[q] = xnext[q];
      xnext[q] = xnn[q];
      xnn[q] = i + 3 < M ? g.all[(long long)(i + 3) * D + e[q]] : 0.0f;
    }
    PYZ_LAP(lap, 6);
  }
  if (b == 0 && tid == 0) *g.epoch = epoch0 + (unsigned)M;
#ifdef PYZ_STAMPS
  if (lane == 0 && b < PYZ_STAMP_BLOCKS)
    for (int k = 0; k < 8; ++k) pyz_dbg_buf[3][b][w][k][0] = lap[k];
#endif
}

// ---------------------------------------------------------------- peer-write exchange: the consumer's wait
// One wave: lane l re-reads flags[l] (system scope: the writers are other devices, or other processes on this one) until all
// n values have reached `value`; a stream parked on it goes on when every rank's rows of the step have landed.
__global__ void k_wait_flags(const unsigned long long *flags, int n, unsigned long long value, int spin_limit, int *fail) {
  typedef __attribute__((address_space(1))) unsigned long long gu64;
  const int l = threadIdx.x;
  for (int spins = 0;; ++spins) {
    const unsigned long long v = l < n ? __hip_atomic_load((gu64 *)(flags + l), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : value;
    if (__all(v >= value)) break;
    if (spins > spin_limit) {   // (wave-uniform)
      if (l == 0 && fail) *fail = 1;
      break;
    }
    __builtin_amdgcn_s_sleep(8);
  }
  __threadfence_system();
}

// d_loss[0] = sum_i loss_i / M   (SVGD.py:125); `fail` (the resident sweep's workgroups did not meet): NaN, counted
__global__ void k_svgd_loss(const float *loss, int n_local, int M, float *out, int *fail, int *nonfinite) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.0f;
    for (int i = 0; i < n_local; ++i) s += loss[i] / (float)M;
    if (fail && *fail) {
      s = __builtin_nanf("");
      *fail = 0;
      atomicAdd(nonfinite, 1);
    }
    out[0] = s;
  }
}

// ---------------------------------------------------------------- predict
// BayesianModel.predict (BayesianModel.py:119-128): per-sample outputs with
// NaN -> 0 and their mean.  last = (S, max_batch, C) logits or outputs.
// Two launches that cover the chip (one thread per row looping over the samples left 40 workgroups with a serial
// chain of S round trips each: 0.65 ms for 100 x 10 000 rows): k_predict_rows normalises every (sample, row) in place
// (and into `samples`), k_predict_mean sums the samples of every output element in sample order.
__global__ void k_predict_rows(float *last, long long pstride, int C, int softmax, int n, float *samples) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y;
  if (m >= n) return;
  float *z = last + s * pstride + (long long)m * C;
  float mx = 0.0f, lse = 0.0f;
  if (softmax) {
    mx = z[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, z[c]);
    float se = 0.0f;
    for (int c = 0; c < C; ++c) se += expf(z[c] - mx);
    lse = mx + logf(se);
  }
  for (int c = 0; c < C; ++c) {
    float v = softmax ? expf(z[c] - lse) : z[c];
    v = (v != v) ? 0.0f : v;
    if (samples) samples[((long long)s * n + m) * C + c] = v;
    z[c] = v;
  }
}

__global__ void k_predict_mean(const float *last, long long pstride, long long n_out, int S, float *mean, int mean_accumulate,
                               float inv_total) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_out) return;
  float acc = mean_accumulate ? mean[e] : 0.0f;
  for (int s = 0; s < S; ++s) acc += last[s * pstride + e] * inv_total;
  mean[e] = acc;
}

// n_rows independent draws of a vector Normal(loc, scale) written into columns [col0, col0 + len) of a
// row-major (n_rows, row_stride) matrix: out[r][col0 + e] = loc[e] + scale[e] z, z from Philox (seed, stream,
// step = first_draw + r, e / 4).  (BayesianModel._sample_weights, BayesianModel.py:63-77, for Normal posteriors.)
__global__ void k_sample_normal_rows(float *out, long long row_stride, long long col0, long long len, const float *loc,
                                     const float *scale, uint64_t seed, uint32_t stream, uint32_t first_draw) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long e0 = 4 * t;
  if (e0 >= len) return;
  const int r = blockIdx.y;
  const float4 q = pyz_normal4(seed, stream, first_draw + (uint32_t)r, (uint64_t)t);
  const float z[4] = {q.x, q.y, q.z, q.w};
  float *o = out + (long long)r * row_stride + col0;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (e0 + j < len) o[e0 + j] = loc[e0 + j] + scale[e0 + j] * z[j];
}

// model output for pyz_mlp_forward: softmax (if any) applied, no NaN scrubbing
__global__ void k_forward_finish(const float *last, long long pstride, int C, int softmax, int n, float *out) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y;
  if (m >= n) return;
  const float *z = last + s * pstride + (long long)m * C;
  float lse = 0.0f;
  if (softmax) {
    float mx = z[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, z[c]);
    float se = 0.0f;
    for (int c = 0; c < C; ++c) se += expf(z[c] - mx);
    lse = mx + logf(se);
  }
  for (int c = 0; c < C; ++c) out[((long long)s * n + m) * C + c] = softmax ? expf(z[c] - lse) : z[c];
}

// ---------------------------------------------------------------- noise
__global__ void k_fill_normal(float *out, long long n, uint64_t seed, uint32_t stream, uint32_t step, float mean,
                              float std) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long e0 = 4 * t;
  if (e0 >= n) return;
  const float4 q = pyz_normal4(seed, stream, step, (uint64_t)t);
  const float z[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (e0 + j < n) out[e0 + j] = mean + std * z[j];
}
