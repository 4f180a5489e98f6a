// pyz_gemm.h -- exact-fp32 MFMA kernels for the Dense layers of the hot path.
//
// Replaces Keras Dense forward (called at Pyesian/optimizers/SGLD.py:55,
// SGD.py:57, HMC.py:155, BBB.py:144, SVGD.py:106) and the Dense part of
// tf.GradientTape.gradient (SGLD.py:64, HMC.py:134, BBB.py:173, SVGD.py:110).
//
// Design (gfx950): every GEMM of this path is small (<= 1024 x 785 x 400), so
// the kernels are shaped for occupancy and launch count, not for a big tile:
//   * one wave computes one 32x32 output tile over a slice of the reduction
//     dimension with v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate: bit-exact
//     fmaf chain, 64 cycles per SIMD, one accumulator chain reaches full rate);
//   * a workgroup is S in {1,2,4,8,16} waves that split the reduction of the
//     SAME tile and combine through LDS in a fixed order (deterministic, no
//     atomics, no partial slabs in HBM); S is picked at launch so that
//     tiles x S covers the 256 CUs x 4 SIMDs;
//   * operand fragments are loaded straight from L2/HBM in the layout the MFMA
//     wants: a float4 per lane along a contiguous reduction axis feeds four
//     MFMAs (reduction order permuted identically for A and B), or one dword
//     per lane when the tile axis is the contiguous one (128-B coalesced rows);
//   * the bias is row K of the augmented (K+1) x N matrix [W; b], which is
//     exactly the flat layout (kernel then bias), so forward adds it with one
//     extra MFMA step and the weight-gradient kernel writes [dW; db] in place;
//   * blockIdx.y is the particle / chain / sample index: (P, D) parameter
//     matrices are processed in one launch.
#pragma once

#include <cstdlib>

#include "pyz_common.h"

struct DenseArgs {
  const float *in;            // A-side activations (forward/weight-grad: layer input; data-grad: delta)
  long long in_pstride;       // particle stride of `in` (0 when shared, e.g. the data batch)
  int lda;                    // row stride of `in`
  const float *theta;         // (P, D) flat parameters
  long long theta_pstride;
  long long w_off;            // offset of this layer's kernel in the flat vector
  int K, N;                   // layer input / output width
  float *out;                 // destination (see each kernel)
  long long out_pstride;
  const float *aux;           // data-grad: previous layer's output (for act'); weight-grad: delta
  long long aux_pstride;
  int act;                    // forward: this layer's activation; data-grad: previous layer's
  int vec;                    // float4 loads legal along the reduction axis
  const StepCtl *ctl;         // batch (rows) and row-index offset of this step
  const int32_t *row_idx;     // optional gather of `in` rows (layer 0 only)
  float *gather_out;          // optional (max_batch, K): contiguous copy of the gathered rows (forward, layer 0)
  StepCtl init;               // forward only: the step scalars by value when this is the first kernel of an eager step
  int init_on;
  int wt;                     // forward: write-through stores for the activations (pyz_st)
  int rows_cap;               // forward, k_dense_fwd: > 0 = readable rows of a contiguous `in` (see the kernel)
  int n_part, grid_rows;      // forward, k_dense_fwd_ring (1-D grid): particles and rows of the launch
  int n_cgrp, cgrp_w;         // forward, k_dense_fwd_ring: column groups per row block (layers wider than one workgroup's 200 columns) and their width
  int k_split;                // forward, k_dense_fwd_ring: > 1 = the reduction cut into k_split ranges of slabs, one workgroup each; raw partial sums
  float *part_out;            //   (no bias, no activation) go to part_out + split * part_stride + row * N + column: the consumer
  long long part_stride;      //   (k_head_rows) adds them in split order, then the bias and the activation
  const StepCtl *gate;        // forward (k_dense_fwd): when set, the launch does nothing on steps with gate->n % gate_mod == 0
  int gate_mod;               // (the validation forward of a device-resident BBB run: BBB.py:203 skips every tenth step)
};

// Workgroups are dealt round-robin over the 8 XCDs (each with a private 4 MiB L2), so ids
// b and b+8 share an L2.  Give every XCD a CONTIGUOUS range of tile ids instead: tiles
// that share operand panels (the same batch rows / the same weight columns) then hit one
// L2 instead of pulling the panel through the fabric eight times.  Bijective for any n;
// placement only changes speed, never results.
__device__ __forceinline__ int pyz_xcd_remap(const int bid, const int n) {
  const int q = n >> 3, r = n & 7, x = bid & 7, local = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + local;
}

// diagnostic build -DPYZ_EXP_HOT: every operand load of the reduction loops reads row / k index 0
// (same instruction stream, all loads hit L1) -- separates memory time from issue time
#ifdef PYZ_EXP_HOT
#define PYZ_HOT(v, batch) ((v) & ((batch) < 0 ? -1 : 0))
#else
#define PYZ_HOT(v, batch) (v)
#endif

__device__ __forceinline__ f32x16 pyz_mfma(float a, float b, f32x16 c) {
#ifdef PYZ_EXP_NOMFMA   // diagnostic build: keep the operands alive with one VALU op instead of the MFMA
  c[0] += a * b;
  return c;
#else
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#endif
}

__device__ __forceinline__ float pyz_act(float z, int act) {
  switch (act) {
    case PYZ_ACT_RELU: return z < 0.0f ? 0.0f : z;  // NaN propagates, like tf.nn.relu
    case PYZ_ACT_TANH: return tanhf(z);
    case PYZ_ACT_SIGMOID: return 1.0f / (1.0f + expf(-z));
    default: return z;  // linear; softmax is applied by the loss / predict kernels on the logits
  }
}

// d act / d z through the activation OUTPUT h
__device__ __forceinline__ float pyz_act_grad(float h, int act) {
  switch (act) {
    case PYZ_ACT_RELU: return h > 0.0f ? 1.0f : 0.0f;
    case PYZ_ACT_TANH: return 1.0f - h * h;
    case PYZ_ACT_SIGMOID: return h * (1.0f - h);
    default: return 1.0f;
  }
}

// Combine the S partial 32x32 tiles of a workgroup through LDS (fixed order)
// and hand each element to `store(row_in_tile, col_in_tile, value)`.
// C/D layout of v_mfma_f32_32x32x2_f32: col = lane & 31,
// row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
template <class F>
__device__ __forceinline__ void pyz_tile_epilogue(const f32x16 &acc, float *red, F store) {
  const int S = blockDim.x >> 6, w = pyz_wave_id(), l = threadIdx.x & 63;
  const int r = l & 31, h = l >> 5;
  if (S == 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) store((i & 3) + 8 * (i >> 2) + 4 * h, r, acc[i]);
    return;
  }
  float *my = red + w * 1024;
#pragma unroll
  for (int i = 0; i < 16; ++i) my[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
  __syncthreads();
  for (int e = threadIdx.x; e < 1024; e += blockDim.x) {
    float s = red[e];
    for (int ww = 1; ww < S; ++ww) s += red[ww * 1024 + e];
    store(e >> 5, e & 31, s);
  }
}

// ---------------------------------------------------------------- pipelined reduction steps
// The reduction loops are latency-bound (a wave owns a thin slice of a small GEMM) and
// the MFMA pipe of a SIMD is shared by up to four such waves.  Each loop is therefore a
// two-stage software pipeline over groups of G steps: the loads of group g+1 are issued
// before the MFMAs of group g, with scheduling barriers so the compiler cannot sink a
// load below the MFMAs it should overlap.  `load(step, a, b)` only loads (no arithmetic
// on the result, which would force an early wait); `fix(step, a, b)` masks at MFMA time.
// The last group clamps its loads to the last step and skips the MFMAs past the end.
struct PyzNoFix {
  __device__ __forceinline__ void operator()(int, float &, float &) const {}
};

struct PyzNoHook {
  __device__ __forceinline__ void operator()() const {}
};

// `hook` runs once, right after the first group's loads have been issued: work that needs no operand
// (prefetches for the epilogue, noise generation) then sits in the shadow of that first round trip
template <int G, class L, class F, class H = PyzNoHook>
__device__ __forceinline__ void pyz_pipe1(int s, const int se, f32x16 &acc, L load, F fix, H hook = H()) {
  if (s >= se) {
    hook();
    return;
  }
  float an[G], bn[G], a[G], b[G];
  const int last = se - 1;
#pragma unroll
  for (int u = 0; u < G; ++u) load(min(s + u, last), an[u], bn[u]);
  __builtin_amdgcn_sched_barrier(0);
  hook();
  __builtin_amdgcn_sched_barrier(0);
  for (;;) {
    // the copy is the wait for the loads issued one group ago; a single-phase loop keeps
    // the compiler's vmcnt bookkeeping exact (a ping-pong body drains at the back edge)
#pragma unroll
    for (int u = 0; u < G; ++u) {
      a[u] = an[u];
      b[u] = bn[u];
    }
    const int sn = s + G;
    if (sn < se) {
#pragma unroll
      for (int u = 0; u < G; ++u) load(min(sn + u, last), an[u], bn[u]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < G; ++u)
      if (s + u < se) {
        fix(s + u, a[u], b[u]);
        acc = pyz_mfma(a[u], b[u], acc);
      }
    __builtin_amdgcn_sched_barrier(0);
    s = sn;
    if (s >= se) break;
  }
}

#define PYZ_MFMA4(A, B)              \
  acc = pyz_mfma((A).x, (B).x, acc); \
  acc = pyz_mfma((A).y, (B).y, acc); \
  acc = pyz_mfma((A).z, (B).z, acc); \
  acc = pyz_mfma((A).w, (B).w, acc);

// float4 steps: one step = 8 reduction indices = four MFMAs
struct PyzNoUse4 {
  __device__ __forceinline__ void operator()(int, const float4 &) const {}
};

template <int G, class L, class U, class H = PyzNoHook>
__device__ __forceinline__ void pyz_pipe4(int c, const int ce, f32x16 &acc, L load, U use, H hook = H()) {
  if (c >= ce) {
    hook();
    return;
  }
  float4 an[G], bn[G], a[G], b[G];
  const int last = ce - 1;
#pragma unroll
  for (int u = 0; u < G; ++u) load(min(c + u, last), an[u], bn[u]);
  __builtin_amdgcn_sched_barrier(0);
  hook();   // runs once, behind the first group's loads (see pyz_pipe1)
  __builtin_amdgcn_sched_barrier(0);
  for (;;) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      a[u] = an[u];
      b[u] = bn[u];
    }
    const int cn = c + G;
    if (cn < ce) {
#pragma unroll
      for (int u = 0; u < G; ++u) load(min(cn + u, last), an[u], bn[u]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < G; ++u)
      if (c + u < ce) {
        use(c + u, a[u]);
        PYZ_MFMA4(a[u], b[u])
      }
    __builtin_amdgcn_sched_barrier(0);
    c = cn;
    if (c >= ce) break;
  }
}

template <class L, class U, class H = PyzNoHook>
__device__ __forceinline__ void pyz_steps4_all(int c, const int ce, f32x16 &acc, L load, U use, H hook = H()) {
  pyz_pipe4<2>(c, ce, acc, load, use, hook);
}
template <class L, class F, class H = PyzNoHook>
__device__ __forceinline__ void pyz_steps1_all(int s, const int se, f32x16 &acc, L load, F fix, H hook = H()) {
  pyz_pipe1<8>(s, se, acc, load, fix, hook);
}

// ---------------------------------------------------------------- forward
// acc += sum_k A[k] * W[k][n] (+ bias via the augmented row), this wave's slice of K.
// ap = this lane's input row; wl = the layer's [W; b] block (wave-uniform), n = this lane's output
// column: the W operands come through a buffer descriptor (scalar k offset + per-lane constant, no
// vector ALU per load).  gp (optional) = this lane's row of the gathered-batch copy: the A values
// are stored there once they have arrived (at MFMA time, so the store never stalls the load
// phase); the n-tiles of a row block share that copy, tile `copy_rem` of `copy_mod` stores the
// chunks congruent to it.
__device__ __forceinline__ float pyz_buf_load(const __amdgpu_buffer_rsrc_t rsrc, const unsigned voff, const unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)soff, 0));
}

template <class H = PyzNoHook>
__device__ __forceinline__ void pyz_fwd_accumulate(f32x16 &acc, const float *ap, const float *wl, const int n, const int K,
                                                   const int N, const int vec, const int w, const int S, const int h,
                                                   float *gp = nullptr, const int copy_mod = 1, const int copy_rem = 0,
                                                   H hook = H()) {
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wl), 0, (int)((unsigned)(K + 1) * (unsigned)N * 4u), 0x00020000);
  const unsigned n4 = 4u * (unsigned)n, N4 = 4u * (unsigned)N;
  const float bias = pyz_buf_load(rw, n4, (unsigned)K * N4);  // issued first, consumed last
  // The S waves of a workgroup take INTERLEAVED 8-wide chunks of K (wave w: chunks w, w+S, ...):
  // at any moment the workgroup reads S adjacent chunks = a few whole 128-B lines per input row,
  // which stay in the 32 KiB L1 and are shared by the waves.  Contiguous per-wave K ranges put
  // S x 32 different lines in flight (more than the L1 holds) and refetch every line per 16-B piece.
  const int c8 = vec ? (K >> 3) : 0;
  const unsigned v0 = (4u * h) * N4 + n4, v1 = v0 + N4, v2 = v1 + N4, v3 = v2 + N4;
  pyz_steps4_all(
      0, (c8 - w + S - 1) / S, acc,
      [&](int u, float4 &a4, float4 &b4) {
        const int c = PYZ_HOT(w + S * u, K);
        a4 = *reinterpret_cast<const float4 *>(ap + 8 * c + 4 * h);
        const unsigned so = 8u * (unsigned)c * N4;
        b4 = make_float4(pyz_buf_load(rw, v0, so), pyz_buf_load(rw, v1, so), pyz_buf_load(rw, v2, so), pyz_buf_load(rw, v3, so));
      },
      [&](int u, const float4 &a4) {
        const int c = w + S * u;
        if (gp && c % copy_mod == copy_rem) {
#ifdef PYZ_COPY_NT   // diagnostic: streaming stores for the batch copy
          typedef float pyz_f4 __attribute__((ext_vector_type(4)));
          pyz_f4 v = {a4.x, a4.y, a4.z, a4.w};
          __builtin_nontemporal_store(v, reinterpret_cast<pyz_f4 *>(gp + 8 * c + 4 * h));
#else
          *reinterpret_cast<float4 *>(gp + 8 * c + 4 * h) = a4;
#endif
        }
      },
      hook);
  const int t0 = 8 * c8, steps = (K - t0 + 1) >> 1;
  const unsigned vt = (unsigned)h * N4 + n4;
  pyz_steps1_all(
      0, (steps - w + S - 1) / S, acc,
      [&](int u, float &a, float &b) {
        const int kb = t0 + 2 * (w + S * u);
        const int kk = kb + h;
        a = ap[kk < K ? kk : 0];
        b = pyz_buf_load(rw, vt, (unsigned)kb * N4);   // row K (the bias) is inside the block; masked below
      },
      [&](int u, float &a, float &b) {
        const int kk = t0 + 2 * (w + S * u) + h;
        const bool vk = kk < K;
        if (gp && vk && copy_rem == 0) gp[kk] = a;
        a = vk ? a : 0.0f;
        b = vk ? b : 0.0f;
      });
  if (w == 0) {  // bias: row K of [W; b] against a column of ones
    acc = pyz_mfma(h == 0 ? 1.0f : 0.0f, h == 0 ? bias : 0.0f, acc);
  }
}

// out[p][m][n] = act( sum_k in[row(m)][k] * W[k][n] + b[n] ),  m < batch, n < N.
__global__ void k_dense_fwd(DenseArgs g) {
  extern __shared__ float red[];
  PYZ_STAMP(0, 0);
  const int S = blockDim.x >> 6, w = pyz_wave_id(), l = threadIdx.x & 63;
  const int r = l & 31, h = l >> 5;
  const int tiles_n = (g.N + 31) >> 5;
  if (g.gate && g.gate->n % g.gate_mod == 0) return;   // (uniform)
  const int tile = pyz_xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / tiles_n) * 32, n0 = (tile % tiles_n) * 32;
  // rows_cap > 0 (contiguous input with that many readable rows, e.g. the batch assembled ahead): the operand loads
  // do not wait for the step scalars -- those come from memory another XCD wrote (a round trip to the Infinity
  // Cache) and are only needed when the tile is stored: one scalar load, requested in front of the reduction and
  // waited for behind it.  Rows in [batch, rows_cap) are computed and dropped.
  int cap = g.rows_cap;
  long long row_off = 0;
  if (cap <= 0) {
    asm volatile("; the step scalars are needed here" ::: "memory");   // (keeps this a branch: a select would wait on both paths)
    const StepCtl ctl = pyz_ctl_first(g.ctl, g.init, g.init_on);
    cap = __builtin_amdgcn_readfirstlane(ctl.batch);
    row_off = ctl.row_off;
  }
  if (m0 >= cap) return;  // uniform per workgroup
  const int p = blockIdx.y;
  const int K = g.K, N = g.N;
  const int m = min(m0 + r, cap - 1), n = min(n0 + r, N - 1);  // clamped: rows/cols past the edge are never stored
  long long row = m;
  if (g.row_idx) row = g.row_idx[row_off + m];
  const float *ap = g.in + p * g.in_pstride + row * g.lda;
  const float *wl = g.theta + p * g.theta_pstride + g.w_off;
  f32x16 acc = {0};
  PYZ_STAMP(0, 1);
  float *gp = (g.gather_out && p == 0) ? g.gather_out + (long long)m * K : nullptr;
  // The late load is an ordinary vector load (L2-served: the line was written by another kernel) requested behind the
  // first group of operand loads; vector loads return in order, so it holds back at most the second group.
  // (Unconditional: behind a branch the compiler loses count of the loads in flight and waits for all of them.)
  int batch = 0;
  const int32_t *ctl_batch = &g.ctl->batch;
  pyz_fwd_accumulate(acc, ap, wl, n, K, N, g.vec, w, S, h, gp, tiles_n, tile % tiles_n,
                     [&]() { batch = __hip_atomic_load(ctl_batch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); });
  batch = g.rows_cap > 0 ? __builtin_amdgcn_readfirstlane(batch) : cap;
  PYZ_STAMP(0, 2);
  float *op = g.out + p * g.out_pstride;
  const int act = g.act, wt = g.wt;
  pyz_tile_epilogue(acc, red, [&](int ro, int co, float v) {
    const int mm = m0 + ro, nn = n0 + co;
    if (mm < batch && nn < N) pyz_st(op + (long long)mm * N + nn, pyz_act(v, act), wt);
  });
  PYZ_STAMP(0, 3);
}

// ---------------------------------------------------------------- forward, LDS-tiled (many particles / samples)
// The one-wave-per-tile kernel above pulls 200 KB of operands through L1 per 32 x 32 tile; with many particles
// (or posterior samples) in one launch that is what bounds it (L2 -> L1 bandwidth), not the matrix pipe.  Here a
// workgroup of four waves owns 128 rows x 32 NT columns: 16-deep slabs of both operands are staged in LDS
// (A k-major so that the MFMA fragments are read with consecutive lanes; the two reduction halves of a fragment
// sit 32 banks apart), the next slab travels global -> registers while the current one feeds NT MFMAs per A
// fragment.  Operand bytes per 32 x 32 tile of output: 25 KB (NT = 7) instead of 200 KB.
// Needs K % 4 == 0 with 16-byte aligned input rows, and an even N with 8-byte aligned [W; b] blocks (checked at
// launch); exact fp32 as above.
template <int NT>
__global__ void __launch_bounds__(256) k_dense_fwd_lds(DenseArgs g) {
  constexpr int BN = 32 * NT, BM = 128, BK = 16;
  constexpr int AS = BM + 32;                          // row stride of As: the h = 1 half lands 32 banks further
  constexpr int BS = (BN % 64 == 0) ? BN + 32 : BN;    // same for Bs
  constexpr int BJ = NT;                               // 8-byte pieces of a B slab per thread (16 rows x BN / 2 = 256 NT):
                                                       // a particle's [W; b] block is only 8-byte aligned in general
  __shared__ __attribute__((aligned(16))) float As[2][BK][AS];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][BS];
  const int t = threadIdx.x, w = pyz_wave_id(), l = t & 63, r = l & 31, h = l >> 5;
  const StepCtl ctl = pyz_ctl_first(g.ctl, g.init, g.init_on);
  const int batch = ctl.batch;
  const int K = g.K, N = g.N;
  const int tiles_n = (N + BN - 1) / BN;
  const int tile = pyz_xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  if (m0 >= batch) return;  // uniform (also the padding workgroups of the launch)
  const int p = blockIdx.y;
  // staging assignment.  A: thread -> row ra = t / 2 of the tile, two 16-byte pieces at k = 8 (t & 1) + {0, 4}
  const int ra = t >> 1, ka = 8 * (t & 1);
  const int ma = min(m0 + ra, batch - 1);
  long long rowa = ma;
  if (g.row_idx) rowa = g.row_idx[ctl.row_off + ma];
  const float *ap = g.in + p * g.in_pstride + rowa * g.lda + ka;
  const float *wl = g.theta + p * g.theta_pstride + g.w_off;
  float *gp = (g.gather_out && p == 0 && n0 == 0 && m0 + ra < batch) ? g.gather_out + (long long)ma * K + ka : nullptr;
  float4 va[2];
  float2 vb[BJ];
  auto fetch = [&](const int k0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + ka + 4 * j;
      va[j] = k < K ? *reinterpret_cast<const float4 *>(ap + k0 + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int idx = t + 256 * j, kr = idx / (BN / 2), c2 = idx % (BN / 2);
      const int k = k0 + kr, n = n0 + 2 * c2;
      vb[j] = (k < K && n < N) ? *reinterpret_cast<const float2 *>(wl + (long long)k * N + n) : make_float2(0.f, 0.f);
    }
  };
  auto stage = [&](const int buf, const int k0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kk = ka + 4 * j;
      As[buf][kk + 0][ra] = va[j].x;
      As[buf][kk + 1][ra] = va[j].y;
      As[buf][kk + 2][ra] = va[j].z;
      As[buf][kk + 3][ra] = va[j].w;
      if (gp && k0 + kk < K) *reinterpret_cast<float4 *>(gp + k0 + 4 * j) = va[j];
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int idx = t + 256 * j, kr = idx / (BN / 2), c2 = idx % (BN / 2);
      *reinterpret_cast<float2 *>(&Bs[buf][kr][2 * c2]) = vb[j];
    }
  };
  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x16{0};
  const int ns = (K + BK - 1) / BK;
  PYZ_STAMP(0, 0);
  fetch(0);
  stage(0, 0);
  __syncthreads();
  PYZ_STAMP(0, 1);
  for (int s = 0; s < ns; ++s) {
    const int buf = s & 1;
    if (s == 24) PYZ_STAMP(0, 2);
    if (s + 1 < ns) fetch((s + 1) * BK);
    // (reading the fragments of step kp + 1 while the NT matrix instructions of step kp issue -- two register sets
    // pinned by scheduling barriers -- changes nothing: 215.9 against 215.7 us at C5; the second wave of the SIMD
    // already fills the LDS round trips of the first)
#pragma unroll
    for (int kp = 0; kp < BK / 2; ++kp) {
      const float a = As[buf][2 * kp + h][32 * w + r];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = pyz_mfma(a, Bs[buf][2 * kp + h][32 * nt + r], acc[nt]);
    }
    if (s == 24) PYZ_STAMP(0, 3);
    if (s + 1 < ns) stage(buf ^ 1, (s + 1) * BK);
    if (s == 24) PYZ_STAMP(0, 4);
    __syncthreads();
    if (s == 24) PYZ_STAMP(0, 5);
  }
  PYZ_STAMP(0, 6);
  float *op = g.out + p * g.out_pstride;
  const int act = g.act, wt = g.wt;
  // The tile leaves through LDS (the slabs are dead; every wave has 4 KB of its own, no barrier): a lane holds a COLUMN
  // of a 32 x 32 sub-tile, 16 dword stores per column tile -- 112 per thread, and the store phase was 23 us of the
  // 220 us kernel at C5, bound by their issue.  Written to LDS as they stand and read back as rows, a lane stores four
  // consecutive columns with one 16-byte instruction: 28 per thread.  (N % 4 == 0 with 16-byte aligned output rows;
  // otherwise the dword stores.)
  const bool wide = (N & 3) == 0 && (g.out_pstride & 3) == 0 && (reinterpret_cast<uintptr_t>(g.out) & 15) == 0;
  if (wide) {
    __syncthreads();   // every wave is done with the slabs
    float *tw = &As[0][0][0] + 1024 * w;
    static_assert(2 * BK * AS >= 4 * 1024, "tile scratch");
    const int tr = l >> 3, tc = 4 * (l & 7);   // read-back: rows tr, tr + 8, tr + 16, tr + 24, columns tc .. tc + 3
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int nn = n0 + 32 * nt + r;
      const float bias = nn < N ? wl[(long long)K * N + nn] : 0.0f;
#pragma unroll
      for (int i = 0; i < 16; ++i) tw[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = pyz_act(acc[nt][i] + bias, act);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ro = tr + 8 * q, mm = m0 + 32 * w + ro, nc = n0 + 32 * nt + tc;
        const float4 v = *reinterpret_cast<const float4 *>(tw + ro * 32 + tc);
        if (mm < batch && nc < N) {   // (N % 4 == 0: a group of four columns is inside or outside as a whole)
          float *d = op + (long long)mm * N + nc;
          if (wt) {
            const f32x4 vv = {v.x, v.y, v.z, v.w};
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(d), "v"(vv) : "memory");
          } else {
            *reinterpret_cast<float4 *>(d) = v;
          }
        }
      }
    }
    PYZ_STAMP(0, 7);
    return;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int nn = n0 + 32 * nt + r;
    if (nn < N) {
      const float bias = wl[(long long)K * N + nn];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int mm = m0 + 32 * w + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (mm < batch) pyz_st(op + (long long)mm * N + nn, pyz_act(acc[nt][i] + bias, act), wt);
      }
    }
  }
  PYZ_STAMP(0, 7);
}

// ---------------------------------------------------------------- data gradient
// out[p][m][j] = ( sum_n delta[p][m][n] * W[j][n] ) * act'(hprev[p][m][j]),  j < K.
// `in` = delta (row stride N), `aux` = previous layer's output (row stride K).
__global__ void k_dense_bwd_data(DenseArgs g) {
  extern __shared__ float red[];
  const int S = blockDim.x >> 6, w = pyz_wave_id(), l = threadIdx.x & 63;
  const int r = l & 31, h = l >> 5;
  const int batch = g.ctl->batch;
  const int K = g.K, N = g.N;
  const int tiles_j = (K + 31) >> 5;
  const int tile = pyz_xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / tiles_j) * 32, j0 = (tile % tiles_j) * 32;
  if (m0 >= batch) return;
  const int p = blockIdx.y;
  const int m = min(m0 + r, batch - 1), j = min(j0 + r, K - 1);
  const float *ap = g.in + p * g.in_pstride + (long long)m * N;
  const float *wp = g.theta + p * g.theta_pstride + g.w_off + (long long)j * N;
  f32x16 acc = {0};
  const int c8 = g.vec ? (N >> 3) : 0;
  pyz_steps4_all(
      (c8 * w) / S, (c8 * (w + 1)) / S, acc,
      [&](int c, float4 &a4, float4 &b4) {
        const int k = 8 * c + 4 * h;
        a4 = *reinterpret_cast<const float4 *>(ap + k);
        b4 = *reinterpret_cast<const float4 *>(wp + k);
      },
      PyzNoUse4());
  {
    const int t0 = 8 * c8, steps = (N - t0 + 1) >> 1;
    pyz_steps1_all(
        (steps * w) / S, (steps * (w + 1)) / S, acc,
        [&](int s, float &a, float &b) {
          const int kk = t0 + 2 * s + h;
          const int kc = kk < N ? kk : 0;
          a = ap[kc];
          b = wp[kc];
        },
        [&](int s, float &a, float &b) {
          const bool vk = t0 + 2 * s + h < N;
          a = vk ? a : 0.0f;
          b = vk ? b : 0.0f;
        });
  }
  float *op = g.out + p * g.out_pstride;
  const float *hp = g.aux + p * g.aux_pstride;
  const int act = g.act, wt = g.wt;
  pyz_tile_epilogue(acc, red, [&](int ro, int co, float v) {
    const int mm = m0 + ro, jj = j0 + co;
    if (mm < batch && jj < K) {
      const long long o = (long long)mm * K + jj;
      pyz_st(op + o, v * pyz_act_grad(hp[o], act), wt);
    }
  });
}

// ---------------------------------------------------------------- weight gradient
// out[p][w_off + i*N + n] = sum_b A(b, i) * delta[p][b][n],  i <= K, n < N, with
// A(b, i) = in[row(b)][i] for i < K and 1 for i == K (the bias row).
// `in` = layer input, `aux` = delta (row stride N).
// With a row gather, the 32 row indices of a 16-step group are fetched by ONE
// coalesced load (lane j holds the index of batch row 2*s0 + j) and handed to the
// step that needs them with a lane permute, so no load depends on another load.
template <int G, bool GATHER>
__device__ __forceinline__ void pyz_wgrad_steps(int &s, const int se, f32x16 &acc, const float *ap, const float *dp,
                                                const int32_t *idx, const int lda, const int N, const int batch,
                                                const int h, const int r, const bool is_w, const bool is_b) {
  for (; s + G <= se; s += G) {
    float a[G], d[G];
    int idxv = 0;
    if (GATHER) idxv = idx[min(2 * s + r, batch - 1)];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int b = 2 * (s + u) + h;
      const bool vb = b < batch;
      const int bc = vb ? b : 0;
      long long row = bc;
      if (GATHER) row = __shfl(idxv, 2 * u + h, 64);
      const float av = ap[row * lda], dv = dp[(long long)bc * N];
      a[u] = av;
      d[u] = dv;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const bool vb = 2 * (s + u) + h < batch;
      const float av = vb ? (is_w ? a[u] : (is_b ? 1.0f : 0.0f)) : 0.0f;
      acc = pyz_mfma(av, vb ? d[u] : 0.0f, acc);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

__global__ void k_dense_bwd_weight(DenseArgs g) {
  extern __shared__ float red[];
  const int S = blockDim.x >> 6, w = pyz_wave_id(), l = threadIdx.x & 63;
  const int r = l & 31, h = l >> 5;
  const int batch = g.ctl->batch;
  const int K = g.K, N = g.N;
  const int tiles_n = (N + 31) >> 5;
  const int tile = pyz_xcd_remap(blockIdx.x, gridDim.x);
  const int i0 = (tile / tiles_n) * 32, n0 = (tile % tiles_n) * 32;
  const int p = blockIdx.y;
  const int i = i0 + r, n = min(n0 + r, N - 1);
  const int ic = min(i, K - 1);
  const bool is_w = i < K, is_b = i == K;
  const float *ap = g.in + p * g.in_pstride + ic;
  const float *dp = g.aux + p * g.aux_pstride + n;
  const int32_t *idx = g.row_idx ? g.row_idx + g.ctl->row_off : nullptr;
  f32x16 acc = {0};
  const int steps = (batch + 1) >> 1;
  int s = (steps * w) / S;
  const int se = (steps * (w + 1)) / S;
  if (idx) {
    pyz_wgrad_steps<16, true>(s, se, acc, ap, dp, idx, g.lda, N, batch, h, r, is_w, is_b);
    pyz_wgrad_steps<4, true>(s, se, acc, ap, dp, idx, g.lda, N, batch, h, r, is_w, is_b);
    pyz_wgrad_steps<1, true>(s, se, acc, ap, dp, idx, g.lda, N, batch, h, r, is_w, is_b);
  } else {
    pyz_wgrad_steps<16, false>(s, se, acc, ap, dp, idx, g.lda, N, batch, h, r, is_w, is_b);
    pyz_wgrad_steps<4, false>(s, se, acc, ap, dp, idx, g.lda, N, batch, h, r, is_w, is_b);
    pyz_wgrad_steps<1, false>(s, se, acc, ap, dp, idx, g.lda, N, batch, h, r, is_w, is_b);
  }
  float *op = g.out + p * g.out_pstride + g.w_off;
  pyz_tile_epilogue(acc, red, [&](int ro, int co, float v) {
    const int ii = i0 + ro, nn = n0 + co;
    if (ii <= K && nn < N) op[(long long)ii * N + nn] = v;
  });
}

// ---------------------------------------------------------------- launch helpers
static inline int pyz_env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

// compute units of the current device (256 on MI355X); 256 if the query fails
static inline int pyz_cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    return v;
  }();
  return n;
}

// single-chain launches store their outputs write-through (pyz_st); PYZ_WT=0: plain stores
static inline int pyz_wt_for(int P) {
  static const int on = pyz_env_int("PYZ_WT", 1);   // 2: launches of any number of particles (measurement)
  return (on == 2 || (on && P == 1)) ? 1 : 0;
}

// waves per workgroup: split the reduction until the launch covers the chip
// (PYZ_WAVES_TARGET waves) or a wave would get fewer than PYZ_MIN_STEPS MFMA steps.
static inline int pyz_pick_waves(long long tiles, long long mfma_steps) {
  static const int target = pyz_env_int("PYZ_WAVES_TARGET", 3072);
  static const int min_steps = pyz_env_int("PYZ_MIN_STEPS", 8);
  int S = 1;
  while (S < 16 && tiles * S < target && mfma_steps / (2 * S) >= min_steps) S *= 2;
  return S;
}

// grid.x in multiples of 8 when the launch has several particles: workgroups are dealt round-robin over the 8
// XCDs by their LINEAR id, so only then does a tile id land on the same XCD (the same L2) for every particle --
// with 79 row tiles per particle the rows an XCD touches shift from particle to particle and its L2 never keeps
// them (predict, 100 draws x 10 000 rows: 18.0 ms; padded 14.6).  The padding workgroups own no tile.
static inline unsigned pyz_pad8(long long n, int P) { return (unsigned)(P > 1 ? (n + 7) / 8 * 8 : n); }

// launches that fill the chip with one wave per tile (many particles / samples) take the LDS-tiled forward
static inline bool pyz_fwd_takes_lds(const DenseArgs &g, int grid_batch, int P, int S) {
  const int lds_on = pyz_env_int("PYZ_FWD_LDS", 1);  // read per call: tests flip it
  const bool lds_ok = g.K % 4 == 0 && g.N % 2 == 0 && g.lda % 4 == 0 && g.w_off % 2 == 0 && (P == 1 || g.theta_pstride % 2 == 0) &&
                      (P == 1 || g.in_pstride % 4 == 0) && (reinterpret_cast<uintptr_t>(g.theta) & 7) == 0 &&
                      (reinterpret_cast<uintptr_t>(g.in) & 15) == 0 &&
                      (!g.gather_out || (reinterpret_cast<uintptr_t>(g.gather_out) & 15) == 0);
  // (measured: with few row tiles per particle -- C5: 8 -- the LDS kernel wins, 0.23 against 0.32 ms; with many --
  // predict, 79 -- the one-wave kernel runs out of an L2 that keeps its slice of the rows and wins, 14.5 against 16.7 ms)
  const int lds_max_rows = pyz_env_int("PYZ_FWD_LDS_MAXROWS", 2048);
  // ... and its own 128-row tiles must fill the chip: 16 particles x 8 row tiles = 128 workgroups took 130 us, 2.8 times
  // the 8-particle launch of the one-wave kernel.  (A third kernel for the sizes in between -- 32-row tiles, the slab's
  // reduction steps split over the four waves -- was built and measured: with one workgroup per CU and 28 matrix
  // instructions per wave and slab it cannot hide the latency of its own slab loads: 54 - 60 us at 8 particles against
  // 46 for the one-wave kernel, 21 - 27 against 13.5 us on the 784 -> 400 layer; removed.)
  const int NT = g.N <= 64 ? 2 : (g.N <= 128 ? 4 : 7);
  const long long wg128 = (long long)((grid_batch + 127) / 128) * ((g.N + 32 * NT - 1) / (32 * NT)) * P;
  // (row counts past lds_max_rows: only launches of many particles -- predict with all its draws in one launch: 3.3 ms
  // against 4.9 for 100 draws x 10 000 rows; with two draws per launch the one-wave kernel won, 14.5 against 16.7 ms)
  return S == 1 && !g.gate && lds_on && lds_ok && g.N >= 48 && grid_batch >= 128 && (grid_batch <= lds_max_rows || P >= 8) &&
         wg128 >= pyz_env_int("PYZ_FWD_LDS_MINWG", 256);
}

static inline void pyz_launch_fwd(const DenseArgs &g, int grid_batch, int P, hipStream_t st) {
  const long long tiles = (long long)((grid_batch + 31) / 32) * ((g.N + 31) / 32);
  const int S = pyz_pick_waves(tiles * P, (g.K + 1) / 2 + 1);
  if (pyz_fwd_takes_lds(g, grid_batch, P, S)) {
    const int NT = g.N <= 64 ? 2 : (g.N <= 128 ? 4 : 7);
    const long long tl = (long long)((grid_batch + 127) / 128) * ((g.N + 32 * NT - 1) / (32 * NT));
    const dim3 grid(pyz_pad8(tl, P), P), block(256);
    if (NT == 2) PYZ_LAUNCH(k_dense_fwd_lds<2>, grid, block, 0, st, g);
    else if (NT == 4) PYZ_LAUNCH(k_dense_fwd_lds<4>, grid, block, 0, st, g);
    else PYZ_LAUNCH(k_dense_fwd_lds<7>, grid, block, 0, st, g);
    return;
  }
  PYZ_LAUNCH(k_dense_fwd, dim3(pyz_pad8(tiles, P), P), dim3(64 * S), S > 1 ? S * 4096 : 0, st, g);
}

static inline void pyz_launch_bwd_data(const DenseArgs &g, int grid_batch, int P, hipStream_t st) {
  const long long tiles = (long long)((grid_batch + 31) / 32) * ((g.K + 31) / 32);
  const int S = pyz_pick_waves(tiles * P, (g.N + 1) / 2);
  PYZ_LAUNCH(k_dense_bwd_data, dim3(pyz_pad8(tiles, P), P), dim3(64 * S), S > 1 ? S * 4096 : 0, st, g);
}

static inline void pyz_launch_bwd_weight(const DenseArgs &g, int grid_batch, int P, hipStream_t st) {
  const long long tiles = (long long)((g.K + 1 + 31) / 32) * ((g.N + 31) / 32);
  const int S = pyz_pick_waves(tiles * P, (grid_batch + 1) / 2);
  PYZ_LAUNCH(k_dense_bwd_weight, dim3((unsigned)tiles, P), dim3(64 * S), S > 1 ? S * 4096 : 0, st, g);
}
