// pyz_api.hip -- extern "C" entry points declared in include/pyz.h.
// Host-side orchestration only: shape checks, workspace, kernel launches.

#include <algorithm>
#include <cmath>
#include <cstring>

#include "pyz_common.h"
#include "pyz_fused.h"
#include "pyz_gemm.h"
#include "pyz_gemm_ring.h"
#include "pyz_hmc_fused.h"
#include "pyz_hmc_multi.h"
#include "pyz_kernels.h"
#include "pyz_rng.h"

namespace {

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

int ensure_bytes(void **ptr, size_t *cap, size_t need, pyz_mlp *m) {
  if (*cap >= need) return PYZ_OK;
  if (*ptr) PYZ_HIP(hipFree(*ptr));
  *ptr = nullptr;
  *cap = 0;
  PYZ_HIP(hipMalloc(ptr, need));
  m->ws_bytes += need;
  *cap = need;
  return PYZ_OK;
}

struct Extra {  // lazily sized buffers kept beside the plan
  size_t grad_cap = 0, grad2_cap = 0, qsave_cap = 0, part_cap = 0, part2_cap = 0, scal_cap = 0;
  void *hm_buf = nullptr;  // k_hmc_multi: state / slab ping-pong buffers
  size_t hm_cap = 0;
  void *hm_res_buf = nullptr;  // k_hmc_resident: per-chain epochs and the granule rows
  size_t hm_res_cap = 0;
  void *fwd_part = nullptr;    // k_dense_fwd_ring with a split reduction: (k_split, max_batch, width) raw partial sums
  size_t fwd_part_cap = 0;
  void *gs_res_buf = nullptr;  // k_svgd_gs_resident: epoch, fail flag, granules of the partials and of the kernel row
  size_t gs_res_cap = 0;
  hipGraph_t hm_graph = nullptr;  // the launch sequence of one sliced HMC proposal
  hipGraphExec_t hm_exec = nullptr;
  unsigned long long hm_key = 0;
  double *part2 = nullptr;
  // per-run tables go up through two pinned buffers used in turn (an event each: rewritten only once its copy has left)
  void *tab_host = nullptr;
  size_t tab_host_cap = 0;  // bytes per buffer
  hipEvent_t tab_ev[2] = {};
  unsigned long long tab_count = 0;
  // per-proposal HMC scalars (uniforms + HmcCall) go up through a ring of pinned slots: a copy from pageable
  // memory would make every pyz_hmc_step wait for the stream, i.e. serialise the host with the device
  static constexpr int UP_SLOTS = 64;
  unsigned char *up_host = nullptr;
  size_t up_slot_bytes = 0;
  hipEvent_t up_ev[UP_SLOTS] = {};
  unsigned long long up_count = 0;
};

}  // namespace

// The opaque handle = plan + lazily grown buffers.
struct pyz_mlp_full : pyz_mlp {
  Extra x;
};

namespace {

inline pyz_mlp_full *full(pyz_mlp *m) { return static_cast<pyz_mlp_full *>(m); }

void drop_graphs(pyz_mlp *m) {
  for (int c = 0; c < PYZ_GRAPH_CHUNKS; ++c) {
    if (m->graph_exec[c]) (void)hipGraphExecDestroy(m->graph_exec[c]);
    if (m->graph[c]) (void)hipGraphDestroy(m->graph[c]);
    m->graph_exec[c] = nullptr;
    m->graph[c] = nullptr;
    m->graph_len[c] = 0;
  }
}

int need_grad(pyz_mlp *m, int P) {
  m->grad_owner = 2;   // (pyz_svgd_gradients sets 1 after this call)
  return ensure_bytes((void **)&m->grad, &full(m)->x.grad_cap, sizeof(float) * (size_t)P * m->D + 64, m);
}
int need_grad2(pyz_mlp *m, int P) {
  return ensure_bytes((void **)&m->grad2, &full(m)->x.grad2_cap, sizeof(float) * (size_t)P * m->D + 64, m);
}
int need_qsave(pyz_mlp *m, int P) {
  return ensure_bytes((void **)&m->qsave, &full(m)->x.qsave_cap, sizeof(float) * (size_t)P * m->D + 64, m);
}
int need_part(pyz_mlp *m, size_t n_doubles) {
  return ensure_bytes((void **)&m->part, &full(m)->x.part_cap, sizeof(double) * n_doubles, m);
}
int need_part2(pyz_mlp *m, size_t n_doubles, bool keep_kernel_matrix = false) {
  if (!keep_kernel_matrix) m->km_valid = false;   // the float64 scratch is about to be rewritten
  return ensure_bytes((void **)&full(m)->x.part2, &full(m)->x.part2_cap, sizeof(double) * n_doubles, m);
}

int check_call(const pyz_mlp *m, int P, int batch) {
  if (!m) return pyz_fail(PYZ_E_INVALID, "null plan");
  if (batch <= 0 || batch > m->max_batch)
    return pyz_fail(PYZ_E_SHAPE, "batch %d outside [1, %d] of the plan", batch, m->max_batch);
  if (P <= 0 || P > m->max_p)
    return pyz_fail(PYZ_E_SHAPE, "particle count %d outside [1, %d] of the plan", P, m->max_p);
  return PYZ_OK;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int set_ctl(pyz_mlp *m, int slot, int batch, float lr, long long n, long long row_off, int i, hipStream_t st,
            int slot0 = 0, int n_run = 0) {
  m->pend_on = false;
  PYZ_LAUNCH(k_set_ctl, dim3(1), dim3(1), 0, st, m->ctl + slot, batch, lr, n, row_off, i, slot0, n_run);
  PYZ_LAUNCH_CHECK();
  return PYZ_OK;
}

// eager steps: no launch for the scalars -- the first kernel of the fused step carries them (pyz_ctl_first)
void set_ctl_lazy(pyz_mlp *m, int batch, float lr, long long n) {
  m->pend = StepCtl{};
  m->pend.batch = batch;
  m->pend.lr = lr;
  m->pend.n = n;
  m->pend_on = true;
}

void flush_ctl(pyz_mlp *m, hipStream_t st) {
  if (!m->pend_on) return;
  m->pend_on = false;
  PYZ_LAUNCH(k_set_ctl, dim3(1), dim3(1), 0, st, m->ctl, m->pend.batch, m->pend.lr, m->pend.n, m->pend.row_off, m->pend.i,
                     m->pend.slot0, 0);
}

// arguments of the forward pass of layer l (activations land in m->act[l])
DenseArgs forward_args(pyz_mlp *m, int l, const float *theta, long long theta_ps, const float *x, const int32_t *row_idx,
                       const StepCtl *ctl, float *gather_out) {
  DenseArgs g{};
  g.K = m->dims[l];
  g.N = m->dims[l + 1];
  if (l == 0) {
    g.in = x;
    g.in_pstride = 0;
    g.row_idx = row_idx;
    g.gather_out = row_idx ? gather_out : nullptr;
  } else {
    g.in = m->act[l - 1];
    g.in_pstride = (long long)m->max_batch * g.K;
    g.row_idx = nullptr;
  }
  g.lda = g.K;
  g.theta = theta;
  g.theta_pstride = theta_ps;
  g.w_off = m->w_off[l];
  g.out = m->act[l];
  g.out_pstride = (long long)m->max_batch * g.N;
  g.act = m->acts[l];
  g.vec = (g.K % 8 == 0) && aligned16(g.in) ? 1 : 0;
  g.ctl = ctl;
  if (l == 0 && m->pend_on && ctl == m->ctl) {  // first kernel of an eager step: carries the step scalars
    g.init = m->pend;
    g.init_on = 1;
    m->pend_on = false;
  }
  return g;
}

// forward over layers [0, l_end); activations land in m->act[l]
void launch_forward(pyz_mlp *m, const float *theta, long long theta_ps, int P, const float *x,
                    const int32_t *row_idx, int grid_batch, const StepCtl *ctl, hipStream_t st,
                    float *gather_out = nullptr, int l_end = -1, const StepCtl *gate = nullptr, int gate_mod = 0) {
  if (l_end < 0) l_end = m->L;
  for (int l = 0; l < l_end; ++l) {
    DenseArgs g = forward_args(m, l, theta, theta_ps, x, row_idx, ctl, gather_out);
    g.wt = pyz_wt_for(P);
    g.gate = gate;
    g.gate_mod = gate_mod;
    if (pyz_launch_fwd_ring(g, grid_batch, P, st)) continue;   // mid-size launches: 32-row blocks through the LDS-DMA ring
    if (!g.row_idx && !g.init_on && !g.gather_out) g.rows_cap = std::min(grid_batch, m->max_batch);   // (layer 0 of a caller-owned x: its rows cover grid_batch by contract)
    pyz_launch_fwd(g, grid_batch, P, st);
  }
}

void launch_loss(pyz_mlp *m, int P, const void *y, const int32_t *row_idx, int grid_batch, const StepCtl *ctl,
                 bool want_delta, hipStream_t st) {
  LossArgs g{};
  const int C = m->dims[m->L];
  g.out_last = m->act[m->L - 1];
  g.pstride = (long long)m->max_batch * C;
  g.C = C;
  g.y = y;
  g.delta = want_delta ? m->delta[m->L - 1] : nullptr;
  g.part = m->part;
  g.nblk = cdiv(m->max_batch, 256);
  g.act_last = m->acts[m->L - 1];
  g.ctl = ctl;
  g.row_idx = row_idx;
  dim3 grid(cdiv(grid_batch, 256), P);
  // partial slots of blocks past grid_batch must read as zero: the finalisers sum nblk of them
  if ((int)grid.x < g.nblk) g.nblk = grid.x;
  if (m->loss == PYZ_LOSS_SCCE)
    PYZ_LAUNCH(k_loss_scce, grid, dim3(256), 0, st, g);
  else
    PYZ_LAUNCH(k_loss_mse, grid, dim3(256), 0, st, g);
}

inline int loss_nblk(const pyz_mlp *m, int grid_batch) { return std::min(cdiv(m->max_batch, 256), cdiv(grid_batch, 256)); }

void launch_backward(pyz_mlp *m, const float *theta, long long theta_ps, int P, const float *x,
                     const int32_t *row_idx, int grid_batch, const StepCtl *ctl, float *grad, hipStream_t st) {
  for (int l = m->L - 1; l >= 0; --l) {
    const int K = m->dims[l], N = m->dims[l + 1];
    if (l > 0) {  // delta[l-1] = (delta[l] W_l^T) * act'(h_{l-1})
      DenseArgs g{};
      g.K = K;
      g.N = N;
      g.in = m->delta[l];
      g.in_pstride = (long long)m->max_batch * N;
      g.lda = N;
      g.theta = theta;
      g.theta_pstride = theta_ps;
      g.w_off = m->w_off[l];
      g.out = m->delta[l - 1];
      g.out_pstride = (long long)m->max_batch * K;
      g.aux = m->act[l - 1];
      g.aux_pstride = (long long)m->max_batch * K;
      g.act = m->acts[l - 1];
      g.vec = (N % 8 == 0) && (m->w_off[l] % 4 == 0) && (P == 1 || theta_ps % 4 == 0) && aligned16(theta) ? 1 : 0;
      g.ctl = ctl;
      pyz_launch_bwd_data(g, grid_batch, P, st);
    }
    DenseArgs g{};
    g.K = K;
    g.N = N;
    if (l == 0) {
      g.in = x;
      g.in_pstride = 0;
      g.row_idx = row_idx;
    } else {
      g.in = m->act[l - 1];
      g.in_pstride = (long long)m->max_batch * K;
    }
    g.lda = K;
    g.aux = m->delta[l];
    g.aux_pstride = (long long)m->max_batch * N;
    g.out = grad;
    g.out_pstride = m->D;
    g.w_off = m->w_off[l];
    g.ctl = ctl;
    pyz_launch_bwd_weight(g, grid_batch, P, st);
  }
}

// ---- fused path (pyz_fused.h): usable when the last layer fits one 32-wide tile
inline bool can_fuse(const pyz_mlp *m) { return m->dims[m->L] <= 32; }

void launch_head(pyz_mlp *m, const float *theta, long long theta_ps, int P, const float *x, const void *y,
                 const int32_t *row_idx, int grid_batch, const StepCtl *ctl, bool want_delta, hipStream_t st,
                 const StepCtl *gate = nullptr, int gate_mod = 0, int k_split = 0) {
  const int l = m->L - 1;
  HeadArgs g{};
  g.gate = gate;
  g.gate_mod = gate_mod;
  if (k_split > 1) {   // the hidden layer arrives as the partial sums of a split-reduction forward (ksplit_forward_ok)
    g.hparts = static_cast<const float *>(full(m)->x.fwd_part);
    g.n_hparts = k_split;
    g.hpart_stride = (long long)m->max_batch * m->dims[l];
    g.bias_prev = theta + m->w_off[l - 1] + (long long)m->dims[l - 1] * m->dims[l];
    g.h_store = m->act[l - 1];
  }
  g.K = m->dims[l];
  g.N = m->dims[l + 1];
  if (l == 0) {
    g.hin = x;
    g.hin_pstride = 0;
    g.gather_hin = 1;
  } else {
    g.hin = m->act[l - 1];
    g.hin_pstride = (long long)m->max_batch * g.K;
    g.gather_hin = 0;
  }
  g.lda = g.K;
  g.row_idx = row_idx;
  g.theta = theta;
  g.theta_pstride = theta_ps;
  g.w_off = m->w_off[l];
  g.loss = m->loss;
  g.act_last = m->acts[l];
  g.act_prev = l > 0 ? m->acts[l - 1] : PYZ_ACT_LINEAR;
  g.vec = (g.K % 8 == 0) && aligned16(g.hin) ? 1 : 0;
  g.y = y;
  g.out_last = m->act[l];
  g.delta_last = want_delta ? m->delta[l] : nullptr;
  g.delta_prev = (want_delta && l > 0) ? m->delta[l - 1] : nullptr;
  g.last_pstride = (long long)m->max_batch * g.N;
  g.prev_pstride = (long long)m->max_batch * g.K;
  g.part = m->part;
  g.ctl = ctl;
  g.wt = pyz_wt_for(P);
  if (m->pend_on && ctl == m->ctl) {  // no hidden layer: the head is the first kernel of the eager step
    g.init = m->pend;
    g.init_on = 1;
    m->pend_on = false;
  }
  // one wave per batch row when the lane-resident operands fit (UT units x NP classes per lane)
  static const int rows_on = pyz_env_int("PYZ_HEAD_ROWS", 1);
  const int UT = g.K <= 64 ? 1 : g.K <= 256 ? 4 : g.K <= 512 ? 8 : g.K <= 1024 ? 16 : 0;
  const int NP = g.N <= 4 ? 4 : g.N <= 8 ? 8 : g.N <= 12 ? 12 : g.N <= 16 ? 16 : g.N <= 24 ? 24 : 32;
  if (rows_on && UT && UT * NP <= 128) {
    g.nblk = grid_batch;  // one loss partial per row
    m->cur_nblk = g.nblk;
    // rows per wave: four when the launch has tens of thousands of rows (many particles), else one
    static const int rw_rows = pyz_env_int("PYZ_HEAD_RW_ROWS", 32768);
    const int RW = (long long)P * grid_batch >= rw_rows ? 4 : 1;
    const dim3 grid((unsigned)cdiv(cdiv(grid_batch, RW), 4), P), block(256);
    if (g.n_hparts > 0) {   // behind a split-reduction forward (ksplit_for: UT == 4, at most 16 classes, one chain)
      if (NP == 4) PYZ_LAUNCH((k_head_rows<4, 4, 1, true>), grid, block, 0, st, g);
      else if (NP == 8) PYZ_LAUNCH((k_head_rows<4, 8, 1, true>), grid, block, 0, st, g);
      else if (NP == 12) PYZ_LAUNCH((k_head_rows<4, 12, 1, true>), grid, block, 0, st, g);
      else PYZ_LAUNCH((k_head_rows<4, 16, 1, true>), grid, block, 0, st, g);
      return;
    }
#define PYZ_HEAD_ROWS_CASE(U, C)                                                               \
  if (UT == U && NP == C) {                                                                    \
    if (RW == 4) PYZ_LAUNCH((k_head_rows<U, C, 4>), grid, block, 0, st, g);                     \
    else PYZ_LAUNCH((k_head_rows<U, C, 1>), grid, block, 0, st, g);                            \
    return;                                                                                    \
  }
    PYZ_HEAD_ROWS_CASE(1, 4) PYZ_HEAD_ROWS_CASE(1, 8) PYZ_HEAD_ROWS_CASE(1, 12) PYZ_HEAD_ROWS_CASE(1, 16)
    PYZ_HEAD_ROWS_CASE(1, 24) PYZ_HEAD_ROWS_CASE(1, 32)
    PYZ_HEAD_ROWS_CASE(4, 4) PYZ_HEAD_ROWS_CASE(4, 8) PYZ_HEAD_ROWS_CASE(4, 12) PYZ_HEAD_ROWS_CASE(4, 16)
    PYZ_HEAD_ROWS_CASE(4, 24) PYZ_HEAD_ROWS_CASE(4, 32)
    PYZ_HEAD_ROWS_CASE(8, 4) PYZ_HEAD_ROWS_CASE(8, 8) PYZ_HEAD_ROWS_CASE(8, 12) PYZ_HEAD_ROWS_CASE(8, 16)
    PYZ_HEAD_ROWS_CASE(16, 4) PYZ_HEAD_ROWS_CASE(16, 8)
#undef PYZ_HEAD_ROWS_CASE
  }
  g.nblk = cdiv(grid_batch, 32);
  m->cur_nblk = g.nblk;
  static const int head_waves = pyz_env_int("PYZ_HEAD_WAVES", 8) >= 8 ? 8 : 4;
  PYZ_LAUNCH(k_head, dim3(g.nblk, P), dim3(64 * head_waves), head_waves * 4096 + 2 * 32 * 33 * 4 + 64, st, g);
}

// data gradients of layers L-2 .. 1 (the head already produced delta[L-2])
void launch_bwd_data_hidden(pyz_mlp *m, const float *theta, long long theta_ps, int P, int grid_batch,
                            const StepCtl *ctl, hipStream_t st) {
  for (int l = m->L - 2; l >= 1; --l) {
    const int K = m->dims[l], N = m->dims[l + 1];
    DenseArgs g{};
    g.K = K;
    g.N = N;
    g.in = m->delta[l];
    g.in_pstride = (long long)m->max_batch * N;
    g.lda = N;
    g.theta = theta;
    g.theta_pstride = theta_ps;
    g.w_off = m->w_off[l];
    g.out = m->delta[l - 1];
    g.out_pstride = (long long)m->max_batch * K;
    g.aux = m->act[l - 1];
    g.aux_pstride = (long long)m->max_batch * K;
    g.act = m->acts[l - 1];
    g.vec = (N % 8 == 0) && (m->w_off[l] % 4 == 0) && (P == 1 || theta_ps % 4 == 0) && aligned16(theta) ? 1 : 0;
    g.ctl = ctl;
    pyz_launch_bwd_data(g, grid_batch, P, st);
  }
}

inline int wgrad_tiles(const pyz_mlp *m) {   // workgroups of k_wgrad_all that own a 32 x 32 tile (launch_wgrad_all's count)
  int tiles = 0;
  for (int l = 0; l < m->L; ++l) tiles += ((m->dims[l] + 1 + 31) / 32) * ((m->dims[l + 1] + 31) / 32);
  return tiles;
}

// every layer's weight gradient in one launch; `a` carries the update mode and its buffers
void launch_wgrad_all(pyz_mlp *m, int P, const float *x, const int32_t *row_idx, int grid_batch, const StepCtl *ctl,
                      WgradArgs &a, hipStream_t st, const float *gathered) {
  int tiles = 0;
  a.L = m->L;
  for (int l = 0; l < m->L; ++l) {
    WgradLayer &ly = a.lay[l];
    ly.K = m->dims[l];
    ly.N = m->dims[l + 1];
    if (l == 0) {
      ly.in = (row_idx && gathered) ? gathered : x;  // the forward pass left a contiguous copy of the batch rows
      ly.in_pstride = 0;
      ly.gather = (row_idx && !gathered) ? 1 : 0;
    } else {
      ly.in = m->act[l - 1];
      ly.in_pstride = (long long)m->max_batch * ly.K;
      ly.gather = 0;
    }
    ly.lda = ly.K;
    ly.delta = m->delta[l];
    ly.delta_pstride = (long long)m->max_batch * ly.N;
    ly.w_off = m->w_off[l];
    ly.tile0 = tiles;
    tiles += ((ly.K + 1 + 31) / 32) * ((ly.N + 31) / 32);
  }
  a.row_idx = row_idx;
  a.ctl = ctl;
  a.part = m->part;
  a.nblk = m->cur_nblk;
  a.tiles = tiles;
  a.nonfinite = m->nonfinite;
  a.wt = pyz_wt_for(P);
  const int S = pyz_pick_waves((long long)tiles * P, (grid_batch + 1) / 2);
  dim3 grid(pyz_pad8((long long)tiles + 1, P), P);  // + the duties workgroup (+ padding, see pyz_pad8)
  if (a.prep.src && P == 1) grid.x = (unsigned)(tiles + 1 + std::max(pyz_cu_count() - tiles - 1, 24));  // + the batch workers
  switch (S) {
    case 1:
      if (a.mode == PYZ_UPD_NONE) PYZ_LAUNCH((k_wgrad_all<1, true>), grid, dim3(64), 0, st, a);
      else PYZ_LAUNCH(k_wgrad_all<1>, grid, dim3(64), 0, st, a);
      break;
    case 2: PYZ_LAUNCH(k_wgrad_all<2>, grid, dim3(128), 2 * 4096, st, a); break;
    case 4: PYZ_LAUNCH(k_wgrad_all<4>, grid, dim3(256), 4 * 4096, st, a); break;
    case 8: PYZ_LAUNCH(k_wgrad_all<8>, grid, dim3(512), 8 * 4096, st, a); break;
    default: PYZ_LAUNCH(k_wgrad_all<16>, grid, dim3(1024), 16 * 4096, st, a); break;
  }
}

inline float *batch_buf(pyz_mlp *m, int slot) { return m->xb + (size_t)slot * m->max_batch * m->dims[0]; }

// chained runs whose batches are assembled one step ahead (PrepArgs, pyz_fused.h): single chain, fused path, a hidden layer
// ... and a weight-gradient launch that leaves CUs idle for the copy: with more tiles than CUs (C4: 507) the workers only
// take CUs from the tiles (72.6 against 71.7 us per step at C4; C2, 182 tiles: 24.4 against 26.7 the other way).
// PYZ_BATCH_AHEAD=2 forces it on for every shape.
inline bool batch_ahead(const pyz_mlp *m, const int32_t *row_idx) {
  static const int on = pyz_env_int("PYZ_BATCH_AHEAD", 1);
  if (!on || !can_fuse(m) || m->L <= 1 || row_idx == nullptr) return false;
  return on == 2 || wgrad_tiles(m) + 24 <= pyz_cu_count();
}

PrepArgs prep_args(pyz_mlp *m, const float *x, const int32_t *row_idx, int grid_batch, long long row_stride, int dst_slot) {
  PrepArgs p{};
  p.src = x;
  p.dst = batch_buf(m, dst_slot);
  p.row_idx = row_idx;
  p.tab_bs = m->tab_bs;
  p.row_stride = row_stride;
  p.K = p.lda = m->dims[0];
  p.fwd_tiles_n = (m->dims[1] + 31) / 32;
  p.fwd_tiles = ((grid_batch + 31) / 32) * p.fwd_tiles_n;
  return p;
}

// forward + loss (+ backward into `grad` when upd.mode == NONE and grad given, or the fused update).
// ahead_slot >= 0 (chained runs): this step's rows already stand in batch_buf(ahead_slot), and upd.prep says where the
// weight-gradient launch assembles the next step's.
// split-reduction forward (k_dense_fwd_ring, k_split) + partial-summing head for this launch?  Returns the split (0: no).
int ksplit_for(pyz_mlp *m, const float *theta, long long theta_ps, int P, const float *xc, int grid_batch, const StepCtl *ctl) {
  // Opt-in (read when a run's graph is captured): measured SLOWER at C2 -- 27.5 us per step with 8 ranges, 30.3 with 4, against
  // 24.6 (the forward ~10.7 us against 8.5: seven slabs do not amortise the ring's prologue and the 6.5 MB of partial sums;
  // the head 6.6 against 4.6 us: 32 loads per lane instead of 4) -- profiles/r03_ksplit/.
  const int ks_env = pyz_env_int("PYZ_FWD_KSPLIT", 0);
  if (ks_env < 2 || ks_env > 8 || m->L != 2 || P != 1) return 0;
  const int UT = m->dims[1] <= 64 ? 1 : m->dims[1] <= 256 ? 4 : 0, NPc = m->dims[2];
  if (!UT || NPc > 16 || !pyz_env_int("PYZ_HEAD_ROWS", 1)) return 0;          // the head must be k_head_rows
  DenseArgs g = forward_args(m, 0, theta, theta_ps, xc, nullptr, ctl, nullptr);
  if (!pyz_fwd_ring_ksplit_ok(g, grid_batch, P)) return 0;
  if ((long long)((grid_batch + 31) / 32) * ks_env > 2 * pyz_cu_count()) return 0;   // (already enough row blocks)
  // (the buffer of partial sums is allocated with the plan: this runs inside stream capture, where nothing may be allocated)
  const size_t need = sizeof(float) * (size_t)ks_env * m->max_batch * m->dims[1];
  if (!full(m)->x.fwd_part || full(m)->x.fwd_part_cap < need) return 0;
  return ks_env;
}

void launch_loss_backward(pyz_mlp *m, const float *theta, long long theta_ps, int P, const float *x, const void *y,
                          const int32_t *row_idx, int grid_batch, const StepCtl *ctl, bool want_grad, WgradArgs &upd,
                          hipStream_t st, int ahead_slot = -1) {
  if (can_fuse(m)) {
    if (ahead_slot >= 0) {
      const float *xc = batch_buf(m, ahead_slot);
      // one hidden layer of <= 200 units, one chain: the forward with its reduction split over PYZ_FWD_KSPLIT workgroups per
      // row block, the head adds the partial sums (see pyz_gemm_ring.h)
      const int ks = ksplit_for(m, theta, theta_ps, P, xc, grid_batch, ctl);
      if (ks > 1) {
        DenseArgs g = forward_args(m, 0, theta, theta_ps, xc, nullptr, ctl, nullptr);
        g.wt = pyz_wt_for(P);
        g.k_split = ks;
        g.part_out = static_cast<float *>(full(m)->x.fwd_part);
        g.part_stride = (long long)m->max_batch * g.N;
        pyz_launch_fwd_ring_ksplit(g, grid_batch, st);
      } else {
        launch_forward(m, theta, theta_ps, P, xc, nullptr, grid_batch, ctl, st, nullptr, m->L - 1);
      }
      launch_head(m, theta, theta_ps, P, x, y, row_idx, grid_batch, ctl, want_grad, st, nullptr, 0, ks);   // (labels go through row_idx)
      launch_bwd_data_hidden(m, theta, theta_ps, P, grid_batch, ctl, st);
      launch_wgrad_all(m, P, x, row_idx, grid_batch, ctl, upd, st, xc);
      return;
    }
    static const int use_xb = pyz_env_int("PYZ_GATHER_COPY", 1);  // 1: forward leaves a contiguous batch copy
    float *xb = (use_xb && want_grad && row_idx && m->L > 1) ? m->xb : nullptr;
    launch_forward(m, theta, theta_ps, P, x, row_idx, grid_batch, ctl, st, xb, m->L - 1);   // hidden layers
    launch_head(m, theta, theta_ps, P, x, y, row_idx, grid_batch, ctl, want_grad, st);
    if (want_grad) {
      launch_bwd_data_hidden(m, theta, theta_ps, P, grid_batch, ctl, st);
      launch_wgrad_all(m, P, x, row_idx, grid_batch, ctl, upd, st, xb);
    }
    return;
  }
  flush_ctl(m, st);
  launch_forward(m, theta, theta_ps, P, x, row_idx, grid_batch, ctl, st);
  launch_loss(m, P, y, row_idx, grid_batch, ctl, want_grad, st);
  m->cur_nblk = loss_nblk(m, grid_batch);
  if (want_grad) launch_backward(m, theta, theta_ps, P, x, row_idx, grid_batch, ctl, upd.grad, st);
}

int check_loss_combo(const pyz_mlp *m) {
  const int last = m->acts[m->L - 1];
  if (m->loss == PYZ_LOSS_SCCE && last != PYZ_ACT_SOFTMAX)
    return pyz_fail(PYZ_E_INVALID, "SparseCategoricalCrossentropy needs a softmax last layer");
  if (m->loss == PYZ_LOSS_MSE && last == PYZ_ACT_SOFTMAX)
    return pyz_fail(PYZ_E_INVALID, "MeanSquaredError on a softmax last layer is not supported");
  return PYZ_OK;
}

}  // namespace

extern "C" {

int pyz_version(void) { return PYZ_VERSION; }

const char *pyz_last_error(void) { return pyz_err_slot().c_str(); }

int pyz_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int pyz_mlp_create(int n_layers, const int32_t *h_dims, const int32_t *h_acts, int loss, int max_batch,
                   int max_particles, pyz_mlp **out) {
  if (!out || !h_dims || !h_acts) return pyz_fail(PYZ_E_INVALID, "null argument");
  *out = nullptr;
  if (n_layers < 1 || n_layers > PYZ_MAX_LAYERS) return pyz_fail(PYZ_E_INVALID, "n_layers %d outside [1, %d]", n_layers, PYZ_MAX_LAYERS);
  if (max_batch < 1 || max_particles < 1) return pyz_fail(PYZ_E_INVALID, "max_batch and max_particles must be >= 1");
  if (max_particles > 65535) return pyz_fail(PYZ_E_INVALID, "max_particles %d > 65535", max_particles);
  if (loss != PYZ_LOSS_SCCE && loss != PYZ_LOSS_MSE) return pyz_fail(PYZ_E_INVALID, "unknown loss %d", loss);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return pyz_fail(PYZ_E_NODEV, "no HIP device visible");
  pyz_mlp_full *m = new pyz_mlp_full();
  m->L = n_layers;
  m->loss = loss;
  m->max_batch = max_batch;
  m->max_p = max_particles;
  long long off = 0;
  for (int l = 0; l <= n_layers; ++l) {
    if (h_dims[l] < 1) {
      delete m;
      return pyz_fail(PYZ_E_INVALID, "dims[%d] = %d", l, h_dims[l]);
    }
    m->dims[l] = h_dims[l];
  }
  for (int l = 0; l < n_layers; ++l) {
    const int a = h_acts[l];
    if (a < PYZ_ACT_LINEAR || a > PYZ_ACT_SOFTMAX || (a == PYZ_ACT_SOFTMAX && l != n_layers - 1)) {
      delete m;
      return pyz_fail(PYZ_E_INVALID, "activation %d at layer %d is not supported", a, l);
    }
    m->acts[l] = a;
    m->w_off[l] = off;
    off += (long long)(m->dims[l] + 1) * m->dims[l + 1];
  }
  m->D = off;
  {  // the kernels address operands with 32-bit byte offsets from per-layer bases (buffer descriptors)
    long long widest = 0;
    for (int l = 0; l <= n_layers; ++l) widest = std::max<long long>(widest, m->dims[l]);
    if ((long long)max_batch * widest * 4 >= (1LL << 31) || off * 4 >= (1LL << 31)) {
      delete m;
      return pyz_fail(PYZ_E_INVALID, "max_batch x widest layer, or the parameter vector, exceeds 2 GiB");
    }
  }
  auto fail = [&](int rc) {
    pyz_mlp_destroy(m);
    return rc;
  };
  for (int l = 0; l < n_layers; ++l) {
    const size_t bytes = sizeof(float) * (size_t)max_particles * max_batch * m->dims[l + 1] + 64;
    if (hipMalloc((void **)&m->act[l], bytes) != hipSuccess || hipMalloc((void **)&m->delta[l], bytes) != hipSuccess)
      return fail(pyz_fail(PYZ_E_OOM, "workspace allocation of %zu bytes failed", bytes));
    m->ws_bytes += 2 * bytes;
  }
  {
    // two contiguous batches: chained runs assemble step s + 1's rows in the one that step s does not read
    const size_t bytes = 2 * sizeof(float) * (size_t)max_batch * m->dims[0] + 64;
    if (hipMalloc((void **)&m->xb, bytes) != hipSuccess) return fail(pyz_fail(PYZ_E_OOM, "workspace allocation of %zu bytes failed", bytes));
    m->ws_bytes += bytes;
  }
  if (hipMalloc((void **)&m->nonfinite, 64) != hipSuccess || hipMemset(m->nonfinite, 0, 64) != hipSuccess)
    return fail(pyz_fail(PYZ_E_OOM, "counter allocation failed"));
  if (hipMalloc((void **)&m->ctl, 2 * sizeof(StepCtl)) != hipSuccess) return fail(pyz_fail(PYZ_E_OOM, "ctl allocation failed"));
  if (hipMemset(m->ctl, 0, 2 * sizeof(StepCtl)) != hipSuccess) return fail(pyz_fail(PYZ_E_HIP, "ctl memset failed"));
  const size_t scal = sizeof(float) * (size_t)(max_particles * 16 + 64);
  if (hipMalloc((void **)&m->scal, scal) != hipSuccess) return fail(pyz_fail(PYZ_E_OOM, "scalar allocation failed"));
  m->ws_bytes += scal;
  {
    const int rc = need_part(m, (size_t)max_particles * max_batch + 8);  // up to one loss partial per row (k_head_rows)
    if (rc != PYZ_OK) return fail(rc);
  }
  // a single chain's one hidden layer of <= 200 units (C2): room for the partial sums of the split-reduction forward
  // (only when the opt-in is set when the plan is created: PYZ_FWD_KSPLIT >= 2)
  if (m->L == 2 && m->dims[1] > 192 && m->dims[1] <= 200 && pyz_env_int("PYZ_FWD_KSPLIT", 0) >= 2) {
    const int rc = ensure_bytes(&m->x.fwd_part, &m->x.fwd_part_cap, sizeof(float) * (size_t)8 * max_batch * m->dims[1], m);
    if (rc != PYZ_OK) return fail(rc);
  }
  // kernels with more than 64 KB of dynamic LDS: the allowance is a per-device attribute of the function, set here for the
  // device this plan lives on (one process may drive several devices, each through plans of its own)
  {
    const int gs_lds = (int)pyz_svgd_gs_lds_bytes();
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_svgd_gs<false>), hipFuncAttributeMaxDynamicSharedMemorySize, gs_lds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_svgd_gs<true>), hipFuncAttributeMaxDynamicSharedMemorySize, gs_lds) != hipSuccess)
      return fail(pyz_fail(PYZ_E_HIP, "hipFuncSetAttribute(k_svgd_gs) failed"));
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_svgd_gs_resident), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)pyz_svgd_gs_res_lds_bytes()) != hipSuccess)
      return fail(pyz_fail(PYZ_E_HIP, "hipFuncSetAttribute(k_svgd_gs_resident) failed"));
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_svgd_gs_resident2), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)pyz_svgd_gs_res_lds_bytes()) != hipSuccess)
      return fail(pyz_fail(PYZ_E_HIP, "hipFuncSetAttribute(k_svgd_gs_resident2) failed"));
    const int gr_lds = (int)pyz_svgd_gram_lds_bytes();
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_svgd_gram_tile<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, gr_lds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_svgd_gram_tile<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, gr_lds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_svgd_gram_tile<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, gr_lds) != hipSuccess)
      return fail(pyz_fail(PYZ_E_HIP, "hipFuncSetAttribute(k_svgd_gram_tile) failed"));
  }
  *out = m;
  return PYZ_OK;
}

int pyz_mlp_destroy(pyz_mlp *mm) {
  if (!mm) return PYZ_OK;
  pyz_mlp_full *m = full(mm);
  drop_graphs(m);
  if (m->x.hm_exec) (void)hipGraphExecDestroy(m->x.hm_exec);
  if (m->x.hm_graph) (void)hipGraphDestroy(m->x.hm_graph);
  for (int l = 0; l < PYZ_MAX_LAYERS; ++l) {
    if (m->act[l]) (void)hipFree(m->act[l]);
    if (m->delta[l]) (void)hipFree(m->delta[l]);
  }
  void *ptrs[] = {m->grad, m->grad2, m->qsave, m->part, m->x.part2, m->scal, m->ctl, m->tab_bs, m->xb, m->x.hm_buf, m->x.hm_res_buf, m->x.gs_res_buf, m->x.fwd_part, m->nonfinite};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  if (m->x.tab_host) {
    (void)hipHostFree(m->x.tab_host);
    for (auto &e : m->x.tab_ev) (void)hipEventDestroy(e);
  }
  if (m->x.up_host) {
    (void)hipHostFree(m->x.up_host);
    for (auto &e : m->x.up_ev) (void)hipEventDestroy(e);
  }
  if (m->h_pinned) (void)hipHostFree(m->h_pinned);
  delete m;
  return PYZ_OK;
}

int64_t pyz_mlp_param_count(const pyz_mlp *m) { return m ? m->D : -1; }
int64_t pyz_mlp_workspace_bytes(const pyz_mlp *m) { return m ? (int64_t)m->ws_bytes : -1; }

// ---------------------------------------------------------------- G1
int pyz_mlp_forward(pyz_mlp *m, const float *d_theta, int P, const float *d_x, const int32_t *d_row_idx, int batch,
                    float *d_out, void *stream) {
  int rc = check_call(m, P, batch);
  if (rc) return rc;
  if (!d_theta || !d_x || !d_out) return pyz_fail(PYZ_E_INVALID, "null device pointer");
  hipStream_t st = as_stream(stream);
  if ((rc = set_ctl(m, 0, batch, 0.0f, 0, 0, 0, st))) return rc;
  launch_forward(m, d_theta, m->D, P, d_x, d_row_idx, batch, m->ctl, st);
  const int C = m->dims[m->L];
  PYZ_LAUNCH(k_forward_finish, dim3(cdiv(batch, 256), P), dim3(256), 0, st, m->act[m->L - 1],
                     (long long)m->max_batch * C, C, m->acts[m->L - 1] == PYZ_ACT_SOFTMAX ? 1 : 0, batch, d_out);
  PYZ_LAUNCH_CHECK();
  return PYZ_OK;
}

// ---------------------------------------------------------------- G1-G3
int pyz_mlp_loss_grad(pyz_mlp *m, const float *d_theta, int P, const float *d_x, const void *d_y,
                      const int32_t *d_row_idx, int batch, float *d_grad, float *d_loss, void *stream) {
  int rc = check_call(m, P, batch);
  if (rc) return rc;
  if ((rc = check_loss_combo(m))) return rc;
  if (!d_theta || !d_x || !d_y || !d_loss) return pyz_fail(PYZ_E_INVALID, "null device pointer");
  hipStream_t st = as_stream(stream);
  set_ctl_lazy(m, batch, 0.0f, 0);
  WgradArgs u{};
  u.mode = PYZ_UPD_NONE;
  u.grad = d_grad;
  u.grad_pstride = m->D;
  launch_loss_backward(m, d_theta, m->D, P, d_x, d_y, d_row_idx, batch, m->ctl, d_grad != nullptr, u, st);
  PYZ_LAUNCH(k_loss_finalize, dim3(P), dim3(64), 0, st, m->part, m->cur_nblk, m->ctl, d_loss, m->nonfinite);
  PYZ_LAUNCH_CHECK();
  return PYZ_OK;
}

// ---------------------------------------------------------------- S1
int pyz_sgd_step(pyz_mlp *m, float *d_theta, const float *d_x, const void *d_y, const int32_t *d_row_idx, int batch,
                 float lr, float *d_loss, void *stream) {
  int rc = check_call(m, 1, batch);
  if (rc) return rc;
  if ((rc = check_loss_combo(m))) return rc;
  if (!d_theta || !d_x || !d_y || !d_loss) return pyz_fail(PYZ_E_INVALID, "null device pointer");
  const bool fused = can_fuse(m);
  if (!fused && (rc = need_grad(m, 1))) return rc;
  hipStream_t st = as_stream(stream);
  set_ctl_lazy(m, batch, lr, 0);
  WgradArgs u{};
  if (fused) {  // the update runs in the epilogue of the weight-gradient kernel
    u.mode = PYZ_UPD_SGD;
    u.theta = d_theta;
    u.loss = d_loss;
  } else {
    u.mode = PYZ_UPD_NONE;
    u.grad = m->grad;
    u.grad_pstride = m->D;
  }
  launch_loss_backward(m, d_theta, m->D, 1, d_x, d_y, d_row_idx, batch, m->ctl, true, u, st);
  if (!fused)
    PYZ_LAUNCH(k_sgd_update, dim3(cdiv(m->D, 256)), dim3(256), 0, st, d_theta, m->grad, m->D, m->ctl, m->part,
                       m->cur_nblk, d_loss, m->nonfinite);
  PYZ_LAUNCH_CHECK();
  return PYZ_OK;
}

// ---------------------------------------------------------------- SWAG (survey 8f, rank 2)
int pyz_swag_step(pyz_mlp *m, float *d_theta, float *d_mean, float *d_sq_mean, float *d_dev_row, const float *d_x,
                  const void *d_y, const int32_t *d_row_idx, int batch, float lr, int64_t n, int update_moments,
                  float *d_loss, void *stream) {
  int rc = check_call(m, 1, batch);
  if (rc) return rc;
  if ((rc = check_loss_combo(m))) return rc;
  if (!d_theta || !d_mean || !d_sq_mean || !d_x || !d_y || !d_loss) return pyz_fail(PYZ_E_INVALID, "null device pointer");
  if (n < 0) return pyz_fail(PYZ_E_INVALID, "negative step count");
  const bool fused = can_fuse(m);
  if (!fused && (rc = need_grad(m, 1))) return rc;
  hipStream_t st = as_stream(stream);
  set_ctl_lazy(m, batch, lr, n);
  WgradArgs u{};
  if (fused) {
    u.mode = PYZ_UPD_SWAG;
    u.theta = d_theta;
    u.mean = d_mean;
    u.sq_mean = d_sq_mean;
    u.dev_row = d_dev_row;
    u.swag_update = update_moments ? 1 : 0;
    u.loss = d_loss;
  } else {
    u.mode = PYZ_UPD_NONE;
    u.grad = m->grad;
    u.grad_pstride = m->D;
  }
  launch_loss_backward(m, d_theta, m->D, 1, d_x, d_y, d_row_idx, batch, m->ctl, true, u, st);
  if (!fused)
    PYZ_LAUNCH(k_swag_update, dim3(cdiv(m->D, 256)), dim3(256), 0, st, d_theta, d_mean, d_sq_mean, d_dev_row,
                       m->grad, m->D, update_moments ? 1 : 0, m->ctl, m->part, m->cur_nblk, d_loss, m->nonfinite);
  PYZ_LAUNCH_CHECK();
  return PYZ_OK;
}

// ---------------------------------------------------------------- L2/L3
struct SwagChain {  // chained SWAG run: the (k, D) deviation matrix and the two hyper-parameters its bookkeeping needs
  float *dev;
  int k, freq;
};

struct BbbChain {  // chained BBB run: what the step needs beside mu (= theta), and the optional validation forward
  float *rho, *w;
  float *w2;                // sampling fused into the weight-gradient epilogue: the second weight buffer (steps on StepCtl slot 1), or nullptr
  float alpha, prior_mean, prior_rho;
  const float *pm_vec, *pr_vec;
  pyz_mlp *val;             // plan of the validation forward (max_batch >= n_val), or nullptr
  const float *val_x;
  const void *val_y;
  int n_val;
  float *val_losses;        // [slot] per step; written on the steps that validate (BBB.py:203: step % 10 != 0)
};

static void launch_sgld_step(pyz_mlp *m, float *theta, float *mean, float *sq, const float *x, const void *y,
                             const int32_t *row_idx, int grid_batch, int slot, bool chained, long long row_stride,
                             uint64_t seed, const float *unit_noise, float *loss, hipStream_t st, int mode = PYZ_UPD_SGLD,
                             const SwagChain *swag = nullptr, const BbbChain *bbb = nullptr, bool first = true) {
  const StepCtl *ctl = m->ctl + slot;
  if (bbb) {   // BBB.step (BBB.py:128-211) inside a device-resident run: scalars from StepCtl, cost into slot (slot0 + i)
    // Sampling fused (bbb->w2): the weights and the log q - log p partials of step i + 1 come out of step i's weight-gradient
    // epilogue, into the buffers of the other StepCtl slot; only the first step of a chunk launches k_bbb_sample (for a later
    // chunk it rewrites what the chunk before left: same Philox step, same mu / rho).
    const int nblk_sample = cdiv(cdiv(m->D, 4), 256), tiles = wgrad_tiles(m);
    const bool fuse = bbb->w2 != nullptr;
    const size_t kl_stride = (size_t)std::max(nblk_sample, tiles);
    double *kl_cur = full(m)->x.part2 + (fuse ? (size_t)slot * kl_stride : 0);
    double *kl_next = full(m)->x.part2 + (size_t)(slot ^ 1) * kl_stride;
    float *w_cur = (fuse && slot) ? bbb->w2 : bbb->w, *w_next = slot ? bbb->w : bbb->w2;
    const bool sample = !fuse || first;
    if (sample) {
      BbbArgs a{};
      a.mu = theta;
      a.rho = bbb->rho;
      a.w = w_cur;
      a.D = m->D;
      a.alpha = bbb->alpha;
      a.prior_mean = bbb->prior_mean;
      a.prior_rho = bbb->prior_rho;
      a.pm_vec = bbb->pm_vec;
      a.pr_vec = bbb->pr_vec;
      a.seed = seed;
      a.part_kl = kl_cur;
      a.nblk_kl = nblk_sample;
      a.ctl = ctl;
      a.chained = 1;
      PYZ_LAUNCH(k_bbb_sample, dim3(nblk_sample), dim3(256), 0, st, a);
    }
    WgradArgs u{};
    u.mode = PYZ_UPD_BBB;
    u.theta = theta;
    u.mean = bbb->rho;
    u.sq_mean = w_cur;
    u.seed = seed;
    u.alpha = bbb->alpha;
    u.prior_mean = bbb->prior_mean;
    u.prior_rho = bbb->prior_rho;
    u.pm_vec = bbb->pm_vec;
    u.pr_vec = bbb->pr_vec;
    u.bbb_chained = 1;
    u.part_kl = kl_cur;
    u.nblk_kl = sample ? nblk_sample : tiles;
    if (fuse) {
      u.w_next = w_next;
      u.part_kl_next = kl_next;
    }
    u.cost = loss;
    u.next = chained ? m->ctl + (slot ^ 1) : nullptr;
    u.tab_bs = m->tab_bs;
    u.tab_lr = m->tab_lr;
    u.row_stride = row_stride;
    const bool ahead = chained && batch_ahead(m, row_idx);
    if (ahead) u.prep = prep_args(m, x, row_idx, grid_batch, row_stride, slot ^ 1);
    launch_loss_backward(m, w_cur, m->D, 1, x, y, row_idx, grid_batch, ctl, true, u, st, ahead ? slot : -1);
    if (bbb->val) {   // the validation split through the weights this step sampled; the launches do nothing on every tenth step
      pyz_mlp *v = bbb->val;
      launch_forward(v, w_cur, v->D, 1, bbb->val_x, nullptr, bbb->n_val, v->ctl, st, nullptr, v->L - 1, ctl, 10);
      launch_head(v, w_cur, v->D, 1, bbb->val_x, bbb->val_y, nullptr, bbb->n_val, v->ctl, false, st, ctl, 10);
      PYZ_LAUNCH(k_loss_finalize_gated, dim3(1), dim3(64), 0, st, v->part, v->cur_nblk, v->ctl, bbb->val_losses, v->nonfinite, ctl, 10);
    }
    return;
  }
  if (can_fuse(m)) {
    WgradArgs u{};
    u.mode = mode;  // PYZ_UPD_SGLD, or PYZ_UPD_SGD / PYZ_UPD_SWAG for the chained SGD / SWAG runs (fused path only)
    if (swag) {
      u.swag_freq = swag->freq;
      u.swag_k = swag->k;
      u.dev_base = swag->dev;
      u.dev_stride = m->D;
    }
    u.theta = theta;
    u.mean = mean;
    u.sq_mean = sq;
    u.seed = seed;
    u.unit_noise = unit_noise;
    u.loss = loss;
    u.loss_indexed = chained ? 1 : 0;
    u.next = chained ? m->ctl + (slot ^ 1) : nullptr;
    u.tab_bs = m->tab_bs;
    u.tab_lr = m->tab_lr;
    u.row_stride = row_stride;
    const bool ahead = chained && batch_ahead(m, row_idx);
    if (ahead) u.prep = prep_args(m, x, row_idx, grid_batch, row_stride, slot ^ 1);
    launch_loss_backward(m, theta, m->D, 1, x, y, row_idx, grid_batch, ctl, true, u, st, ahead ? slot : -1);
    return;
  }
  WgradArgs u{};
  u.mode = PYZ_UPD_NONE;
  u.grad = m->grad;
  u.grad_pstride = m->D;
  launch_loss_backward(m, theta, m->D, 1, x, y, row_idx, grid_batch, ctl, true, u, st);
  SgldArgs a{};
  a.theta = theta;
  a.mean = mean;
  a.sq_mean = sq;
  a.grad = m->grad;
  a.D = m->D;
  a.ctl = ctl;
  a.next = chained ? m->ctl + (slot ^ 1) : nullptr;
  a.tab_bs = m->tab_bs;
  a.tab_lr = m->tab_lr;
  a.row_stride = row_stride;
  a.seed = seed;
  a.unit_noise = unit_noise;
  a.part = m->part;
  a.nblk = m->cur_nblk;
  a.loss = loss;
  a.loss_indexed = chained ? 1 : 0;
  a.nonfinite = m->nonfinite;
  PYZ_LAUNCH(k_sgld_update, dim3(cdiv(cdiv(m->D, 4), 256)), dim3(256), 0, st, a);
}

int pyz_sgld_step(pyz_mlp *m, float *d_theta, float *d_mean, float *d_sq_mean, const float *d_x, const void *d_y,
                  const int32_t *d_row_idx, int batch, float lr, int64_t n, uint64_t seed, const float *d_unit_noise,
                  float *d_loss, void *stream) {
  int rc = check_call(m, 1, batch);
  if (rc) return rc;
  if ((rc = check_loss_combo(m))) return rc;
  if (!d_theta || !d_mean || !d_sq_mean || !d_x || !d_y || !d_loss) return pyz_fail(PYZ_E_INVALID, "null device pointer");
  if (n < 0) return pyz_fail(PYZ_E_INVALID, "negative step count");
  if (!can_fuse(m) && (rc = need_grad(m, 1))) return rc;
  hipStream_t st = as_stream(stream);
  set_ctl_lazy(m, batch, lr, n);
  launch_sgld_step(m, d_theta, d_mean, d_sq_mean, d_x, d_y, d_row_idx, batch, 0, false, 0, seed, d_unit_noise, d_loss, st);
  PYZ_LAUNCH_CHECK();
  return PYZ_OK;
}

static int sgld_run_impl(pyz_mlp *m, float *d_theta, float *d_mean, float *d_sq_mean, const float *d_x, const void *d_y,
                         const int32_t *d_row_idx, const int32_t *h_batch_sizes, const float *h_lr, int n_steps,
                         int64_t n0, int64_t slot0, uint64_t seed, float *d_losses, int use_graph, void *stream,
                         int mode = PYZ_UPD_SGLD, const SwagChain *swag = nullptr, const BbbChain *bbb = nullptr) {
  if (!m) return pyz_fail(PYZ_E_INVALID, "null plan");
  if (mode != PYZ_UPD_SGLD && !can_fuse(m))
    return pyz_fail(PYZ_E_INVALID, "the chained SGD / SWAG runs need a last layer of at most 32 units");
  if (mode == PYZ_UPD_SGD) d_mean = d_sq_mean = d_theta;  // not touched in this mode
  int rc = check_loss_combo(m);
  if (rc) return rc;
  if (n_steps <= 0) return pyz_fail(PYZ_E_INVALID, "n_steps must be positive");
  if (slot0 < 0 || slot0 + n_steps > 0x7fffffff) return pyz_fail(PYZ_E_INVALID, "slot0 out of range");
  if (!d_theta || !d_mean || !d_sq_mean || !d_x || !d_y || !d_row_idx || !h_batch_sizes || !h_lr || !d_losses)
    return pyz_fail(PYZ_E_INVALID, "null pointer");
  int bmax = 0;
  for (int s = 0; s < n_steps; ++s) {
    if (h_batch_sizes[s] <= 0 || h_batch_sizes[s] > m->max_batch)
      return pyz_fail(PYZ_E_SHAPE, "batch size %d of step %d outside [1, %d]", h_batch_sizes[s], s, m->max_batch);
    bmax = std::max(bmax, h_batch_sizes[s]);
  }
  if (!can_fuse(m) && (rc = need_grad(m, 1))) return rc;
  pyz_mlp_full *f = full(m);
  hipStream_t st = as_stream(stream);
  BbbChain bbb_local{};
  if (bbb) {
    static const int fuse_sample = pyz_env_int("PYZ_BBB_FUSE_SAMPLE", 1);
    if ((rc = need_part2(m, 2 * (size_t)std::max(cdiv(cdiv(m->D, 4), 256), wgrad_tiles(m))))) return rc;
    if (fuse_sample) {   // the plan's gradient buffer (idle on the fused path) is the second weight buffer
      if ((rc = need_grad(m, 1))) return rc;
      bbb_local = *bbb;
      bbb_local.w2 = m->grad;
      bbb = &bbb_local;
    }
    if (bbb->val) {   // the validation plan's step scalars: its batch is the whole split, for every step of the run
      pyz_mlp *v = bbb->val;
      if (v->D != m->D || !can_fuse(v) || bbb->n_val < 1 || bbb->n_val > v->max_batch || !bbb->val_x || !bbb->val_y || !bbb->val_losses)
        return pyz_fail(PYZ_E_INVALID, "validation plan / split does not fit the model");
      if ((rc = set_ctl(v, 0, bbb->n_val, 0.0f, 0, 0, 0, st))) return rc;
    }
  }
  // per-run tables (one padding entry: the last step prepares a slot nobody reads); batch sizes and learning
  // rates share one device allocation
  const size_t n_tab = (size_t)n_steps + 1;
  if (m->tab_cap < (int)n_tab) {
    const size_t cap = std::max<size_t>(n_tab, 65536);  // generous: the table pointers are baked into the graphs
    if (m->tab_bs) PYZ_HIP(hipFree(m->tab_bs));
    m->tab_bs = nullptr;
    m->tab_lr = nullptr;
    PYZ_HIP(hipMalloc((void **)&m->tab_bs, 8 * cap));
    m->tab_lr = reinterpret_cast<float *>(m->tab_bs + cap);
    m->tab_cap = (int)cap;
    drop_graphs(m);
  }
  // the tables go up through two pinned buffers used in turn: a buffer is rewritten once the copy of the run
  // before the previous one has left it (no stream-wide synchronisation per call)
  if (f->x.tab_host_cap < 8 * n_tab) {
    if (f->x.tab_host) {
      PYZ_HIP(hipStreamSynchronize(st));
      PYZ_HIP(hipHostFree(f->x.tab_host));
      f->x.tab_host = nullptr;
    } else {
      for (auto &e : f->x.tab_ev) PYZ_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const size_t cap = std::max<size_t>(8 * n_tab, 8 * 4096);
    PYZ_HIP(hipHostMalloc(&f->x.tab_host, 2 * cap));
    f->x.tab_host_cap = cap;
    f->x.tab_count = 0;
  }
  const long long row_stride = m->max_batch;
  bool first_batch_done = false;
  if (n_steps <= PYZ_INLINE_TAB) {
    // short run: the tables ride in the arguments of the launch that sets the first step's scalars
    InlineTabs tabs{};
    for (int s = 0; s < n_steps; ++s) {
      tabs.bs[s] = h_batch_sizes[s];
      tabs.lr[s] = h_lr[s];
    }
    tabs.bs[n_steps] = h_batch_sizes[n_steps - 1];
    tabs.lr[n_steps] = h_lr[n_steps - 1];
    tabs.n = n_steps + 1;
    m->pend_on = false;
    if (batch_ahead(m, d_row_idx)) {   // ... together with the first batch of the run
      PYZ_LAUNCH(k_run_start, dim3(256), dim3(256), 0, st, m->ctl, tabs, m->tab_bs, m->tab_lr, (long long)n0, slot0 * row_stride,
                 (int)slot0, prep_args(m, d_x, d_row_idx, bmax, row_stride, 0));
      first_batch_done = true;
    } else {
      PYZ_LAUNCH(k_set_ctl_tabs, dim3(1), dim3(64), 0, st, m->ctl, tabs, m->tab_bs, m->tab_lr, (long long)n0, slot0 * row_stride, (int)slot0);
    }
  } else {
    const unsigned slot = (unsigned)(f->x.tab_count & 1);
    if (f->x.tab_count >= 2) PYZ_HIP(hipEventSynchronize(f->x.tab_ev[slot]));
    ++f->x.tab_count;
    int32_t *hb = reinterpret_cast<int32_t *>(static_cast<unsigned char *>(f->x.tab_host) + slot * f->x.tab_host_cap);
    float *hl = reinterpret_cast<float *>(hb + n_tab);
    std::memcpy(hb, h_batch_sizes, sizeof(int32_t) * n_steps);
    std::memcpy(hl, h_lr, sizeof(float) * n_steps);
    hb[n_steps] = h_batch_sizes[n_steps - 1];
    hl[n_steps] = h_lr[n_steps - 1];
    PYZ_HIP(hipMemcpyAsync(m->tab_bs, hb, sizeof(int32_t) * n_tab, hipMemcpyHostToDevice, st));
    PYZ_HIP(hipMemcpyAsync(m->tab_lr, hl, sizeof(float) * n_tab, hipMemcpyHostToDevice, st));
    PYZ_HIP(hipEventRecord(f->x.tab_ev[slot], st));
    if ((rc = set_ctl(m, 0, h_batch_sizes[0], h_lr[0], n0, slot0 * row_stride, 0, st, (int)slot0, n_steps))) return rc;
  }

  if (batch_ahead(m, d_row_idx) && !first_batch_done)   // the first batch of the run (every later one is assembled by the step before it)
    PYZ_LAUNCH(k_prep_batch, dim3(256), dim3(256), 0, st, prep_args(m, d_x, d_row_idx, bmax, row_stride, 0), m->ctl);

  int s = 0;
  m->run_graph_steps = m->run_eager_steps = m->run_graph_launches = 0;
  // A run of ANY length is replayed from captured graphs: chunks of G steps (slot 0) and ONE graph for the
  // remainder, captured for its exact length and kept (the other slots, least recently used one replaced) -- a
  // caller that repeats its run lengths (train(n) again, bench.py's warm-up / timed pair) pays one capture per
  // length and afterwards one graph launch per chunk.  Every chunk starts on StepCtl slot 0: G is even (the
  // ping-pong returns to slot 0) and the remainder comes last.
  static const int G = std::min(128, std::max(2, pyz_env_int("PYZ_GRAPH_STEPS", 32) & ~1));
  if (use_graph && st != nullptr && !pyz_probe().on) {
    // everything baked into the graphs goes into the key
    unsigned long long key = 1469598103934665603ull;
    auto mix = [&](unsigned long long v) { key = (key ^ v) * 1099511628211ull; };
    mix((unsigned long long)(uintptr_t)d_theta); mix((unsigned long long)(uintptr_t)d_mean);
    mix((unsigned long long)(uintptr_t)d_sq_mean); mix((unsigned long long)(uintptr_t)d_x);
    mix((unsigned long long)(uintptr_t)d_y); mix((unsigned long long)(uintptr_t)d_row_idx);
    mix((unsigned long long)(uintptr_t)d_losses); mix(seed); mix((unsigned long long)bmax);
    mix((unsigned long long)(uintptr_t)m->tab_bs); mix((unsigned long long)G); mix((unsigned long long)mode);
    // (not the stream: an instantiated graph launches on any stream, and a caller that takes a fresh stream per run --
    // torch hands them out of a pool -- would otherwise re-capture every graph on every call)
    if (swag) { mix((unsigned long long)(uintptr_t)swag->dev); mix((unsigned long long)swag->k); mix((unsigned long long)swag->freq); }
    if (bbb) {
      unsigned fb[3];
      memcpy(&fb[0], &bbb->alpha, 4); memcpy(&fb[1], &bbb->prior_mean, 4); memcpy(&fb[2], &bbb->prior_rho, 4);
      for (unsigned b : fb) mix(b);
      mix((unsigned long long)(uintptr_t)bbb->pm_vec); mix((unsigned long long)(uintptr_t)bbb->pr_vec);
      mix((unsigned long long)(uintptr_t)bbb->val); mix((unsigned long long)(uintptr_t)bbb->val_x); mix((unsigned long long)(uintptr_t)bbb->val_y);
      mix((unsigned long long)bbb->n_val); mix((unsigned long long)(uintptr_t)bbb->val_losses);
      mix((unsigned long long)(uintptr_t)bbb->w2); mix((unsigned long long)(uintptr_t)bbb->rho); mix((unsigned long long)(uintptr_t)full(m)->x.part2);
    }
    if (m->graph_key != key) {
      drop_graphs(m);
      m->graph_key = key;
    }
    while (s < n_steps) {
      const int len = std::min(G, n_steps - s);
      int ci = 0;
      if (len < G) {  // the remainder: its own graph, by exact length
        ci = -1;
        int free_slot = -1, lru = -1;
        for (int c = 1; c < PYZ_GRAPH_CHUNKS; ++c) {
          if (!m->graph_exec[c]) {
            if (free_slot < 0) free_slot = c;
            continue;
          }
          if (m->graph_len[c] == len) ci = c;
          if (lru < 0 || m->graph_use[c] < m->graph_use[lru]) lru = c;
        }
        if (ci < 0) {
          ci = free_slot >= 0 ? free_slot : lru;
          if (m->graph_exec[ci]) (void)hipGraphExecDestroy(m->graph_exec[ci]);
          if (m->graph[ci]) (void)hipGraphDestroy(m->graph[ci]);
          m->graph_exec[ci] = nullptr;
          m->graph[ci] = nullptr;
        }
      }
      if (!m->graph_exec[ci]) {
        PYZ_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        for (int k = 0; k < len; ++k)
          launch_sgld_step(m, d_theta, d_mean, d_sq_mean, d_x, d_y, d_row_idx, bmax, k & 1, true, row_stride, seed,
                           nullptr, d_losses, st, mode, swag, bbb, k == 0);
        hipGraph_t gr = nullptr;
        PYZ_HIP(hipStreamEndCapture(st, &gr));
        m->graph[ci] = gr;
        PYZ_HIP(hipGraphInstantiate(&m->graph_exec[ci], gr, nullptr, nullptr, 0));
        m->graph_len[ci] = len;
      }
      m->graph_use[ci] = ++m->graph_clock;
      PYZ_HIP(hipGraphLaunch(m->graph_exec[ci], st));
      s += len;
      m->run_graph_steps += len;
      ++m->run_graph_launches;
    }
  }
  for (; s < n_steps; ++s) {
    launch_sgld_step(m, d_theta, d_mean, d_sq_mean, d_x, d_y, d_row_idx, bmax, s & 1, true, row_stride, seed, nullptr,
                     d_losses, st, mode, swag, bbb, s == 0);
    ++m->run_eager_steps;
  }
  // fused sampling: the weights of the last step stand in the second buffer when it ran on StepCtl slot 1
  if (bbb && bbb->w2 && ((n_steps - 1) & 1))
    PYZ_HIP(hipMemcpyAsync(bbb->w, bbb->w2, sizeof(float) * (size_t)m->D, hipMemcpyDeviceToDevice, st));
  PYZ_LAUNCH_CHECK();
  return PYZ_OK;
}

int pyz_sgld_run(pyz_mlp *m, float *d_theta, float *d_mean, float *d_sq_mean, const float *d_x, const void *d_y,
                 const int32_t *d_row_idx, const int32_t *h_batch_sizes, const float *h_lr, int n_steps, int64_t n0,
                 int64_t slot0, uint64_t seed, float *d_losses, int use_graph, void *stream) {
  return sgld_run_impl(m, d_theta, d_mean, d_sq_mean, d_x, d_y, d_row_idx, h_batch_sizes, h_lr, n_steps, n0, slot0, seed,
                       d_losses, use_graph, stream);
}

// The SGD train loop (SGD.py:42-69 inside Optimizer.py:121-134) as one device-resident run: the launch sequence
// of pyz_sgld_run with the plain update in the weight-gradient epilogue (no noise, no moments).
int pyz_sgd_run(pyz_mlp *m, float *d_theta, const float *d_x, const void *d_y, const int32_t *d_row_idx,
                const int32_t *h_batch_sizes, const float *h_lr, int n_steps, int64_t slot0, float *d_losses, int use_graph,
                void *stream) {
  return sgld_run_impl(m, d_theta, d_theta, d_theta, d_x, d_y, d_row_idx, h_batch_sizes, h_lr, n_steps, 0, slot0, 0, d_losses,
                       use_graph, stream, PYZ_UPD_SGD);
}

// The SWAG train loop (SWAG.py:43-94 inside Optimizer.py:121-134) as one device-resident run.  The moment / deviation
// bookkeeping follows from the step count on the device: it must have started at count 0 with every step run.
int pyz_swag_run(pyz_mlp *m, float *d_theta, float *d_mean, float *d_sq_mean, float *d_dev, int k, int frequency,
                 const float *d_x, const void *d_y, const int32_t *d_row_idx, const int32_t *h_batch_sizes, const float *h_lr,
                 int n_steps, int64_t n0, int64_t slot0, float *d_losses, int use_graph, void *stream) {
  if (!d_dev || k < 1 || frequency < 1) return pyz_fail(PYZ_E_INVALID, "bad deviation matrix / k / frequency");
  if (n0 < 0) return pyz_fail(PYZ_E_INVALID, "negative step count");
  const SwagChain sc{d_dev, k, frequency};
  return sgld_run_impl(m, d_theta, d_mean, d_sq_mean, d_x, d_y, d_row_idx, h_batch_sizes, h_lr, n_steps, n0, slot0, 0, d_losses,
                       use_graph, stream, PYZ_UPD_SWAG, &sc);
}

// ---------------------------------------------------------------- B2-B4
int pyz_bbb_step(pyz_mlp *m, float *d_mu, float *d_rho, float *d_w, const float *d_x, const void *d_y,
                 const int32_t *d_row_idx, int batch, float lr, float alpha, float prior_mean, float prior_rho,
                 const float *d_prior_mean_vec, const float *d_prior_rho_vec, int64_t step, uint64_t seed,
                 const float *d_eps, float *d_cost, void *stream) {
  int rc = check_call(m, 1, batch);
  if (rc) return rc;
  if ((rc = check_loss_combo(m))) return rc;
  if (!d_mu || !d_rho || !d_w || !d_x || !d_y || !d_cost) return pyz_fail(PYZ_E_INVALID, "null device pointer");
  if ((rc = need_grad(m, 1))) return rc;
  const int nblk_kl = cdiv(cdiv(m->D, 4), 256);
  if ((rc = need_part2(m, nblk_kl))) return rc;
  hipStream_t st = as_stream(stream);
  set_ctl_lazy(m, batch, lr, step);
  BbbArgs a{};
  a.mu = d_mu;
  a.rho = d_rho;
  a.w = d_w;
  a.grad = m->grad;
  a.D = m->D;
  a.lr = lr;
  a.alpha = alpha;
  a.prior_mean = prior_mean;
  a.prior_rho = prior_rho;
  a.pm_vec = d_prior_mean_vec;
  a.pr_vec = d_prior_rho_vec;
  a.seed = seed;
  a.step = (uint32_t)step;
  a.eps = d_eps;
  a.part_kl = full(m)->x.part2;
  a.nblk_kl = nblk_kl;
  a.part_loss = m->part;
  a.nblk_loss = loss_nblk(m, batch);
  a.ctl = m->ctl;
  a.cost = d_cost;
  PYZ_LAUNCH(k_bbb_sample, dim3(nblk_kl), dim3(256), 0, st, a);
  WgradArgs u{};
  if (can_fuse(m)) {  // the mu / rho update runs in the epilogue of the weight-gradient kernel
    u.mode = PYZ_UPD_BBB;
    u.theta = d_mu;
    u.mean = d_rho;
    u.sq_mean = d_w;
    u.seed = seed;
    u.unit_noise = d_eps;
    u.alpha = alpha;
    u.prior_mean = prior_mean;
    u.prior_rho = prior_rho;
    u.bbb_lr = lr;
    u.pm_vec = d_prior_mean_vec;
    u.pr_vec = d_prior_rho_vec;
    u.bbb_step = (uint32_t)step;
    u.part_kl = full(m)->x.part2;
    u.nblk_kl = nblk_kl;
    u.cost = d_cost;
    launch_loss_backward(m, d_w, m->D, 1, d_x, d_y, d_row_idx, batch, m->ctl, true, u, st);
  } else {
    u.mode = PYZ_UPD_NONE;
    u.grad = m->grad;
    u.grad_pstride = m->D;
    launch_loss_backward(m, d_w, m->D, 1, d_x, d_y, d_row_idx, batch, m->ctl, true, u, st);
    a.nblk_loss = m->cur_nblk;
    PYZ_LAUNCH(k_bbb_update, dim3(nblk_kl), dim3(256), 0, st, a);
  }
  PYZ_LAUNCH_CHECK();
  return PYZ_OK;
}

// The BBB train loop (BBB.py:128-211 inside Optimizer.py:121-134) as one device-resident run: per step the launch
// sequence of pyz_bbb_step with the step scalars on the device (learning rate from the table, Philox step = step0 + i),
// batches assembled one step ahead, the cost triple of step i in d_costs[4 (slot0 + i) ..], and -- with a validation
// plan -- the validation split forwarded through the weights the step sampled on the steps BBB.py:203 validates
// (step % 10 != 0; the launches are in every step of the graph and return at once on the others).
int pyz_bbb_run(pyz_mlp *m, float *d_mu, float *d_rho, float *d_w, const float *d_x, const void *d_y,
                const int32_t *d_row_idx, const int32_t *h_batch_sizes, const float *h_lr, int n_steps, float alpha,
                float prior_mean, float prior_rho, const float *d_prior_mean_vec, const float *d_prior_rho_vec, int64_t step0,
                int64_t slot0, uint64_t seed, float *d_costs, pyz_mlp *val_plan, const float *d_val_x, const void *d_val_y,
                int n_val, float *d_val_losses, int use_graph, void *stream) {
  if (!d_rho || !d_w) return pyz_fail(PYZ_E_INVALID, "null device pointer");
  if (step0 < 0) return pyz_fail(PYZ_E_INVALID, "negative step count");
  BbbChain bc{};
  bc.rho = d_rho;
  bc.w = d_w;
  bc.alpha = alpha;
  bc.prior_mean = prior_mean;
  bc.prior_rho = prior_rho;
  bc.pm_vec = d_prior_mean_vec;
  bc.pr_vec = d_prior_rho_vec;
  bc.val = val_plan;
  bc.val_x = d_val_x;
  bc.val_y = d_val_y;
  bc.n_val = n_val;
  bc.val_losses = d_val_losses;
  return sgld_run_impl(m, d_mu, d_rho, d_w, d_x, d_y, d_row_idx, h_batch_sizes, h_lr, n_steps, step0, slot0, seed, d_costs,
                       use_graph, stream, PYZ_UPD_BBB, nullptr, &bc);
}

// ---------------------------------------------------------------- H2-H5
int pyz_hmc_step(pyz_mlp *m, float *d_q, int P, const float *d_x, const void *d_y, int n_rows, int L, float epsilon,
                 float mass, float prior_mean, float prior_sigma, const float *d_prior_mean_vec,
                 const float *d_prior_sigma_vec, int burning, const float *h_uniform, int64_t step, uint64_t seed,
                 const float *d_unit_p, float *d_stats, void *stream) {
  int rc = check_call(m, P, n_rows);
  if (rc) return rc;
  if ((rc = check_loss_combo(m))) return rc;
  if (!d_q || !d_x || !d_y || !d_stats || !h_uniform) return pyz_fail(PYZ_E_INVALID, "null pointer");
  if (L < 0) return pyz_fail(PYZ_E_INVALID, "L must be >= 0");
  if ((rc = need_grad(m, P)) || (rc = need_grad2(m, P)) || (rc = need_qsave(m, P))) return rc;   // (also: the gradients of an SVGD phase 1 are gone)
  const int nblk4 = cdiv(cdiv(m->D, 4), 256), nblk1 = cdiv(m->D, 256);
  if ((rc = need_part2(m, (size_t)P * 2 * nblk1 + 8))) return rc;
  hipStream_t st = as_stream(stream);
  pyz_mlp_full *f = full(m);
  float *loss = m->scal;               // [P]
  float *energies = m->scal + m->max_p;  // [P*8]
  float *unif = m->scal + 9 * m->max_p;  // [max_p] uniforms, then one HmcCall
  HmcCall *call_dev = reinterpret_cast<HmcCall *>(m->scal + 10 * m->max_p);  // float index 10 max_p is even
  {
    const size_t up_bytes = sizeof(float) * (size_t)m->max_p + sizeof(HmcCall);
    Extra &x = f->x;
    if (!x.up_host) {
      x.up_slot_bytes = (up_bytes + 63) / 64 * 64;
      PYZ_HIP(hipHostMalloc((void **)&x.up_host, x.up_slot_bytes * Extra::UP_SLOTS));
      for (auto &e : x.up_ev) PYZ_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const unsigned slot = (unsigned)(x.up_count % Extra::UP_SLOTS);
    if (x.up_count >= (unsigned long long)Extra::UP_SLOTS) PYZ_HIP(hipEventSynchronize(x.up_ev[slot]));  // its last copy has left
    ++x.up_count;
    unsigned char *up = x.up_host + x.up_slot_bytes * slot;
    memcpy(up, h_uniform, sizeof(float) * P);
    HmcCall hc{seed, (uint32_t)step, burning ? 1 : 0};
    memcpy(up + sizeof(float) * (size_t)m->max_p, &hc, sizeof hc);
    PYZ_HIP(hipMemcpyAsync(unif, up, up_bytes, hipMemcpyHostToDevice, st));
    PYZ_HIP(hipEventRecord(x.up_ev[slot], st));
  }
  {
    // small 2-layer models: the whole proposal in one workgroup per chain (pyz_hmc_fused.h)
    const int allow_fused = pyz_env_int("PYZ_HMC_FUSED", 1);  // read per call: tests flip it
    const int I = m->dims[0], H = m->dims[1], C = m->L == 2 ? m->dims[2] : 0;
    const int MIC = (I <= 2 && C <= 2) ? 2 : ((I <= 4 && C <= 4) ? 4 : 8);
    const size_t lds = m->L == 2 ? pyz_hmc_fused_lds_bytes(n_rows, MIC, MIC, C, (int)m->D, m->loss) : 0;
    // (the one-workgroup form needs the whole data set in LDS; the sliced form only its slice)
    const int allow_multi = pyz_env_int("PYZ_HMC_MULTI", 1);  // read per call: tests flip it
    const int rows_per_wg = std::max(16, pyz_env_int("PYZ_HMC_ROWS_PER_WG", 96));
    const int NW = std::min(PYZ_HM_MAXW, n_rows / rows_per_wg);
    const size_t mlds = m->L == 2 ? pyz_hmc_multi_lds_bytes(cdiv(n_rows, std::max(NW, 1)) + 1, MIC, MIC, C, (int)m->D, m->loss) : 0;
    const bool multi_ok = allow_multi && NW >= 2 && P <= pyz_env_int("PYZ_HMC_MULTI_MAX_CHAINS", 16) && mlds <= 150 * 1024;
    if (allow_fused && !d_prior_mean_vec && !d_prior_sigma_vec && m->L == 2 && I <= PYZ_HF_MAXI && C <= PYZ_HF_MAXC && H + C <= 64 &&
        (lds <= 150 * 1024 || multi_ok) && m->acts[0] != PYZ_ACT_SOFTMAX) {
      HmcFusedArgs f{};
      f.q = d_q;
      f.x = d_x;
      f.y = d_y;
      f.N = n_rows;
      f.I = I;
      f.H = H;
      f.C = C;
      f.D = (int)m->D;
      f.act_hidden = m->acts[0];
      f.act_last = m->acts[1];
      f.loss = m->loss;
      f.L = L;
      f.epsilon = epsilon;
      f.m = mass;
      f.prior_mean = prior_mean;
      f.prior_sigma = prior_sigma;
      f.burning = burning;
      f.uniform = unif;
      f.seed = seed;
      f.step = (uint32_t)step;
      f.unit_p = d_unit_p;
      f.stats = d_stats;
      void (*kern)(HmcFusedArgs) = nullptr;
      const int bucket = (I <= 2 && C <= 2) ? 0 : ((I <= 4 && C <= 4) ? 1 : 2);
#define PYZ_HF_PICK(ACT)                                                                         \
  kern = bucket == 0 ? k_hmc_fused<2, 2, ACT> : (bucket == 1 ? k_hmc_fused<4, 4, ACT> : k_hmc_fused<8, 8, ACT>)
      switch (m->acts[0]) {
        case PYZ_ACT_RELU: PYZ_HF_PICK(PYZ_ACT_RELU); break;
        case PYZ_ACT_TANH: PYZ_HF_PICK(PYZ_ACT_TANH); break;
        case PYZ_ACT_SIGMOID: PYZ_HF_PICK(PYZ_ACT_SIGMOID); break;
        default: PYZ_HF_PICK(PYZ_ACT_LINEAR); break;
      }
#undef PYZ_HF_PICK
      // few chains: spread each one over NW workgroups, one launch per gradient evaluation (pyz_hmc_multi.h);
      // many chains: one workgroup per chain, the whole proposal in one launch
      if (multi_ok) {
        HmcMultiArgs mm{};
        mm.f = f;
        mm.NW = NW;
        mm.max_rows = cdiv(n_rows, NW) + 1;
        const size_t D = (size_t)m->D;
        const size_t n_state = 2 * (size_t)P * D, n_slab = 2 * (size_t)P * NW * D;
        const size_t bytes = sizeof(float) * (2 * n_state + n_slab + 4 * (size_t)P) + sizeof(double) * 2 * (size_t)P * NW + 64;
        pyz_mlp_full *fm = full(m);
        if ((rc = ensure_bytes(&fm->x.hm_buf, &fm->x.hm_cap, bytes, m))) return rc;
        mm.lpart = reinterpret_cast<double *>(fm->x.hm_buf);  // doubles first: 8-byte aligned
        mm.qw = reinterpret_cast<float *>(mm.lpart + 2 * (size_t)P * NW);
        mm.pw = mm.qw + n_state;
        mm.slab = mm.pw + n_state;
        mm.scal = mm.slab + n_slab;
        void (*kmulti)(HmcMultiArgs) = nullptr;
#define PYZ_HM_PICK(ACT)                                                                         \
  kmulti = bucket == 0 ? k_hmc_multi<2, 2, ACT> : (bucket == 1 ? k_hmc_multi<4, 4, ACT> : k_hmc_multi<8, 8, ACT>)
        switch (m->acts[0]) {
          case PYZ_ACT_RELU: PYZ_HM_PICK(PYZ_ACT_RELU); break;
          case PYZ_ACT_TANH: PYZ_HM_PICK(PYZ_ACT_TANH); break;
          case PYZ_ACT_SIGMOID: PYZ_HM_PICK(PYZ_ACT_SIGMOID); break;
          default: PYZ_HM_PICK(PYZ_ACT_LINEAR); break;
        }
#undef PYZ_HM_PICK
        if (mlds > 64 * 1024)
          PYZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kmulti), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
        mm.call = call_dev;
        // the whole proposal as one resident launch (k_hmc_resident) when its NW x chains workgroups fit the chip at once
        void (*kres)(HmcMultiArgs) = nullptr;
#define PYZ_HR_PICK(ACT)                                                                         \
  kres = bucket == 0 ? k_hmc_resident<2, 2, ACT> : (bucket == 1 ? k_hmc_resident<4, 4, ACT> : k_hmc_resident<8, 8, ACT>)
        switch (m->acts[0]) {
          case PYZ_ACT_RELU: PYZ_HR_PICK(PYZ_ACT_RELU); break;
          case PYZ_ACT_TANH: PYZ_HR_PICK(PYZ_ACT_TANH); break;
          case PYZ_ACT_SIGMOID: PYZ_HR_PICK(PYZ_ACT_SIGMOID); break;
          default: PYZ_HR_PICK(PYZ_ACT_LINEAR); break;
        }
#undef PYZ_HR_PICK
        bool resident = pyz_env_int("PYZ_HMC_RESIDENT", 1) != 0 && NW * P <= pyz_cu_count();   // (read per call: tests flip it)
        if (resident) {
          if (mlds > 64 * 1024)
            PYZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kres), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
          int per_cu = 0;
          if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kres), PYZ_HM_THREADS, mlds) != hipSuccess || per_cu < 1)
            resident = false;
        }
        if (resident) {
          const int Dp = (int)((D + 2 + 31) / 32 * 32);
          const size_t rbytes = 256 + sizeof(unsigned long long) * 2 * (size_t)P * NW * Dp;
          const size_t had = fm->x.hm_res_cap;
          if ((rc = ensure_bytes(&fm->x.hm_res_buf, &fm->x.hm_res_cap, rbytes, m))) return rc;
          if (fm->x.hm_res_cap != had)   // a fresh buffer: epochs and tags start at zero (the kernel's tags are >= 1 and only grow)
            PYZ_HIP(hipMemsetAsync(fm->x.hm_res_buf, 0, fm->x.hm_res_cap, st));
          mm.epoch = reinterpret_cast<unsigned *>(fm->x.hm_res_buf);
          mm.gran = reinterpret_cast<unsigned long long *>(static_cast<unsigned char *>(fm->x.hm_res_buf) + 256);
          mm.Dp = Dp;
          mm.spin_limit = pyz_env_int("PYZ_HMC_SPIN_LIMIT", 1 << 20);
#ifdef PYZ_HMC_DIAG
          mm.diag = pyz_env_int("PYZ_HMC_DIAG", 0);
#endif
        }
        // one graph per (kernel, shapes, pointers, scalars): a proposal is L + 2 dependent launches (resident: a memset and one launch)
        unsigned long long key = 1469598103934665603ull;
        auto mix = [&](unsigned long long v) { key = (key ^ v) * 1099511628211ull; };
        mix((unsigned long long)(uintptr_t)(resident ? kres : kmulti)); mix((unsigned long long)(uintptr_t)mm.gran); mix((unsigned long long)mm.spin_limit + 7 * (unsigned long long)mm.diag); mix((unsigned long long)L); mix((unsigned long long)P); mix((unsigned long long)NW);
        mix((unsigned long long)n_rows); mix((unsigned long long)(uintptr_t)d_q); mix((unsigned long long)(uintptr_t)d_x);
        mix((unsigned long long)(uintptr_t)d_y); mix((unsigned long long)(uintptr_t)d_unit_p); mix((unsigned long long)(uintptr_t)d_stats);
        mix((unsigned long long)(uintptr_t)fm->x.hm_buf); mix((unsigned long long)(uintptr_t)st);
        unsigned fbits[4];
        memcpy(&fbits[0], &epsilon, 4); memcpy(&fbits[1], &mass, 4); memcpy(&fbits[2], &prior_mean, 4); memcpy(&fbits[3], &prior_sigma, 4);
        for (unsigned b : fbits) mix(b);
        const int use_graph = pyz_env_int("PYZ_HMC_GRAPH", 1);   // (read per call: tests flip it)
        auto launch_all = [&]() {
          if (resident) {
            PYZ_LAUNCH(kres, dim3(NW, P), dim3(PYZ_HM_THREADS), mlds, st, mm);
            return;
          }
          for (int t = 0; t <= L; ++t) {
            mm.t = t;
            PYZ_LAUNCH(kmulti, dim3(NW, P), dim3(PYZ_HM_THREADS), mlds, st, mm);
          }
          PYZ_LAUNCH(k_hmc_multi_final, dim3(P), dim3(PYZ_HM_THREADS), 0, st, mm);
        };
        if (!use_graph || st == nullptr) {  // the legacy default stream cannot be captured
          launch_all();
          PYZ_LAUNCH_CHECK();
          return PYZ_OK;
        }
        if (!fm->x.hm_exec || fm->x.hm_key != key) {
          if (fm->x.hm_exec) { (void)hipGraphExecDestroy(fm->x.hm_exec); fm->x.hm_exec = nullptr; }
          if (fm->x.hm_graph) { (void)hipGraphDestroy(fm->x.hm_graph); fm->x.hm_graph = nullptr; }
          PYZ_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
          launch_all();
          hipGraph_t gr = nullptr;
          PYZ_HIP(hipStreamEndCapture(st, &gr));
          fm->x.hm_graph = gr;
          PYZ_HIP(hipGraphInstantiate(&fm->x.hm_exec, fm->x.hm_graph, nullptr, nullptr, 0));
          fm->x.hm_key = key;
        }
        PYZ_HIP(hipGraphLaunch(fm->x.hm_exec, st));
        return PYZ_OK;
      }
      if (lds > 64 * 1024)
        PYZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      PYZ_LAUNCH(kern, dim3(P), dim3(PYZ_HF_THREADS), lds, st, f);
      PYZ_LAUNCH_CHECK();
      return PYZ_OK;
    }
  }
  if ((rc = set_ctl(m, 0, n_rows, 0.0f, step, 0, 0, st))) return rc;
  HmcArgs a{};
  a.q = d_q;
  a.p = m->grad2;
  a.qsave = m->qsave;
  a.grad = m->grad;
  a.D = m->D;
  a.m = mass;
  a.prior_mean = prior_mean;
  a.prior_sigma = prior_sigma;
  a.pm_vec = d_prior_mean_vec;
  a.ps_vec = d_prior_sigma_vec;
  a.n_train = (float)n_rows;
  a.seed = seed;
  a.step = (uint32_t)step;
  a.unit_p = d_unit_p;
  a.part = f->x.part2;
  auto grad_eval = [&]() {
    WgradArgs u{};
    u.mode = PYZ_UPD_NONE;
    u.grad = m->grad;
    u.grad_pstride = m->D;
    launch_loss_backward(m, d_q, m->D, P, d_x, d_y, nullptr, n_rows, m->ctl, true, u, st);
    PYZ_LAUNCH(k_loss_finalize, dim3(P), dim3(64), 0, st, m->part, m->cur_nblk, m->ctl, loss, m->nonfinite);
  };
  // momentum, snapshot, K0 and the prior part of U0
  a.nblk = nblk4;
  PYZ_LAUNCH(k_hmc_begin, dim3(nblk4, P), dim3(256), 0, st, a);
  grad_eval();
  PYZ_LAUNCH(k_hmc_energy_finalize, dim3(P), dim3(64), 0, st, f->x.part2, nblk4, loss, a.n_train, mass, energies, 0);
  // half kick + first drift (HMC.py:82-84); with L == 0 the two half kicks share the gradient
  a.nblk = nblk1;
  if (L == 0) {
    a.kick1 = epsilon / 2;
    a.kick2 = epsilon / 2;
    a.drift = 0.0f;
    PYZ_LAUNCH(k_hmc_kick_drift, dim3(nblk1, P), dim3(256), 0, st, a);
  } else {
    a.kick1 = epsilon / 2;
    a.kick2 = 0.0f;
    a.drift = epsilon / mass;
    PYZ_LAUNCH(k_hmc_kick_drift, dim3(nblk1, P), dim3(256), 0, st, a);
    for (int i = 1; i <= L; ++i) {
      grad_eval();
      a.kick1 = epsilon;
      if (i < L) {
        a.kick2 = 0.0f;
        a.drift = epsilon / mass;
      } else {  // last full kick and the closing half kick use the same gradient (HMC.py:86-87)
        a.kick2 = epsilon / 2;
        a.drift = 0.0f;
      }
      PYZ_LAUNCH(k_hmc_kick_drift, dim3(nblk1, P), dim3(256), 0, st, a);
    }
  }
  // K1, U1 (the loss of the last gradient evaluation is the loss at the proposal)
  PYZ_LAUNCH(k_hmc_end_energy, dim3(nblk1, P), dim3(256), 0, st, a);
  PYZ_LAUNCH(k_hmc_energy_finalize, dim3(P), dim3(64), 0, st, f->x.part2, nblk1, loss, a.n_train, mass, energies, 1);
  PYZ_LAUNCH(k_hmc_accept, dim3(cdiv(P, 64)), dim3(64), 0, st, energies, unif, burning, d_stats, P);
  PYZ_LAUNCH(k_hmc_restore, dim3(nblk1, P), dim3(256), 0, st, d_q, m->qsave, d_stats, m->D);
  PYZ_LAUNCH_CHECK();
  return PYZ_OK;
}

// one-time check of the lane layout k_svgd_gram_tile assumes for v_mfma_f64_16x16x4_f64
static bool mfma_f64_layout_ok(hipStream_t st) {
  static int state = -1;  // -1 unknown, 0 no, 1 yes
  if (state >= 0) return state == 1;
  int *d = nullptr;
  int h[512];
  state = 0;
  if (hipMalloc((void **)&d, sizeof h) != hipSuccess) return false;
  PYZ_LAUNCH(k_probe_mfma_f64, dim3(1), dim3(64), 0, st, d);
  if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
    bool ok = true;
    for (int l = 0; l < 64 && ok; ++l)
      for (int r = 0; r < 4; ++r)
        if (h[4 * l + r] != (l >> 4) + 4 * r || h[256 + 4 * l + r] != (l & 15)) ok = false;
    state = ok ? 1 : 0;
  }
  (void)hipFree(d);
  return state == 1;
}

// ---------------------------------------------------------------- V2-V4
// phase 1 of SVGD.step: all loss gradients of the local particles in one particle-batched pass -- g_i depends on
// particle i only (SVGD.py:104-111); they stay in the plan's gradient buffer, the losses in its scalars
static int svgd_gradients_impl(pyz_mlp *m, const float *d_particles, int n_local, const float *d_x, const void *d_y,
                               const int32_t *d_row_idx, int batch, hipStream_t st) {
  int rc = check_call(m, n_local, batch);
  if (rc) return rc;
  if ((rc = check_loss_combo(m))) return rc;
  if (!d_particles || !d_x || !d_y) return pyz_fail(PYZ_E_INVALID, "null device pointer");
  if ((rc = need_grad(m, n_local))) return rc;
  m->grad_owner = 1;
  m->grad_rows = n_local;
  set_ctl_lazy(m, batch, 0.0f, 0);   // the first kernel of the pass carries the step scalars
  WgradArgs u{};
  u.mode = PYZ_UPD_NONE;
  u.grad = m->grad;
  u.grad_pstride = m->D;
  const bool fused = can_fuse(m);
  if (fused) u.ploss = m->scal;      // per-particle losses in the weight-gradient launch (every particle's spare workgroup)
  launch_loss_backward(m, d_particles, m->D, n_local, d_x, d_y, d_row_idx, batch, m->ctl, true, u, st);
  if (!fused) PYZ_LAUNCH(k_loss_finalize, dim3(n_local), dim3(64), 0, st, m->part, m->cur_nblk, m->ctl, m->scal, m->nonfinite);
  PYZ_LAUNCH_CHECK();
  return PYZ_OK;
}

// ---- phase 2 on a snapshot, all rows at once (Jacobi; M <= 64, local rows in multiples of four), in two halves:
//   kernel matrix   the squared distances of the local rows against the snapshot (Gram matrix on the float64 matrix
//                   cores, or pairwise), [the median-heuristic bandwidth,] K rows and their sums -- a function of the
//                   snapshot ALONE: it may run on a second stream while phase 1 computes the gradients;
//   combine         phi, the legacy Adam step and the step's loss for every local row -- needs both.
// Both carve the same layout out of the plan's float64 scratch.
struct TileLayout {
  SvgdTileArgs ta;   // the local rows
  SvgdTileArgs td;   // the distance pass (all rows under the median heuristic)
};

static bool svgd_tile_shape(int n_local, int n_total, int row0, const float *d_all, const float *d_particles) {
  return n_total <= 64 && row0 % 4 == 0 && n_local % 4 == 0 && d_all != d_particles;
}

static int svgd_tile_layout(pyz_mlp *m, const float *d_all, int n_total, int row0, int n_local, float gamma, TileLayout &L) {
  const bool median = gamma == PYZ_SVGD_GAMMA_MEDIAN;
  SvgdTileArgs ta{};
  ta.all = d_all;
  ta.D = m->D;
  ta.M = n_total;
  ta.n_local = n_local;
  ta.row0 = row0;
  ta.gamma = gamma;
  ta.range = PYZ_SV_E * cdiv(m->D, 256LL * PYZ_SV_E);  // one round of workgroups on the 256 CUs
  ta.nblk = cdiv(m->D, ta.range);
  // the median heuristic (SVGD.py:165-181) needs the squared distances of ALL pairs: the distance pass then covers
  // every row of the gathered matrix on every rank (the matrix is read once either way)
  const int dist_rows = median ? n_total : n_local, dist_row0 = median ? 0 : row0;
  const size_t n_part = (size_t)dist_rows * ta.nblk * 64, n_k = (size_t)n_local * 64, n_diag = (size_t)ta.nblk * 64;
  const size_t n_dmat = median ? (size_t)n_total * 64 + 8 : 0;
  const int rc = need_part2(m, n_part + n_k + 2 * (size_t)n_local + n_diag + n_dmat + 16, /*keep_kernel_matrix=*/true);
  if (rc) return rc;
  ta.part = full(m)->x.part2;
  ta.kmat = ta.part + n_part;
  ta.ksumd = ta.kmat + n_k;
  ta.ksum = reinterpret_cast<float *>(ta.ksumd + n_local);
  double *after_ksum = ta.ksumd + n_local + (n_local + 1) / 2 + 1;
  SvgdTileArgs td = ta;
  td.n_local = dist_rows;
  td.row0 = dist_row0;
  td.diag = after_ksum;                        // (only read when the Gram form ran: see the kernel-matrix half)
  if (median) {
    td.dmat = after_ksum + n_diag;             // (M, 64) squared distances, then the bandwidth
    td.gamma_dev = td.dmat + (size_t)n_total * 64;
    ta.dmat = td.dmat;
    ta.gamma_dev = td.gamma_dev;
  }
  L.ta = ta;
  L.td = td;
  return PYZ_OK;
}

static int svgd_check_rows(const pyz_mlp *m, int n_local, int n_total, int row0, float gamma) {
  if (!m) return pyz_fail(PYZ_E_INVALID, "null plan");
  if (n_local <= 0 || n_local > m->max_p) return pyz_fail(PYZ_E_SHAPE, "particle count %d outside [1, %d] of the plan", n_local, m->max_p);
  if (n_total < n_local || row0 < 0 || row0 + n_local > n_total) return pyz_fail(PYZ_E_INVALID, "rows [%d, %d) outside the %d particles", row0, row0 + n_local, n_total);
  if (n_total > 1024) return pyz_fail(PYZ_E_INVALID, "more than 1024 particles");
  if (gamma != PYZ_SVGD_GAMMA_MEDIAN && !(gamma > 0.0f)) return pyz_fail(PYZ_E_INVALID, "gamma must be positive (or PYZ_SVGD_GAMMA_MEDIAN)");
  return PYZ_OK;
}

// the distance pass over blocks [blk0, blk0 + n_blocks) of the layout (partials of td's rows into the plan's scratch)
static void svgd_launch_distance_pass(SvgdTileArgs td, int blk0, int n_blocks, bool gram, hipStream_t st) {
  td.blk0 = blk0;
  if (gram) {
    // a shard whose rows lie in one or two row blocks of 16 computes those blocks only; every other row range takes the
    // 10 upper blocks of the whole matrix (identical bits per entry, pyz_kernels.h)
    const int rb_lo = td.row0 >> 4, rb_hi = (td.row0 + td.n_local - 1) >> 4;
    const size_t gr_lds = pyz_svgd_gram_lds_bytes();
    if (rb_hi == rb_lo) PYZ_LAUNCH((k_svgd_gram_tile<1, false>), dim3(n_blocks), dim3(512), gr_lds, st, td);
    else if (rb_hi == rb_lo + 1) PYZ_LAUNCH((k_svgd_gram_tile<2, false>), dim3(n_blocks), dim3(512), gr_lds, st, td);
    else PYZ_LAUNCH((k_svgd_gram_tile<4, true>), dim3(n_blocks), dim3(512), gr_lds, st, td);
  } else {
    PYZ_LAUNCH(k_svgd_dist_tile, dim3(n_blocks), dim3(256), 0, st, td);
  }
}

// d_groups != nullptr: the group sums of the partial squared distances of ALL rows, gathered from the ranks that each
// summed some groups (pyz_svgd_gram_groups); nullptr: this device runs the whole distance pass itself
static int svgd_kernel_matrix_impl(pyz_mlp *m, const float *d_all, int n_total, int row0, int n_local, float gamma,
                                   const double *d_groups, hipStream_t st) {
  int rc = svgd_check_rows(m, n_local, n_total, row0, gamma);
  if (rc) return rc;
  if (!d_all) return pyz_fail(PYZ_E_INVALID, "null device pointer");
  const bool median = gamma == PYZ_SVGD_GAMMA_MEDIAN;
  if (!svgd_tile_shape(n_local, n_total, row0, d_all, nullptr) || (median && n_total % 4 != 0))
    return pyz_fail(PYZ_E_INVALID, "the kernel matrix of a snapshot needs at most 64 particles and local rows [row0, row0 + n) in "
                                   "multiples of four (the median heuristic: the particle count too)");
  m->km_valid = false;
  TileLayout L;
  if ((rc = svgd_tile_layout(m, d_all, n_total, row0, n_local, gamma, L))) return rc;
  L.ta.groups = L.td.groups = d_groups;
  if (!d_groups) {
    // distances through the Gram matrix on the float64 matrix cores when the instruction's lane layout is the
    // one the kernel assumes (probed once); else the pairwise float64 VALU kernel
    const int gram_on = pyz_env_int("PYZ_SVGD_GRAM", 1);  // read per call: tests flip it
    svgd_launch_distance_pass(L.td, 0, L.td.nblk, gram_on && mfma_f64_layout_ok(st), st);
  }
  if (median) {
    PYZ_LAUNCH(k_svgd_kmat, dim3(n_total), dim3(1024), 0, st, L.td, 1);   // distances only
    PYZ_LAUNCH(k_svgd_median, dim3(1), dim3(1024), 0, st, L.td);
  }
  PYZ_LAUNCH(k_svgd_kmat, dim3(n_local), dim3(1024), 0, st, L.ta, 0);
  PYZ_LAUNCH_CHECK();
  m->km_valid = true;
  m->km_all = d_all;
  m->km_total = n_total;
  m->km_row0 = row0;
  m->km_local = n_local;
  m->km_gamma = gamma;
  return PYZ_OK;
}

// the distance pass of groups [g_lo, g_hi) of the PYZ_SVGD_GROUPS groups of blocks (this rank's share of the ELEMENTS),
// all rows; the group sums go to the caller's (PYZ_SVGD_GROUPS, 64, 64) buffer at the groups' places
static int svgd_gram_groups_impl(pyz_mlp *m, const float *d_all, int n_total, int g_lo, int g_hi, double *d_groups, hipStream_t st) {
  if (!m) return pyz_fail(PYZ_E_INVALID, "null plan");
  if (!d_all || !d_groups) return pyz_fail(PYZ_E_INVALID, "null device pointer");
  if (n_total < 4 || n_total > 64 || n_total % 4 != 0) return pyz_fail(PYZ_E_INVALID, "the grouped distance pass takes 4 .. 64 particles in multiples of four");
  if (g_lo < 0 || g_hi > PYZ_SVGD_GROUPS || g_lo >= g_hi) return pyz_fail(PYZ_E_INVALID, "groups [%d, %d) outside [0, %d)", g_lo, g_hi, PYZ_SVGD_GROUPS);
  m->km_valid = false;
  TileLayout L;
  // (the layout of the WHOLE matrix: partials of all rows; gamma does not enter the distance pass)
  int rc = svgd_tile_layout(m, d_all, n_total, 0, n_total, 1.0f, L);
  if (rc) return rc;
  const int nblk = L.td.nblk, nb8 = cdiv(nblk, PYZ_SVGD_GROUPS);
  const int blk0 = std::min(g_lo * nb8, nblk), blk1 = std::min(g_hi * nb8, nblk);
  const int gram_on = pyz_env_int("PYZ_SVGD_GRAM", 1);
  if (blk1 > blk0) svgd_launch_distance_pass(L.td, blk0, blk1 - blk0, gram_on && mfma_f64_layout_ok(st), st);
  PYZ_LAUNCH(k_svgd_group_reduce, dim3(n_total, g_hi - g_lo), dim3(128), 0, st, (const double *)L.td.part, nblk, g_lo, d_groups);
  PYZ_LAUNCH_CHECK();
  return PYZ_OK;
}

static int svgd_check_gradients(const pyz_mlp *m, int n_local) {
  if (!m->grad || m->grad_owner != 1 || m->grad_rows != n_local)
    return pyz_fail(PYZ_E_INVALID, "no gradients of these %d particles in the plan: pyz_svgd_gradients must be the last "
                                   "gradient-producing call on it", n_local);
  return PYZ_OK;
}

static int svgd_combine_impl(pyz_mlp *m, float *d_particles, int n_local, const float *d_all, int n_total, int row0,
                             float *d_adam_m, float *d_adam_v, float lr, float gamma, int64_t t, float *d_loss, hipStream_t st) {
  int rc = svgd_check_rows(m, n_local, n_total, row0, gamma);
  if (rc) return rc;
  if (!d_particles || !d_all || !d_adam_m || !d_adam_v || !d_loss) return pyz_fail(PYZ_E_INVALID, "null device pointer");
  if (t < 1) return pyz_fail(PYZ_E_INVALID, "Adam step t must be >= 1");
  if ((rc = svgd_check_gradients(m, n_local))) return rc;
  if (!m->km_valid || m->km_all != d_all || m->km_total != n_total || m->km_row0 != row0 || m->km_local != n_local || m->km_gamma != gamma)
    return pyz_fail(PYZ_E_INVALID, "pyz_svgd_combine without the kernel matrix of this snapshot (pyz_svgd_kernel_matrix with the "
                                   "same d_all, rows and gamma must precede it, with no other call on the plan's scratch in between)");
  TileLayout L;
  if ((rc = svgd_tile_layout(m, d_all, n_total, row0, n_local, gamma, L))) return rc;
  SvgdTileArgs &ta = L.ta;
  ta.particles = d_particles;
  ta.adam_m = d_adam_m;
  ta.adam_v = d_adam_v;
  ta.grad = m->grad;
  const double b1t = std::pow(0.9, (double)t), b2t = std::pow(0.999, (double)t);
  ta.lr_t = (float)((double)lr * std::sqrt(1.0 - b2t) / (1.0 - b1t));
  ta.loss_in = m->scal;   // [n_local], written by phase 1
  ta.loss_out = d_loss;
  PYZ_LAUNCH(k_svgd_update_tile, dim3(cdiv(m->D, 256)), dim3(256), 0, st, ta, (const double *)ta.kmat, (const float *)ta.ksum,
             (const double *)ta.ksumd, (const double *)ta.gamma_dev);
  PYZ_LAUNCH_CHECK();
  m->km_valid = false;     // consumed (the next step has another snapshot)
  return PYZ_OK;
}

// phase 2: kernel row(s), repulsion, Adam (SVGD.py:54-68,112-123) on the gradients phase 1 left
static int svgd_sweep_impl(pyz_mlp *m, float *d_particles, int n_local, const float *d_all, int n_total, int row0,
                           float *d_adam_m, float *d_adam_v, float lr, float gamma, int64_t t, int sweep, float *d_loss,
                           hipStream_t st) {
  int rc = svgd_check_rows(m, n_local, n_total, row0, gamma);
  if (rc) return rc;
  if (!d_particles || !d_all || !d_adam_m || !d_adam_v || !d_loss) return pyz_fail(PYZ_E_INVALID, "null device pointer");
  if (t < 1) return pyz_fail(PYZ_E_INVALID, "Adam step t must be >= 1");
  const bool median = gamma == PYZ_SVGD_GAMMA_MEDIAN;
  if (sweep != PYZ_SWEEP_GAUSS_SEIDEL && sweep != PYZ_SWEEP_JACOBI) return pyz_fail(PYZ_E_INVALID, "unknown sweep %d", sweep);
  if (sweep == PYZ_SWEEP_GAUSS_SEIDEL && (d_all != d_particles || n_local != n_total))
    return pyz_fail(PYZ_E_INVALID, "the Gauss-Seidel sweep needs the whole particle matrix on this device");
  if ((rc = svgd_check_gradients(m, n_local))) return rc;
  const int tiles_on = pyz_env_int("PYZ_SVGD_TILES", 1);  // read per call: tests flip it
  const bool tile_ok = sweep == PYZ_SWEEP_JACOBI && svgd_tile_shape(n_local, n_total, row0, d_all, d_particles);
  if (median && !(tile_ok && n_total % 4 == 0))
    return pyz_fail(PYZ_E_INVALID, "the median-heuristic bandwidth needs the Jacobi sweep on a snapshot, at most 64 particles, "
                                   "and particle / local-row counts in multiples of four");
  if (tile_ok && (tiles_on || median)) {
    // every row from the same snapshot: the particle matrix is read once per pass (k_svgd_*_tile)
    if ((rc = svgd_kernel_matrix_impl(m, d_all, n_total, row0, n_local, gamma, nullptr, st))) return rc;
    return svgd_combine_impl(m, d_particles, n_local, d_all, n_total, row0, d_adam_m, d_adam_v, lr, gamma, t, d_loss, st);
  }
  const int nblk = cdiv(m->D, PYZ_SVGD_BLOCK_ELEMS);  // one float64 partial per workgroup of k_svgd_dist
  const int rows_at_once = sweep == PYZ_SWEEP_JACOBI ? n_local : 1;
  if ((rc = need_part2(m, (size_t)rows_at_once * nblk * n_total + 8))) return rc;
  float *loss = m->scal;  // [n_local], written by phase 1
  SvgdArgs a{};
  a.particles = d_particles;
  a.all = d_all;
  a.all_rw = nullptr;
  a.adam_m = d_adam_m;
  a.adam_v = d_adam_v;
  a.grad = m->grad;
  a.D = m->D;
  a.M = n_total;
  a.n_local = n_local;
  a.row0 = row0;
  const double b1t = std::pow(0.9, (double)t), b2t = std::pow(0.999, (double)t);
  a.lr_t = (float)((double)lr * std::sqrt(1.0 - b2t) / (1.0 - b1t));
  a.gamma = gamma;
  a.part = full(m)->x.part2;
  a.nblk = nblk;
  const size_t lds = sizeof(double) * (size_t)(5 * n_total);
  const int jgroups = cdiv(n_total, 8);
  if (sweep == PYZ_SWEEP_JACOBI) {
    a.i_local = -1;
    PYZ_LAUNCH(k_svgd_dist, dim3(nblk, jgroups, n_local), dim3(256), 0, st, a);
    PYZ_LAUNCH(k_svgd_update, dim3(cdiv(m->D, 256), n_local), dim3(256), lds, st, a);
  } else {
    const int gs_fused = pyz_env_int("PYZ_SVGD_GS_FUSED", 1);  // read per call: tests flip it
    const long long gs_range = 256LL * PYZ_GS_E;
    if (gs_fused && n_total <= 64 && m->D <= 256 * gs_range) {
      // one launch per particle, the matrix read once per launch (k_svgd_gs)
      SvgdGsArgs ga{};
      ga.all = d_particles;
      ga.adam_m = d_adam_m;
      ga.adam_v = d_adam_v;
      ga.grad = m->grad;
      ga.D = m->D;
      ga.M = n_total;
      ga.lr_t = a.lr_t;
      ga.gamma = gamma;
      ga.nblk = (int)cdiv(m->D, gs_range);
      const size_t n_part = (size_t)ga.nblk * 64;
      if ((rc = need_part2(m, 2 * n_part + 8))) return rc;
      double *pp[2] = {full(m)->x.part2, full(m)->x.part2 + n_part};
      const size_t gs_lds = pyz_svgd_gs_lds_bytes();
      // (k_svgd_gs's 130 KB of dynamic LDS: allowed when the plan was created, pyz_mlp_create)
      // the whole sweep as ONE resident launch when its workgroups fit the chip at once (k_svgd_gs_resident); read per
      // call: tests flip it
      // workgroups that only sum columns of partials: four columns each in k_svgd_gs_resident, eight in k_svgd_gs_resident2
      const int gs_reducers = pyz_env_int("PYZ_SVGD_GS_RESIDENT", 1) == 2 ? cdiv(n_total, 8) : cdiv(n_total, 4);
      // (1: the partials of row i + 1 through the reducers behind the update of row i; 2: the distances one step early and the
      //  one critical distance in a single hop, k_svgd_gs_resident2)
      const int gs_res_mode = pyz_env_int("PYZ_SVGD_GS_RESIDENT", 1);
      bool resident = gs_res_mode != 0 && ga.nblk + gs_reducers <= pyz_cu_count();
      const void *gs_res_fn = gs_res_mode == 2 ? reinterpret_cast<const void *>(k_svgd_gs_resident2) : reinterpret_cast<const void *>(k_svgd_gs_resident);
      if (resident) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gs_res_fn, 256,
                                                         pyz_svgd_gs_res_lds_bytes()) != hipSuccess || per_cu < 1)
          resident = false;
      }
      if (resident) {
        pyz_mlp_full *fm = full(m);
        const size_t gbytes = 256 + sizeof(unsigned long long) * 2 * (2 * (size_t)256 * 64 + 2 * 64 + 2 * 256);
        const size_t had = fm->x.gs_res_cap;
        if ((rc = ensure_bytes(&fm->x.gs_res_buf, &fm->x.gs_res_cap, gbytes, m))) return rc;
        if (fm->x.gs_res_cap != had)   // a fresh buffer: epoch and tags start at zero (the kernel's tags are >= 1 and only grow)
          PYZ_HIP(hipMemsetAsync(fm->x.gs_res_buf, 0, fm->x.gs_res_cap, st));
        unsigned char *gb = static_cast<unsigned char *>(fm->x.gs_res_buf);
        SvgdGsResArgs ra{};
        ra.all = d_particles;
        ra.adam_m = d_adam_m;
        ra.adam_v = d_adam_v;
        ra.grad = m->grad;
        ra.D = m->D;
        ra.M = n_total;
        ra.lr_t = a.lr_t;
        ra.gamma = gamma;
        ra.nblk = ga.nblk;
        ra.epoch = reinterpret_cast<unsigned *>(gb);
        ra.fail = reinterpret_cast<int *>(gb + 64);
        ra.kgran = reinterpret_cast<unsigned long long *>(gb + 256);
        ra.pgran = ra.kgran + 2 * 64 * 2;
        ra.cgran = ra.pgran + 2 * (size_t)256 * 64 * 2;
        ra.spin_limit = pyz_env_int("PYZ_SVGD_GS_SPIN_LIMIT", 1 << 20);
        if (gs_res_mode == 2) PYZ_LAUNCH(k_svgd_gs_resident2, dim3(ga.nblk + gs_reducers), dim3(256), pyz_svgd_gs_res_lds_bytes(), st, ra);
        else PYZ_LAUNCH(k_svgd_gs_resident, dim3(ga.nblk + gs_reducers), dim3(256), pyz_svgd_gs_res_lds_bytes(), st, ra);
        PYZ_LAUNCH(k_svgd_loss, dim3(1), dim3(64), 0, st, loss, n_local, n_total, d_loss, ra.fail, m->nonfinite);
        PYZ_LAUNCH_CHECK();
        return PYZ_OK;
      }
      const int zigzag = pyz_env_int("PYZ_SVGD_GS_ZIGZAG", 1);   // alternate the row direction from launch to launch (read per call)
      for (int i = -1; i < n_total; ++i) {
        ga.i = i;
        ga.part_in = pp[(i + 2) & 1];   // what launch i - 1 wrote
        ga.part_out = pp[(i + 1) & 1];
        if (zigzag && (i & 1) == 0) PYZ_LAUNCH(k_svgd_gs<true>, dim3(ga.nblk), dim3(256), gs_lds, st, ga);
        else PYZ_LAUNCH(k_svgd_gs<false>, dim3(ga.nblk), dim3(256), gs_lds, st, ga);
      }
    } else {
      for (int i = 0; i < n_local; ++i) {
        a.i_local = i;
        PYZ_LAUNCH(k_svgd_dist, dim3(nblk, jgroups, 1), dim3(256), 0, st, a);
        PYZ_LAUNCH(k_svgd_update, dim3(cdiv(m->D, 256), 1), dim3(256), lds, st, a);
      }
    }
  }
  PYZ_LAUNCH(k_svgd_loss, dim3(1), dim3(64), 0, st, loss, n_local, n_total, d_loss, (int *)nullptr, m->nonfinite);
  PYZ_LAUNCH_CHECK();
  return PYZ_OK;
}

int pyz_svgd_gradients(pyz_mlp *m, const float *d_particles, int n_local, const float *d_x, const void *d_y,
                       const int32_t *d_row_idx, int batch, void *stream) {
  return svgd_gradients_impl(m, d_particles, n_local, d_x, d_y, d_row_idx, batch, as_stream(stream));
}

int pyz_svgd_sweep(pyz_mlp *m, float *d_particles, int n_local, const float *d_all, int n_total, int row0,
                   float *d_adam_m, float *d_adam_v, float lr, float gamma, int64_t t, int sweep, float *d_loss,
                   void *stream) {
  return svgd_sweep_impl(m, d_particles, n_local, d_all, n_total, row0, d_adam_m, d_adam_v, lr, gamma, t, sweep, d_loss,
                         as_stream(stream));
}

int pyz_svgd_kernel_matrix(pyz_mlp *m, const float *d_all, int n_total, int row0, int n_local, float gamma, void *stream) {
  return svgd_kernel_matrix_impl(m, d_all, n_total, row0, n_local, gamma, nullptr, as_stream(stream));
}

int pyz_svgd_gram_groups(pyz_mlp *m, const float *d_all, int n_total, int g_lo, int g_hi, double *d_groups, void *stream) {
  return svgd_gram_groups_impl(m, d_all, n_total, g_lo, g_hi, d_groups, as_stream(stream));
}

int pyz_svgd_kernel_matrix_groups(pyz_mlp *m, const double *d_groups, const float *d_all, int n_total, int row0, int n_local,
                                  float gamma, void *stream) {
  if (!d_groups) return pyz_fail(PYZ_E_INVALID, "null device pointer");
  return svgd_kernel_matrix_impl(m, d_all, n_total, row0, n_local, gamma, d_groups, as_stream(stream));
}

int pyz_svgd_combine(pyz_mlp *m, float *d_particles, int n_local, const float *d_all, int n_total, int row0,
                     float *d_adam_m, float *d_adam_v, float lr, float gamma, int64_t t, float *d_loss, void *stream) {
  return svgd_combine_impl(m, d_particles, n_local, d_all, n_total, row0, d_adam_m, d_adam_v, lr, gamma, t, d_loss,
                           as_stream(stream));
}

int pyz_svgd_step(pyz_mlp *m, float *d_particles, int n_local, const float *d_all, int n_total, int row0,
                  float *d_adam_m, float *d_adam_v, const float *d_x, const void *d_y, const int32_t *d_row_idx,
                  int batch, float lr, float gamma, int64_t t, int sweep, float *d_loss, void *stream) {
  // Under the Jacobi sweep d_particles is only written (pyz.h): the rows' current values -- what the gradients are taken
  // at -- are rows [row0, row0 + n_local) of the snapshot
  if (!d_all || !d_particles) return pyz_fail(PYZ_E_INVALID, "null device pointer");
  if (n_total < n_local || row0 < 0 || row0 + n_local > n_total) return pyz_fail(PYZ_E_INVALID, "rows [%d, %d) outside the %d particles", row0, row0 + n_local, n_total);
  const float *cur = (sweep == PYZ_SWEEP_JACOBI && d_all != d_particles && m) ? d_all + (long long)row0 * m->D : d_particles;
  const int rc = svgd_gradients_impl(m, cur, n_local, d_x, d_y, d_row_idx, batch, as_stream(stream));
  if (rc) return rc;
  return svgd_sweep_impl(m, d_particles, n_local, d_all, n_total, row0, d_adam_m, d_adam_v, lr, gamma, t, sweep, d_loss,
                         as_stream(stream));
}

// ---------------------------------------------------------------- R1
int pyz_predict(pyz_mlp *m, const float *d_weights, int n_samples, const float *d_x, int n, float *d_samples,
                float *d_mean, void *stream) {
  if (!m) return pyz_fail(PYZ_E_INVALID, "null plan");
  if (n_samples <= 0 || n <= 0) return pyz_fail(PYZ_E_INVALID, "n_samples and n must be positive");
  if (n > m->max_batch) return pyz_fail(PYZ_E_SHAPE, "n %d exceeds the plan's max_batch %d", n, m->max_batch);
  if (!d_weights || !d_x || !d_mean) return pyz_fail(PYZ_E_INVALID, "null device pointer");
  hipStream_t st = as_stream(stream);
  int rc = set_ctl(m, 0, n, 0.0f, 0, 0, 0, st);
  if (rc) return rc;
  const int C = m->dims[m->L];
  const int softmax = m->acts[m->L - 1] == PYZ_ACT_SOFTMAX ? 1 : 0;
  for (int s0 = 0; s0 < n_samples; s0 += m->max_p) {
    const int S = std::min(m->max_p, n_samples - s0);
    launch_forward(m, d_weights + (long long)s0 * m->D, m->D, S, d_x, nullptr, n, m->ctl, st);
    PYZ_LAUNCH(k_predict_rows, dim3(cdiv(n, 256), S), dim3(256), 0, st, m->act[m->L - 1], (long long)m->max_batch * C, C,
               softmax, n, d_samples ? d_samples + (long long)s0 * n * C : nullptr);
    PYZ_LAUNCH(k_predict_mean, dim3((unsigned)cdiv((long long)n * C, 256)), dim3(256), 0, st, m->act[m->L - 1],
               (long long)m->max_batch * C, (long long)n * C, S, d_mean, s0 > 0 ? 1 : 0, 1.0f / (float)n_samples);
  }
  PYZ_LAUNCH_CHECK();
  return PYZ_OK;
}

// ---------------------------------------------------------------- noise
int pyz_sample_normal_rows(float *d_out, int64_t n_rows, int64_t row_stride, int64_t col0, int64_t len, const float *d_loc,
                           const float *d_scale, uint64_t seed, uint32_t stream_id, uint32_t first_draw, void *stream) {
  if (!d_out || !d_loc || !d_scale) return pyz_fail(PYZ_E_INVALID, "null device pointer");
  if (n_rows <= 0 || n_rows > 65535 || len <= 0 || col0 < 0 || col0 + len > row_stride)
    return pyz_fail(PYZ_E_INVALID, "bad draw count / column range");
  PYZ_LAUNCH(k_sample_normal_rows, dim3(cdiv(cdiv(len, 4), 256), (unsigned)n_rows), dim3(256), 0, as_stream(stream), d_out,
                     (long long)row_stride, (long long)col0, (long long)len, d_loc, d_scale, seed, stream_id, first_draw);
  PYZ_LAUNCH_CHECK();
  return PYZ_OK;
}

int pyz_fill_normal(float *d_out, int64_t n, uint64_t seed, uint32_t stream_id, uint32_t step, float mean, float std,
                    void *stream) {
  if (!d_out || n <= 0) return pyz_fail(PYZ_E_INVALID, "bad output buffer");
  PYZ_LAUNCH(k_fill_normal, dim3(cdiv(cdiv(n, 4), 256)), dim3(256), 0, as_stream(stream), d_out, (long long)n,
                     seed, stream_id, step, mean, std);
  PYZ_LAUNCH_CHECK();
  return PYZ_OK;
}

// ---------------------------------------------------------------- measurement hook
int pyz_bench_dense_kernel(pyz_mlp *m, int kind, int layer, const float *d_theta, int P, const float *d_x,
                           const int32_t *d_row_idx, int batch, float *d_grad, int iters, void *stream) {
  int rc = check_call(m, P, batch);
  if (rc) return rc;
  if (layer < 0 || layer >= m->L || kind < 0 || kind > 2 || iters < 1) return pyz_fail(PYZ_E_INVALID, "bad kind/layer/iters");
  if (kind == 1 && layer == 0) return pyz_fail(PYZ_E_INVALID, "layer 0 has no data gradient");
  if (!d_theta || !d_x || (kind == 2 && !d_grad)) return pyz_fail(PYZ_E_INVALID, "null device pointer");
  hipStream_t st = as_stream(stream);
  if ((rc = set_ctl(m, 0, batch, 0.0f, 0, 0, 0, st))) return rc;
  const int K = m->dims[layer], N = m->dims[layer + 1];
  DenseArgs g{};
  g.K = K;
  g.N = N;
  g.theta = d_theta;
  g.theta_pstride = m->D;
  g.w_off = m->w_off[layer];
  g.ctl = m->ctl;
  if (kind == 1) {
    g.in = m->delta[layer];
    g.in_pstride = (long long)m->max_batch * N;
    g.lda = N;
    g.out = m->delta[layer - 1];
    g.out_pstride = (long long)m->max_batch * K;
    g.aux = m->act[layer - 1];
    g.aux_pstride = (long long)m->max_batch * K;
    g.act = m->acts[layer - 1];
    g.vec = (N % 8 == 0) && (m->w_off[layer] % 4 == 0) && (P == 1 || m->D % 4 == 0) && aligned16(d_theta) ? 1 : 0;
  } else if (kind == 0) {
    if (layer == 0) {
      g.in = d_x;
      g.in_pstride = 0;
      g.row_idx = d_row_idx;
    } else {
      g.in = m->act[layer - 1];
      g.in_pstride = (long long)m->max_batch * K;
    }
    g.lda = K;
    g.out = m->act[layer];
    g.out_pstride = (long long)m->max_batch * N;
    g.act = m->acts[layer];
    g.vec = (K % 8 == 0) && aligned16(g.in) ? 1 : 0;
  }
  WgradArgs u{};
  u.mode = PYZ_UPD_NONE;
  u.grad = d_grad;
  u.grad_pstride = m->D;
  for (int i = 0; i < iters; ++i) {
    if (kind == 0) {
      if (!pyz_launch_fwd_ring(g, batch, P, st)) pyz_launch_fwd(g, batch, P, st);
    }
    else if (kind == 1) pyz_launch_bwd_data(g, batch, P, st);
    else launch_wgrad_all(m, P, d_x, d_row_idx, batch, m->ctl, u, st, nullptr);  // the launch the step uses (all layers)
  }
  PYZ_LAUNCH_CHECK();
  return PYZ_OK;
}

// ---------------------------------------------------------------- per-launch timing
int pyz_probe_begin(int max_launches) {
  if (max_launches < 1 || max_launches > (1 << 20)) return pyz_fail(PYZ_E_INVALID, "max_launches outside [1, 2^20]");
  PyzProbe &p = pyz_probe();
  if (p.on) return pyz_fail(PYZ_E_INVALID, "a probe is already open on this thread");
  while ((int)p.ev.size() < 2 * max_launches) {
    hipEvent_t e = nullptr;
    PYZ_HIP(hipEventCreate(&e));
    p.ev.push_back(e);
  }
  p.name.assign((size_t)max_launches, nullptr);
  p.cap = max_launches;
  p.n = 0;
  p.on = true;
  return PYZ_OK;
}

int pyz_probe_end(void *stream, float *h_us, char *h_names, int name_stride, int *h_n) {
  PyzProbe &p = pyz_probe();
  if (!p.on) return pyz_fail(PYZ_E_INVALID, "no probe open on this thread");
  p.on = false;
  if (!h_us || !h_n) return pyz_fail(PYZ_E_INVALID, "null pointer");
  PYZ_HIP(hipStreamSynchronize(as_stream(stream)));
  for (int i = 0; i < p.n; ++i) {
    float ms = 0.0f;
    PYZ_HIP(hipEventElapsedTime(&ms, p.ev[2 * i], p.ev[2 * i + 1]));
    h_us[i] = ms * 1e3f;
    if (h_names && name_stride > 1) {
      const char *nm = p.name[i] ? p.name[i] : "";
      const bool wrapped = nm[0] == '(';  // "(k_head_rows<4, 12>)": the launch site wraps template kernels in parentheses
      char *dst = h_names + (size_t)i * name_stride;
      std::snprintf(dst, (size_t)name_stride, "%s", nm + (wrapped ? 1 : 0));
      const size_t len = std::strlen(dst);
      if (wrapped && len > 0 && dst[len - 1] == ')') dst[len - 1] = '\0';
    }
  }
  *h_n = p.n;
  return PYZ_OK;
}

int pyz_last_run_info(const pyz_mlp *m, int32_t *h_graph_steps, int32_t *h_eager_steps, int32_t *h_graph_launches) {
  if (!m) return pyz_fail(PYZ_E_INVALID, "null plan");
  if (h_graph_steps) *h_graph_steps = m->run_graph_steps;
  if (h_eager_steps) *h_eager_steps = m->run_eager_steps;
  if (h_graph_launches) *h_graph_launches = m->run_graph_launches;
  return PYZ_OK;
}

// ---------------------------------------------------------------- non-finite sentinel
int pyz_check_finite(pyz_mlp *m, void *stream) {
  if (!m) return pyz_fail(PYZ_E_INVALID, "null plan");
  hipStream_t st = as_stream(stream);
  int n = 0;
  PYZ_HIP(hipMemcpyAsync(&n, m->nonfinite, sizeof n, hipMemcpyDeviceToHost, st));
  PYZ_HIP(hipMemsetAsync(m->nonfinite, 0, sizeof n, st));
  PYZ_HIP(hipStreamSynchronize(st));
  if (n > 0) return pyz_fail(PYZ_E_NAN, "%d step(s) since the last check produced a NaN / Inf loss", n);
  return PYZ_OK;
}

// ---------------------------------------------------------------- memory of callers without a device allocator
int pyz_malloc(size_t bytes, void **d_out) {
  if (!d_out || bytes == 0) return pyz_fail(PYZ_E_INVALID, "bad allocation request");
  *d_out = nullptr;
  PYZ_HIP(hipMalloc(d_out, bytes));
  return PYZ_OK;
}

int pyz_free(void *d_ptr) {
  if (d_ptr) PYZ_HIP(hipFree(d_ptr));
  return PYZ_OK;
}

int pyz_upload(void *d_dst, const void *h_src, size_t bytes, void *stream) {
  if (!d_dst || !h_src) return pyz_fail(PYZ_E_INVALID, "null pointer");
  PYZ_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
  PYZ_HIP(hipStreamSynchronize(as_stream(stream)));  // the host buffer is the caller's again on return
  return PYZ_OK;
}

int pyz_download(void *h_dst, const void *d_src, size_t bytes, void *stream) {
  if (!h_dst || !d_src) return pyz_fail(PYZ_E_INVALID, "null pointer");
  PYZ_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
  PYZ_HIP(hipStreamSynchronize(as_stream(stream)));
  return PYZ_OK;
}

int pyz_wait_flags(const uint64_t *d_flags, int n, uint64_t value, int spin_limit, int *d_fail, void *stream) {
  if (!d_flags) return pyz_fail(PYZ_E_INVALID, "null device pointer");
  if (n < 1 || n > 64) return pyz_fail(PYZ_E_INVALID, "flag count %d outside [1, 64]", n);
  PYZ_LAUNCH(k_wait_flags, dim3(1), dim3(64), 0, as_stream(stream), reinterpret_cast<const unsigned long long *>(d_flags), n,
             (unsigned long long)value, spin_limit, d_fail);
  PYZ_LAUNCH_CHECK();
  return PYZ_OK;
}

int pyz_sync(void *stream) {
  PYZ_HIP(hipStreamSynchronize(as_stream(stream)));
  return PYZ_OK;
}

int pyz_debug_mfma_f64_layout(int32_t *h_out512) {
  if (!h_out512) return pyz_fail(PYZ_E_INVALID, "null pointer");
  int *d = nullptr;
  PYZ_HIP(hipMalloc((void **)&d, 512 * sizeof(int)));
  PYZ_LAUNCH(k_probe_mfma_f64, dim3(1), dim3(64), 0, nullptr, d);
  const hipError_t e = hipMemcpy(h_out512, d, 512 * sizeof(int), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  PYZ_HIP(e);
  return PYZ_OK;
}

int pyz_debug_stamps(uint64_t *h_out, int64_t n_words) {
#ifdef PYZ_STAMPS
  const int64_t total = (int64_t)PYZ_STAMP_KERNELS * PYZ_STAMP_BLOCKS * PYZ_STAMP_WAVES * PYZ_STAMP_SLOTS * 2;
  if (!h_out || n_words < total) return pyz_fail(PYZ_E_INVALID, "need %lld words", (long long)total);
  PYZ_HIP(hipDeviceSynchronize());
  PYZ_HIP(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(pyz_dbg_buf), sizeof(unsigned long long) * total));
  return PYZ_OK;
#else
  (void)h_out;
  (void)n_words;
  return pyz_fail(PYZ_E_INVALID, "not a PYZ_STAMPS build");
#endif
}

}  // extern "C"
