// pyz_common.h -- shared host/device definitions of the gfx950 backend.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/pyz.h"

#define PYZ_MAX_LAYERS 16

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Per-step scalars.  They live in device memory so that a captured hipGraph of
// one step can be replayed for every step: the last kernel of a step advances
// the block from the per-run tables (k_advance_ctl).
struct StepCtl {
  int32_t batch;       // rows in this step's batch (ragged last batch allowed)
  float lr;            // learning rate of this step
  long long n;         // optimizer step count (moment divisor / RNG step)
  long long row_off;   // offset of this step's rows inside the row-index table
  int32_t i;           // index of the step inside the current run
  int32_t slot0;       // first loss / row-table slot of the current run
  int32_t n_run;       // steps of the current device-resident run (0: a single eager step); step i + 1 exists iff i + 1 < n_run
};

// ---------------------------------------------------------------- in-kernel stamps (diagnostic build only)
// -DPYZ_STAMPS builds libpyz_stamps.so: wave 0 of the first 256 workgroups records
// {s_memtime, s_memrealtime} at phase boundaries.  The shipped library has no stamps.
#ifdef PYZ_STAMPS
#define PYZ_STAMP_KERNELS 6   // 0 forward, 1 head, 2 weight gradients, 3 k_svgd_gs, 4 k_svgd_gram_tile, 5 k_svgd_update_tile
#define PYZ_STAMP_BLOCKS 256
#define PYZ_STAMP_WAVES 16
#define PYZ_STAMP_SLOTS 8
__device__ unsigned long long pyz_dbg_buf[PYZ_STAMP_KERNELS][PYZ_STAMP_BLOCKS][PYZ_STAMP_WAVES][PYZ_STAMP_SLOTS][2];
#define PYZ_STAMP(kid, slot)                                                                   \
  do {                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < PYZ_STAMP_BLOCKS && blockIdx.y == 0) {        \
      unsigned long long t0_, t1_;                                                             \
      asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_), "=s"(t1_)::"memory"); \
      pyz_dbg_buf[kid][blockIdx.x][threadIdx.x >> 6][slot][0] = t0_;                          \
      pyz_dbg_buf[kid][blockIdx.x][threadIdx.x >> 6][slot][1] = t1_;                          \
    }                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                         \
  } while (0)
// lap counters kept in registers (`lap`: unsigned long long[16], [15] = the last reading): cycles since the previous lap are
// added to lap[slot]; every wave runs it (scalar), the kernel decides which wave writes its counters out
#define PYZ_LAP(lap, slot)                                                                   \
  do {                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    unsigned long long t_;                                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
    (lap)[slot] += t_ - (lap)[15];                                                           \
    (lap)[15] = t_;                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                       \
  } while (0)
#else
#define PYZ_STAMP(kid, slot)
#define PYZ_LAP(lap, slot)
#endif

// wave index inside the workgroup as a SCALAR: threadIdx.x >> 6 is wave-uniform, but the compiler
// only knows it once it has gone through readfirstlane; everything derived from it (the wave's slice of
// the reduction, loop bounds, clamps) then lives in SGPRs and the loops branch on scalar compares
// instead of exec-masked vector compares
__device__ __forceinline__ int pyz_wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Eager step entry points do not spend a launch on the step scalars: the FIRST kernel of the step gets
// them by value (`init`), uses them, and its first thread publishes them in the device StepCtl for the
// kernels behind it (which start after that kernel has ended).
__device__ __forceinline__ StepCtl pyz_ctl_first(const StepCtl *ctl, const StepCtl &init, const int on) {
  if (on && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *const_cast<StepCtl *>(ctl) = init;
  // Both sources are wave-uniform and should stay scalar.  The by-value fields pass through an empty asm:
  // without it the compiler rewrites "select between two loaded values" as "load through a selected
  // address", and that address is a flat VGPR pointer -- every field becomes a vector load.
  int ib = init.batch, ii = init.i, is0 = init.slot0, inr = init.n_run;
  float il = init.lr;
  long long in_ = init.n, iro = init.row_off;
  asm volatile("" : "+s"(ib), "+s"(ii), "+s"(is0), "+s"(il), "+s"(in_), "+s"(iro), "+s"(inr));
  StepCtl c;
  c.batch = on ? ib : ctl->batch;
  c.lr = on ? il : ctl->lr;
  c.n = on ? in_ : ctl->n;
  c.row_off = on ? iro : ctl->row_off;
  c.i = on ? ii : ctl->i;
  c.slot0 = on ? is0 : ctl->slot0;
  c.n_run = on ? inr : ctl->n_run;
  return c;
}

// Stores of kernel outputs that another kernel reads next (activations, deltas, optimizer state).  wt != 0: write-through
// (sc1) -- the bytes leave L2 while the kernel still runs.  Plain stores stay dirty in the XCD's L2 and the END of the
// kernel waits for their write-back: on the short kernels of a single-chain step that wait is on the critical path
// (C2: 25.2 -> 24.45 us per step with all three kernels storing write-through).  Launches with many particles keep
// plain stores (their consumers re-read the lines from the same L2).  wt is wave-uniform.
__device__ __forceinline__ void pyz_st(float *p, const float v, const int wt) {
  if (wt)
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else
    *p = v;
}

// ---------------------------------------------------------------- host errors
inline std::string &pyz_err_slot() {
  static thread_local std::string s;
  return s;
}

inline int pyz_fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  pyz_err_slot() = buf;
  return code;
}

#define PYZ_HIP(call)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return pyz_fail(e_ == hipErrorOutOfMemory ? PYZ_E_OOM : PYZ_E_HIP, "%s failed: %s (%s:%d)", \
                      #call, hipGetErrorString(e_), __FILE__, __LINE__);                     \
  } while (0)

#define PYZ_LAUNCH_CHECK()                                                                    \
  do {                                                                                        \
    hipError_t e_ = hipGetLastError();                                                        \
    if (e_ != hipSuccess)                                                                     \
      return pyz_fail(PYZ_E_HIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(e_),   \
                      __FILE__, __LINE__);                                                    \
  } while (0)

// ---------------------------------------------------------------- per-launch timing (measurement only)
// Between pyz_probe_begin and pyz_probe_end every kernel launch of the calling thread goes out through
// hipExtLaunchKernelGGL with a start and a stop event of its own: the pair reads the dispatch's begin / end
// timestamps (what rocprofv3 --kernel-trace reports as the kernel's duration), with no extra packet between
// two kernels of the pipeline.  Off (the normal state): a plain launch.
struct PyzProbe {
  std::vector<hipEvent_t> ev;       // 2 per launch
  std::vector<const char *> name;   // the kernel expression of the launch site
  int n = 0, cap = 0;
  bool on = false;
};
inline PyzProbe &pyz_probe() {
  static thread_local PyzProbe p;
  return p;
}
#define PYZ_LAUNCH(kern, grid, block, lds, st, ...)                                                              \
  do {                                                                                                           \
    PyzProbe &pp_ = pyz_probe();                                                                                 \
    if (pp_.on && pp_.n < pp_.cap) {                                                                             \
      hipExtLaunchKernelGGL(kern, grid, block, lds, st, pp_.ev[2 * pp_.n], pp_.ev[2 * pp_.n + 1], 0, __VA_ARGS__); \
      pp_.name[pp_.n++] = #kern;                                                                                 \
    } else {                                                                                                     \
      hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                               \
    }                                                                                                            \
  } while (0)

// ---------------------------------------------------------------- the plan
#define PYZ_GRAPH_CHUNKS 8  // captured step graphs: slot 0 = G steps, the others = remainders by exact length (LRU)
struct pyz_mlp {
  int L = 0;
  int dims[PYZ_MAX_LAYERS + 1] = {0};
  int acts[PYZ_MAX_LAYERS] = {0};
  long long w_off[PYZ_MAX_LAYERS] = {0};  // offset of layer l's kernel (bias follows) in the flat vector
  int loss = 0;
  int max_batch = 0, max_p = 0;
  long long D = 0;
  size_t ws_bytes = 0;
  // device workspace (library-owned)
  float *act[PYZ_MAX_LAYERS] = {nullptr};    // output of layer l: (P, max_batch, dims[l+1])
  float *delta[PYZ_MAX_LAYERS] = {nullptr};  // d loss / d pre-activation of layer l, same shape
  float *xb = nullptr;                       // (max_batch, dims[0]) contiguous copy of the gathered batch rows
  float *grad = nullptr;                     // (P, D)
  float *grad2 = nullptr;                    // (P, D) second scratch (HMC momentum, SVGD phi)
  float *qsave = nullptr;                    // (P, D) HMC snapshot
  double *part = nullptr;                    // reduction partials
  int part_len = 0;
  int cur_nblk = 0;                          // number of loss partials the last loss launch wrote per particle
  float *scal = nullptr;                     // small device scalars
  StepCtl *ctl = nullptr;                    // device StepCtl
  StepCtl pend{};                            // eager steps: the scalars of the step about to be launched ...
  bool pend_on = false;                      // ... not yet on the device: the first kernel of the step publishes them
  int32_t *tab_bs = nullptr;                 // per-run tables (device)
  float *tab_lr = nullptr;
  int tab_cap = 0;
  float *h_pinned = nullptr;                 // pinned host staging
  // device-resident runs: one captured graph per chunk length (everything baked into them goes into graph_key)
  hipGraph_t graph[PYZ_GRAPH_CHUNKS] = {nullptr};
  hipGraphExec_t graph_exec[PYZ_GRAPH_CHUNKS] = {nullptr};
  int graph_len[PYZ_GRAPH_CHUNKS] = {0};
  unsigned long long graph_use[PYZ_GRAPH_CHUNKS] = {0}, graph_clock = 0;   // last use of a slot (eviction order)
  unsigned long long graph_key = 0;
  // what the last pyz_*_run call did: steps inside replayed graphs, eager steps, graph launches
  int run_graph_steps = 0, run_eager_steps = 0, run_graph_launches = 0;
  int *nonfinite = nullptr;                  // device counter: steps whose loss was NaN / Inf (pyz_check_finite)
  // who wrote `grad` / `scal` last: pyz_svgd_sweep / pyz_svgd_combine consume what pyz_svgd_gradients left there and refuse
  // anything else (pyz_hmc_step, pyz_bbb_step, ... use the same buffers)
  int grad_owner = 0;                        // 0 nobody, 1 pyz_svgd_gradients for grad_rows particles, 2 another entry point
  int grad_rows = 0;
  // the snapshot whose kernel matrix (K rows of the local particles, their sums, the bandwidth) stands in the plan's
  // float64 scratch: written by pyz_svgd_kernel_matrix, consumed by pyz_svgd_combine, dropped by every other user of the scratch
  bool km_valid = false;
  const float *km_all = nullptr;
  int km_total = 0, km_row0 = 0, km_local = 0;
  float km_gamma = 0.0f;
};
