// pyz_gemm_ring.h -- Dense forward for MID-SIZE launches: 32-row blocks, operands through an LDS-DMA ring.
//
// Replaces the same Keras Dense forward as pyz_gemm.h (call sites SVGD.py:106, BBB.py:144, SGLD.py:55) for the
// launches that neither kernel there serves well: a few particles of a layer a few hundred wide (one rank's share of
// the sharded SVGD step: 8 - 16 particles of 784 -> 200 at batch 1024).  k_dense_fwd (one wave per 32 x 32 tile,
// operands straight from L2) pulls 200 KB per tile through one CU's vector-memory path and is bound by it;
// k_dense_fwd_lds (128 x 224 tiles) has too few tiles to fill 256 CUs below ~32 particles.
//
// Here a workgroup owns 32 batch rows x ALL N columns of one particle, i.e. P * batch / 32 workgroups:
//   * both operands arrive by LDS-DMA (buffer_load / global_load ... lds, 16 bytes per lane, no VGPR in between) into a
//     ring of NBUF = 4 slabs of 16 SUB reduction steps; three slabs are in flight while one is consumed; one raw
//     s_barrier per slab, counted s_waitcnt vmcnt(N) (cdna_hip_programming.md section 5, "Pipelining across barriers");
//   * the B slab (rows k0 .. of [W; b]) is copied as it stands: row kk of the image holds NL floats from the
//     16-byte-aligned address at or below &W[k0 + kk][0] -- a particle's block is only 4- or 8-byte aligned in general
//     ((P, D) row-major with D % 4 != 0), `mis` floats of slack in front make every DMA source 16-byte aligned;
//     NL % 8 == 4 puts the four reduction quarters of a fragment read on disjoint banks;
//   * the A slab is 32 rows x 16 floats per sub-slab, gathered by the DMA's per-lane source address when the rows go
//     through row_idx (the workgroups of the launch share the duty of leaving the contiguous batch copy the
//     weight-gradient kernel reads: workgroup (p, rows) stores the slabs s with s % P == p);
//   * v_mfma_f32_16x16x4_f32 (exact fp32, 32 cycles): four computing waves own (row half, column group) of the 2 x NCT
//     sub-tiles of 16 x 16 -- 7 / 7 / 6 / 6 for N = 200, the shorter groups run one padding sub-tile so that every wave
//     executes the same instruction stream; a lane reads its A row's four reduction quarters with one ds_read_b128 per
//     sub-slab (reduction index k0 + 4 q + j for instruction j, lane quarter q: permuted identically for A and B) and
//     one ds_read_b32 per instruction for B;
//   * with NW = 8, four more waves (one per SIMD) do nothing but move data: a DMA instruction costs its wave ~120 cycles
//     of issue, ~480 per slab, which a computing wave cannot hide behind its own matrix instructions.
// What the measurements of round 3 said on the way (profiles/r03_ring/):
//   - every LDS read is inline asm with hand-counted waits: the compiler's wait-count pass drains EVERY LDS-DMA in flight
//     (vmcnt(0)) in front of an LDS read it sees, and around its own reads it waits lgkmcnt(0) with the next step's reads
//     just issued (one exposed LDS round trip per step);
//   - one loop body for all slabs and ONE sub-tile count: per-slab bodies behind `if (s + 1 < ns)` and a sub-tile count
//     chosen per wave were control-flow merges at which the compiler shuffled the accumulators and the fragment registers
//     (~350 cycles per slab, and it moved registers the asm reads were still filling);
//   - at the end of a slab no read is in flight: the next slab's first fragments are requested one step EARLY (beside the
//     last step's), so the drain costs nothing and no half-written register crosses the loop's back edge;
//   - the barrier that opens slab s + 1 sits in the MIDDLE of slab s: a computing wave never starts a slab with an
//     exposed LDS round trip.
// Needs K % 4 == 0, N % 4 == 0, lda % 4 == 0, 16-byte aligned input rows; checked at launch (pyz_fwd_ring_variant).
#pragma once

#include <type_traits>
#include <utility>

#include "pyz_gemm.h"

typedef __attribute__((address_space(3))) void pyz_lds_void;
typedef const __attribute__((address_space(1))) void pyz_glb_void;

// source of the A pieces past the end of a row's reduction range (K % 16 != 0): sixteen zero bytes
__device__ __attribute__((aligned(16))) float pyz_zero16[4] = {0.0f, 0.0f, 0.0f, 0.0f};

template <int N>
__device__ __forceinline__ void pyz_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS reads the COMPILER does not see (and therefore does not wait for): the caller counts them in lgkmcnt itself.
template <int OFF>
__device__ __forceinline__ float pyz_lds_read_b32(const unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ f32x4 pyz_lds_read_b128(const unsigned addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <class F, int... I>
__device__ __forceinline__ void pyz_static_for(F f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>()), ...);
}

// SUB = 16-deep sub-slabs per ring slot (per barrier): a slab is 16 SUB reduction steps.
// NW = waves per workgroup.  4: every wave issues its share of the DMA and computes its share of the tile.  8: waves 0 - 3
// compute, waves 4 - 7 only move data (one of each kind per SIMD).
// Computing wave c (of 4) owns row half c & 1 and column group c >> 1 of the two groups the NCT column sub-tiles are dealt into;
// moving wave d (of 4) issues DMA instructions d, d + 4, ... of every slab.
template <int NL, int NCT, int SUB, int NW>
struct PyzRingGeom {
  static constexpr int BK = 16, NBUF = 4;
  static constexpr int KS = BK * SUB;                     // reduction steps per slab
  static constexpr int PR = NL / 4;                       // 16-byte pieces per image row of B
  static constexpr int B_INSTR = (BK * PR + 63) / 64;     // DMA wave-instructions per B sub-slab (1 KiB each)
  static constexpr int NI1 = B_INSTR + 2;                 // + two for the 32 x 16 A sub-slab
  static constexpr int NI = SUB * NI1;                    // per slab
  static constexpr int A_OFF = B_INSTR * 1024;            // inside a sub-slab image
  static constexpr int SUBSLOT = A_OFF + 2048;
  static constexpr int SLOT = SUB * SUBSLOT;
  static constexpr int LDS_BYTES = NBUF * SLOT + 1024;    // (+ slack behind the last slot: padding sub-tiles read past their row)
  static constexpr int CT = (NCT + 1) / 2;                // column sub-tiles per computing wave (the second group may have one fewer)
  static constexpr int T = 4 * SUB;                       // matrix-instruction steps per slab (CT instructions each)
  static_assert(NW == 4 || NW == 8, "waves per workgroup");
  static_assert(NI >= 4, "every moving wave issues DMA instructions");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS of a CU");
  static_assert(NBUF == 4, "the wait counts assume two slabs in flight behind the one awaited");
  static_assert(NL % 8 == 4, "row stride of the B image: the four reduction quarters on disjoint banks");
  static_assert(NCT >= 2 && NCT <= 14 && 16 * NCT <= NL + 12, "column sub-tiles");
};

// Reads younger than B(i, ct) that have been issued when matrix instruction (i, ct) is about to issue (the immediate of
// its s_waitcnt lgkmcnt): the schedule of fragment reads inside a slab is
//   behind instruction (i, ct), i < T - 2 :  [A(i + 1) when ct == 0 and step i + 1 opens a sub-slab,]  B(i + 1, ct)
//   behind instruction (T - 2, ct)        :  B(T - 1, ct),  [A'(0) when ct == 0,]  B'(0, ct)      (' = the NEXT slab)
//   behind instruction (T - 1, ct)        :  nothing; the slab ends with everything landed (lgkmcnt(0), free by then)
__host__ __device__ constexpr int pyz_ring_younger(const int i, const int ct, const int CT, const int T) {
  if (i == 0) return -1;                                  // group(0) landed in the previous slab: no wait
  if (i == T - 1) return (ct == 0 ? 1 : 0) + 1 + 2 * (CT - 1 - ct);
  const int a_next = ((i + 1) % 4 == 0) && ct >= 1 ? 1 : 0;                  // A(i + 1) went out behind instruction (i, 0)
  if (i == T - 2) return (CT - 1 - ct) + 2 * ct + (ct >= 1 ? 1 : 0);         // (step T - 1 opens no sub-slab: a_next = 0)
  return CT - 1 + a_next;
}

template <int NL, int NCT, int SUB, int NW>
__global__ void __launch_bounds__(64 * NW) k_dense_fwd_ring(DenseArgs g) {
  using G = PyzRingGeom<NL, NCT, SUB, NW>;
  constexpr int BK = G::BK, NBUF = G::NBUF, KS = G::KS, PR = G::PR, B_INSTR = G::B_INSTR, NI = G::NI, NI1 = G::NI1,
                A_OFF = G::A_OFF, SLOT = G::SLOT, SUBSLOT = G::SUBSLOT, CT = G::CT, T = G::T, TB = G::T / 2;
  // ONE LDS object; the compiler sees no load from it (every read below is inline asm), so its wait-count pass has
  // nothing to protect from the DMA in flight
  __shared__ __attribute__((aligned(16))) unsigned char ring[G::LDS_BYTES];
  const int t = threadIdx.x, w = pyz_wave_id(), l = t & 63;
  const int c16 = l & 15, q = l >> 4;
  const StepCtl ctl = pyz_ctl_first(g.ctl, g.init, g.init_on);
  const int batch = ctl.batch;
  const int K = g.K, N = g.N, P = g.n_part;
  if (g.gate && g.gate->n % g.gate_mod == 0) return;      // (uniform) a gated launch on a step it skips
  const int row_tiles = (g.grid_rows + 31) >> 5, n_cg = g.n_cgrp, n_ks = g.k_split > 1 ? g.k_split : 1;
  const int n_tiles_1 = row_tiles * P * n_cg, n_tiles = n_tiles_1 * n_ks;
  if ((int)blockIdx.x >= n_tiles) return;                 // padding of the launch
  // particle-major tile ids: an XCD's contiguous range is (part of) one particle's row blocks, its L2 holds that particle's
  // weights.  (Blocks of 2 / 4 / 8 particles x fewer row blocks per XCD -- every input line then asked for by several
  // workgroups of the XCD -- measured the same: 39.3 - 39.7 us at 8 particles.)  Placement only changes speed.
  // A layer wider than the 200 columns a workgroup takes is cut into column groups: the groups of a row block are
  // neighbours in the tile order (same XCD: the block's input rows come out of its L2 for all but the first).
  // Split reduction (a single chain's launch has too few row blocks to fill the chip): the range of slabs is the SLOW index
  // of the tile order -- an XCD's contiguous range of tiles is then the row blocks of one range, which share its rows of
  // [W; b] in that XCD's L2.
  const int tile_ks = pyz_xcd_remap(blockIdx.x, n_tiles);
  const int ksp = tile_ks / n_tiles_1, tile_all = tile_ks - ksp * n_tiles_1;
  const int tile = tile_all / n_cg, cgp = tile_all - tile * n_cg, col0 = cgp * g.cgrp_w;
  const int p = tile / row_tiles, m0 = (tile - p * row_tiles) * 32;
  if (m0 >= batch) return;                                // uniform
  const bool moves = NW == 4 || w >= 4, computes = NW == 4 || w < 4;   // roles (scalar)
  const int wq = w & 3;                                   // index among the 4 waves of its role
  const int rh = wq & 1, cg = wq >> 1;                    // a computing wave's row half and column group
  const unsigned ring_lds = (unsigned)(uintptr_t)(pyz_lds_void *)ring;

  // ---- DMA sources.  Per sub-slab, instruction ids 0 .. B_INSTR - 1 copy B, B_INSTR and B_INSTR + 1 the 32 x 16 block of A;
  //      a slab has NI = SUB NI1 of them, moving wave wq issues ids wq, wq + 4, ... in this order, for every slab.
  constexpr int IPW = (NI + 3) / 4;
  const int n_mine = (NI - wq + 3) / 4;                   // instructions of this wave per slab (scalar): IPW or IPW - 1
  const float *wl = g.theta + (long long)p * g.theta_pstride + g.w_off;
  const int mis = (int)((reinterpret_cast<uintptr_t>(wl) & 15u) >> 2);      // floats between the aligned base and W[0][0]
  const float *wl_al = wl - mis;
  // bytes from the aligned base to the end of the (P, D) parameter matrix: loads past it return zero
  const long long b_bytes = ((long long)(P - p) * g.theta_pstride - g.w_off + mis) * 4;
  const __amdgpu_buffer_rsrc_t rB =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wl_al), 0, (int)(b_bytes < 0x7fffffffLL ? b_bytes : 0x7fffffffLL), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;                   // stays out of range under the per-slab advance (D * 4 < 2^31)
  // per instruction of this wave: a B piece (byte offset from the aligned base, advanced per slab) or an A piece (this
  // lane's source address, advanced per slab; koffA = its reduction index inside the slab, for the tail of K)
  unsigned voffB[IPW];
  const float *srcA[IPW];
  int koffA[IPW];
#pragma unroll
  for (int u = 0; u < IPW; ++u) {
    const int id = wq + 4 * u, sub = id / NI1, id1 = id - sub * NI1;   // sub-slab and instruction inside it
    voffB[u] = OOB;
    srcA[u] = nullptr;
    koffA[u] = 0;
    if (id < NI && moves) {
      if (id1 < B_INSTR) {
        const int tt = 64 * id1 + l, kk = tt / PR, c4 = tt - kk * PR;
        voffB[u] = kk < BK ? (unsigned)(((BK * sub + kk) * N + col0) * 4 + c4 * 16) : OOB;
      } else {
        const int row = 16 * (id1 - B_INSTR) + (l >> 2);
        koffA[u] = BK * sub + 4 * (l & 3);
        const int m = min(m0 + row, batch - 1);
        long long grow = m;
        if (g.row_idx) grow = g.row_idx[ctl.row_off + m];
        srcA[u] = g.in + (long long)p * g.in_pstride + grow * g.lda + koffA[u];
      }
    }
  }
  const unsigned slabB = (unsigned)(KS * N * 4);          // bytes per slab of B
  const int ns_all = (K + KS - 1) / KS;
  const int s_lo = (int)((long long)ns_all * ksp / n_ks), ns = (int)((long long)ns_all * (ksp + 1) / n_ks) - s_lo;   // this workgroup's slabs
  if (s_lo > 0) {
#pragma unroll
    for (int u = 0; u < IPW; ++u) {
      if (voffB[u] != OOB) voffB[u] += (unsigned)s_lo * slabB;
      if (srcA[u]) srcA[u] += (long long)s_lo * KS;
    }
  }

  auto issue = [&](const int s) {                         // the DMA of slab s into ring slot s % NBUF
    unsigned char *base = ring + (s & (NBUF - 1)) * SLOT;
#pragma unroll
    for (int u = 0; u < IPW; ++u) {
      const int id = wq + 4 * u, sub = id / NI1, id1 = id - sub * NI1;
      if (id < NI) {
        if (id1 < B_INSTR) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (pyz_lds_void *)(base + sub * SUBSLOT + id1 * 1024), 16, (int)voffB[u], 0, 0, 0);
          voffB[u] += slabB;
        } else {
          const float *src = (KS * (s_lo + s) + koffA[u] < K) ? srcA[u] : pyz_zero16;
          __builtin_amdgcn_global_load_lds((pyz_glb_void *)src, (pyz_lds_void *)(base + sub * SUBSLOT + id1 * 1024), 16, 0, 0);
          srcA[u] += KS;
        }
      }
    }
  };

  // bias of this lane's columns (consumed in the epilogue); a computing wave of the second column group may own one
  // sub-tile fewer than CT: its last one is padding (computed like the others, never stored)
  const int ct_first = cg * (NCT / 2) + min(cg, NCT % 2);
  const int n_ct = NCT / 2 + (cg < NCT % 2 ? 1 : 0);
  float bias[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int col = 16 * (ct_first + ct) + c16;
    bias[ct] = wl[(long long)K * N + min(col0 + col, N - 1)];
  }
  f32x4 acc[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) acc[ct] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  // ---- The pipeline.  Slab s lives in ring slot s % 4.  Barrier B_s makes slab s readable (every moving wave has waited for
  //      its share of it) and tells the moving waves that nobody reads slab s - 2 any more.  A computing wave meets B_{s+1}
  //      in the MIDDLE of slab s; behind it the moving waves request slab s + 3 into the slot of slab s - 1.
  const bool copies = g.gather_out != nullptr && moves && wq < 2 && cgp == 0 && n_ks == 1;   // the two waves that store the batch copy (scalar)
#ifdef PYZ_STAMPS   // diagnostic build: cycles of this wave per phase, summed over the slabs
  unsigned long long ph[4] = {0, 0, 0, 0}, ph_t = 0;
#define PYZ_RING_PHASE(i)                                         \
  do {                                                            \
    __builtin_amdgcn_sched_barrier(0);                            \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    ph[i] += now_ - ph_t;                                         \
    ph_t = now_;                                                  \
    __builtin_amdgcn_sched_barrier(0);                            \
  } while (0)
  ph_t = __builtin_amdgcn_s_memtime();
  const unsigned long long ph_start = ph_t;
#else
#define PYZ_RING_PHASE(i)
#endif
  // this wave's DMA of all but its `groups` youngest slabs has landed (scalar branches to the immediate of s_waitcnt)
  auto wait_own = [&](const int groups) {
    if (groups >= 2) {
      if (n_mine == IPW) pyz_wait_vmcnt<2 * IPW>();
      else pyz_wait_vmcnt<2 * (IPW - 1)>();
    } else if (groups == 1) {
      if (n_mine == IPW) pyz_wait_vmcnt<IPW>();
      else pyz_wait_vmcnt<IPW - 1>();
    } else {
      pyz_wait_vmcnt<0>();
    }
  };
  // between the halves of slab s: slab s + 1 becomes readable, slab s + 3 is requested
  auto sync_point = [&](const int s) {
    PYZ_RING_PHASE(3);
    if (moves) {
      wait_own(s + 2 < ns ? 1 : 0);                      // slab s + 1 (requested three sync points ago) has landed
      if (copies && (s % P) == p) {                      // this workgroup's share of the batch copy: rows tc / 4, piece tc % 4
        const int tc = (wq << 6) | l, row = tc >> 2, pc = tc & 3;
        const unsigned src = ring_lds + (unsigned)((s & (NBUF - 1)) * SLOT + A_OFF + row * 64 + pc * 16);
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub) {
          f32x4 v;
          asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(src), "n"(sub * SUBSLOT));
          const int k = KS * (s_lo + s) + BK * sub + 4 * pc;
          if (m0 + row < batch && k < K) *reinterpret_cast<f32x4 *>(g.gather_out + (long long)(m0 + row) * K + k) = v;
        }
      }
    }
    PYZ_RING_PHASE(0);
    if (s + 1 < ns) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();                      // B_{s+1} (no lgkmcnt wait: the fragment reads in flight are of slab s)
      asm volatile("" ::: "memory");
    }
    PYZ_RING_PHASE(1);
#ifndef PYZ_RING_NODMA      // (diagnostic: fragment reads + matrix instructions alone, on whatever the prologue left in LDS)
    if (moves && s + NBUF - 1 < ns) issue(s + NBUF - 1);
#endif
    PYZ_RING_PHASE(2);
  };

  // ---- prologue: three slabs in flight, B_0
  if (moves) {
    issue(0);
    if (1 < ns) issue(1);
    if (2 < ns) issue(2);
    wait_own(min(2, ns - 1));
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  if (!computes) {          // a wave that only moves data: one sync point per slab
    for (int s = 0; s < ns; ++s) sync_point(s);
#ifdef PYZ_STAMPS
    if (l == 0 && blockIdx.x < PYZ_STAMP_BLOCKS) {
      for (int i = 0; i < 4; ++i) pyz_dbg_buf[0][blockIdx.x][w][i][0] = ph[i];
      pyz_dbg_buf[0][blockIdx.x][w][4][0] = __builtin_amdgcn_s_memtime() - ph_start;
      pyz_dbg_buf[0][blockIdx.x][w][5][0] = (unsigned long long)ns;
    }
#endif
    return;
  }
#ifdef PYZ_RING_NOCOMPUTE   // diagnostic: the data movement alone
  for (int s = 0; s < ns; ++s) sync_point(s);
#else
  // byte addresses (LDS) of this lane's fragments inside slot 0
  const unsigned a_rd = ring_lds + (unsigned)(A_OFF + (16 * rh + c16) * 64 + q * 16);            // A pieces of sub-slab 0
  const unsigned b_rd = ring_lds + (unsigned)((4 * q * NL + mis + 16 * ct_first + c16) * 4);     // B[4 q][first column] of sub-slab 0
  f32x4 av[SUB], avn;                                     // A fragments of this slab's sub-slabs; of the next slab's first
  float bf[2][CT], bfn[CT];                               // B fragments of the current / next step; of the next slab's first step
  // group(0) of slab 0
  avn = pyz_lds_read_b128<0>(a_rd);
  pyz_static_for([&](auto c_c) { bfn[decltype(c_c)::value] = pyz_lds_read_b32<64 * decltype(c_c)::value>(b_rd); },
                 std::make_integer_sequence<int, CT>());
#pragma unroll
  for (int sub = 1; sub < SUB; ++sub) av[sub] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) bf[1][ct] = 0.0f;
  for (int s = 0; s < ns; ++s) {
    // the next slab's first fragments (and everything else) have landed: take them over
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bfn[ct]));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(avn));
    av[0] = avn;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) bf[0][ct] = bfn[ct];
    const unsigned la = a_rd + (unsigned)((s & (NBUF - 1)) * SLOT), lb = b_rd + (unsigned)((s & (NBUF - 1)) * SLOT);
    const unsigned na = a_rd + (unsigned)(((s + 1) & (NBUF - 1)) * SLOT), nb = b_rd + (unsigned)(((s + 1) & (NBUF - 1)) * SLOT);
    pyz_static_for(
        [&](auto i_c) {
          constexpr int i = decltype(i_c)::value;
          if constexpr (i == TB) sync_point(s);
          pyz_static_for(
              [&](auto c_c) {
                constexpr int ct = decltype(c_c)::value;
                constexpr int younger = pyz_ring_younger(i, ct, CT, T);
                // the operands of this instruction have arrived (tying the wait to them keeps the instruction behind it)
                if constexpr (younger >= 0) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bf[i & 1][ct]), "+v"(av[i >> 2]) : "n"(younger));
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i >> 2][i & 3], bf[i & 1][ct], acc[ct], 0, 0, 0);
                if constexpr (i + 1 < T) {
                  if constexpr (((i + 1) & 3) == 0 && ct == 0) av[(i + 1) >> 2] = pyz_lds_read_b128<((i + 1) >> 2) * SUBSLOT>(la);
                  bf[(i + 1) & 1][ct] = pyz_lds_read_b32<((i + 1) >> 2) * SUBSLOT + ((i + 1) & 3) * NL * 4 + 64 * ct>(lb);
                }
                if constexpr (i == T - 2) {               // the next slab's group(0), one step early (harmless past the last slab)
                  if constexpr (ct == 0) avn = pyz_lds_read_b128<0>(na);
                  bfn[ct] = pyz_lds_read_b32<64 * ct>(nb);
                }
                __builtin_amdgcn_sched_barrier(0);        // keep the read / instruction interleave as written
              },
              std::make_integer_sequence<int, CT>());
        },
        std::make_integer_sequence<int, T>());
  }
  // (the reads issued beside the last slab's step T - 2 belong to a slab that does not exist: they must have landed
  // before their registers are used for anything else)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#endif

#ifdef PYZ_STAMPS
  if (l == 0 && blockIdx.x < PYZ_STAMP_BLOCKS) {
    for (int i = 0; i < 4; ++i) pyz_dbg_buf[0][blockIdx.x][w][i][0] = ph[i];
    pyz_dbg_buf[0][blockIdx.x][w][4][0] = __builtin_amdgcn_s_memtime() - ph_start;
    pyz_dbg_buf[0][blockIdx.x][w][5][0] = (unsigned long long)ns;
  }
#endif
  // ---- epilogue: D[r] of an instruction = row 4 (lane / 16) + r, column lane % 16
  float *op = g.out + (long long)p * g.out_pstride;
  const int act = g.act, wt = g.wt;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    if (ct >= n_ct) break;         // uniform
    const int col = 16 * (ct_first + ct) + c16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int mm = m0 + 16 * rh + 4 * q + r;
      if (mm < batch && col < g.cgrp_w && col0 + col < N) {
        if (n_ks > 1) pyz_st(g.part_out + (long long)ksp * g.part_stride + (long long)mm * N + col0 + col, acc[ct][r], wt);
        else pyz_st(op + (long long)mm * N + col0 + col, pyz_act(acc[ct][r] + bias[ct], act), wt);
      }
    }
  }
}

// the launches this kernel takes: a first or hidden layer of supported width, more than one particle's worth of
// 32-row blocks (at least PYZ_FWD_RING_MINWG of them), operands aligned as the DMA needs
static inline int pyz_fwd_ring_variant(const DenseArgs &g, int grid_batch, int P) {
  static const int on = pyz_env_int("PYZ_FWD_RING", 1);
  static const int min_wg = pyz_env_int("PYZ_FWD_RING_MINWG", 192);
  static const int max_wg = pyz_env_int("PYZ_FWD_RING_MAXWG", 1024);
  if (!on) return 0;
  const int n_cg = g.N > 200 ? (g.N + 199) / 200 : 1;      // column groups of 200
  const long long wgs = (long long)((grid_batch + 31) / 32) * P * n_cg;
  if (wgs < min_wg || wgs > max_wg) return 0;
  if (g.K % 4 || g.N % 4 || g.lda % 4 || g.w_off % 4 || (P > 1 && g.in_pstride % 4)) return 0;
  if ((reinterpret_cast<uintptr_t>(g.in) & 15) || (g.gather_out && (reinterpret_cast<uintptr_t>(g.gather_out) & 15))) return 0;
  if ((reinterpret_cast<uintptr_t>(g.theta) & 3)) return 0;
  if (g.N > 192 && g.N <= 200) return 1;   // <204, 13>
  // ... per column group of 200 (784 -> 400 -> 400: the 6 000-row validation forward of C4).  Opt-in (PYZ_FWD_RING_WIDE=1, read
  // per call): measured equal to k_dense_fwd there (65.6 + 39.6 against 68.3 + 36.6 us) -- 376 workgroups of 32 rows x 200
  // columns sit two to a CU on 120 CUs and one on the rest, the launch lasts as long as the pairs.
  if (g.N > 200 && g.N % 200 == 0 && g.N <= 1600 && pyz_env_int("PYZ_FWD_RING_WIDE", 0)) return 1;
  return 0;
}

template <int NL, int NCT, int SUB, int NW>
static inline void pyz_launch_fwd_ring_as(const DenseArgs &g, int grid_batch, int P, hipStream_t st) {
  DenseArgs a = g;
  a.n_part = P;
  a.grid_rows = grid_batch;
  a.n_cgrp = g.N > 200 ? (g.N + 199) / 200 : 1;
  a.cgrp_w = g.N > 200 ? 200 : g.N;
  const long long wgs = (long long)((grid_batch + 31) / 32) * P * a.n_cgrp;
  PYZ_LAUNCH((k_dense_fwd_ring<NL, NCT, SUB, NW>), dim3((unsigned)((wgs + 7) / 8 * 8)), dim3(64 * NW), 0, st, a);
}

// the split-reduction form for a single chain's hidden layer of <= 200 units (C2: 1024 x 784 -> 200 is 32 row blocks; cut
// into k_split ranges of slabs it is 256 workgroups): caller checked the shape (pyz_fwd_ring_ksplit_ok) and set part_out
static inline bool pyz_fwd_ring_ksplit_ok(const DenseArgs &g, int grid_batch, int P) {
  if (P != 1 || g.gate || g.row_idx || g.gather_out || g.init_on) return false;
  if (g.K % 4 || g.N % 4 || g.lda % 4 || g.w_off % 4) return false;
  if ((reinterpret_cast<uintptr_t>(g.in) & 15) || (reinterpret_cast<uintptr_t>(g.theta) & 3)) return false;
  return g.N > 192 && g.N <= 200 && g.K >= 16 * 8;
}
static inline void pyz_launch_fwd_ring_ksplit(const DenseArgs &g, int grid_batch, hipStream_t st) {
  DenseArgs a = g;
  a.n_part = 1;
  a.grid_rows = grid_batch;
  a.n_cgrp = 1;
  a.cgrp_w = g.N;
  const long long wgs = (long long)((grid_batch + 31) / 32) * g.k_split;
  PYZ_LAUNCH((k_dense_fwd_ring<204, 13, 1, 8>), dim3((unsigned)((wgs + 7) / 8 * 8)), dim3(64 * 8), 0, st, a);
}

static inline bool pyz_launch_fwd_ring(const DenseArgs &g, int grid_batch, int P, hipStream_t st) {
  switch (pyz_fwd_ring_variant(g, grid_batch, P)) {
    case 1: {
      // Eight waves (four of them only move data).  Sub-slabs of 16 per barrier: two (121 KB of LDS, one workgroup per
      // CU) while the launch has no second workgroup per CU anyway -- 8 particles: 36.9 against 39.3 us -- else one (61 KB,
      // two per CU: 16 particles 62.0 against 70.5 us).  PYZ_FWD_RING_SUB / PYZ_FWD_RING_WAVES force a variant (measurements).
      static const int sub_env = pyz_env_int("PYZ_FWD_RING_SUB", 0), nw = pyz_env_int("PYZ_FWD_RING_WAVES", 8);
      const long long wgs = (long long)((grid_batch + 31) / 32) * P * (g.N > 200 ? (g.N + 199) / 200 : 1);
      const int sub = sub_env ? sub_env : (4 * wgs <= 5 * (long long)pyz_cu_count() ? 2 : 1);
      if (nw == 4 && sub == 2) pyz_launch_fwd_ring_as<204, 13, 2, 4>(g, grid_batch, P, st);
      else if (nw == 4) pyz_launch_fwd_ring_as<204, 13, 1, 4>(g, grid_batch, P, st);
      else if (sub == 2) pyz_launch_fwd_ring_as<204, 13, 2, 8>(g, grid_batch, P, st);
      else pyz_launch_fwd_ring_as<204, 13, 1, 8>(g, grid_batch, P, st);
      return true;
    }
    default: return false;
  }
}
