// bag_project.hip — the bag projection of up to TWO models (teacher and student) in ONE pass over the raw fp32 bag:
//
//     H_g[N,E] = dropout_g( act( X[N,D] . W_g[E,D]^T + b_g ) ),  g = 0 (teacher), 1 (student)      (mhim.py:69-76,84)
//
// The reference's student computes the feature on ALL N rows before it masks (modules/mhim.py:335-336) and its teacher
// does the same on the same bag (mhim.py:186): the two projections are one GEMM with B = [W_teacher; W_student]
// (2E = 1024 output columns).  X is read from HBM once per step instead of twice, there is no separate "pair planes"
// pass over the bag (the fp32 -> bf16 hi/lo split happens on the way into LDS, once per workgroup), and the per-CU
// operand ingest per output halves against the 160 x 128 tiling of feat_gemm.hip.
//
// Shape on MI355X: N = 10 000 rows x 1024 columns = 10.24 M outputs over 256 CUs = 40 000 per CU, so the workgroup
// tile is 160 x 256 (40 960 outputs): ceil(10000/160) x 4 = 252 tiles — one balanced round over the chip.
//   * 8 waves as 2 (M) x 4 (N), 80 x 64 outputs per wave = 5 x 4 blocks of v_mfma_f32_16x16x32_bf16 in the 3-term
//     bf16 form (hi*hi + hi*lo + lo*hi, ~2^-16: the instance scores feed a top-k), 60 MFMAs per 32-deep k-step per wave;
//     two waves per SIMD cover each other's LDS latency.
//   * B (the weights, already paired planes made by the step's prep launch) goes L2 -> LDS by direct DMA
//     (global_load_lds_dwordx4); A (raw fp32 X) goes through registers: 16-byte coalesced loads one k-step ahead,
//     split into bf16 hi / lo (v_cvt_pk_bf16_f32) and written as the same paired 128-byte row image the DMA path uses,
//     so the fragment reads (and their bank swizzle) are those of feat_gemm.hip.
//   * 3-stage LDS ring of [160 + 256 rows][128 B] = 156 KB, one s_barrier per k-step: B is issued two tiles ahead, A one tile
//     ahead; the MFMAs are software-pipelined across the barrier (the last term of tile t-1 runs under the first fragment reads
//     of tile t), so the matrix pipe has work while LDS reads are in flight.
//   * epilogue through LDS in two 80-row halves (one compact loop: bias, GELU and its derivative from one erf, counter
//     dropout with one hash per two elements, 512-byte row stores).  The student's d out / d pre goes out as fp16.
//   * XCD-aware tile order: the four column tiles of one row tile run on the same XCD (X rows shared through its L2).
#include <stdlib.h>
#include "mma_tile.hpp"

namespace mhimx {

constexpr int PBM = 160, PBN = 256, PBK = 32, PTHREADS = 512;
constexpr int PA_BYTES = PBM * 128, PB_BYTES = PBN * 128, PSTAGE = PA_BYTES + PB_BYTES;      // 20 KiB + 32 KiB
constexpr int PNST = 3;                                                                          // 156 KiB ring
constexpr int PTP = PBN + 4;                                                                     // epilogue tile pitch (floats)

typedef __bf16 pj_bf4 __attribute__((ext_vector_type(4)));
typedef __bf16 pj_bf2 __attribute__((ext_vector_type(2)));
typedef _Float16 pj_h4 __attribute__((ext_vector_type(4)));
typedef float pj_f2 __attribute__((ext_vector_type(2)));

// one hash per TWO elements: 16-bit fields against a 16-bit threshold (p quantised to 1/65536; the keep scale uses the
// quantised probability, so the mask is exactly unbiased)
MHIMX_DEV uint32_t pj_pair_hash(uint32_t row_key, uint32_t pair) { return mix32(row_key + pair * 0x85EBCA77u); }

__global__ __launch_bounds__(PTHREADS, 2) void bag_project_kernel(mhimx_bag_project_args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int nN = (int)(g.n_heads * g.E / PBN), nM = (int)((g.N + PBM - 1) / PBM);
  const int xcd = blockIdx.x & 7, sidx = blockIdx.x >> 3;
#ifdef PJ_XCD_SPLIT
  // an XCD works on TWO column tiles only (its slice of the weights, 2 MB, stays in its 4 MB L2 while X streams through)
  const int n_tile = (nN == 4) ? (xcd & 1) * 2 + (sidx & 1) : sidx % nN;
  const int m_tile = (nN == 4) ? (sidx >> 1) * 4 + (xcd >> 1) : (sidx / nN) * 8 + xcd;
#else
  const int m_tile = (sidx / nN) * 8 + xcd, n_tile = sidx % nN;
#endif
  if (m_tile >= nM) return;
  const int64_t m0 = (int64_t)m_tile * PBM;
  const int tiles_per_head = (int)(g.E / PBN);
  const int hd = n_tile / tiles_per_head;                     // which model this column tile belongs to
  const int64_t n0 = (int64_t)(n_tile % tiles_per_head) * PBN;   // first output column inside that model
  // (a dynamically indexed by-value struct would be copied to scratch: pick the fields with selects)
  mhimx_proj_head H = g.head[0];
  if (hd == 1) H = g.head[1];

  // ---- A (raw fp32 rows), every thread alike: two 16-byte units u = tid + 512 j (row u >> 3, slot u & 7; rows 0..127) and one
  // 8-byte unit of rows 128..159 (row 128 + (tid >> 4), half-slot tid & 15)
  unsigned aoff[3];                                           // byte offsets from g.X (a bag is < 4 GiB)
  unsigned a_hi[3], a_lo[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int row = j < 2 ? (tid >> 3) + 64 * j : 128 + (tid >> 4);
    const int slot = j < 2 ? (tid & 7) : ((tid & 15) >> 1);
    const int sub = j < 2 ? 0 : (tid & 1) * 4;
    int64_t m = m0 + row;
    if (m >= g.N) m = g.N - 1;                               // clamped rows feed accumulators that are never stored
    aoff[j] = (unsigned)((m * g.ldx + slot * 4 + (sub >> 1)) * 4);
    const int sw = mt_swz(row), kg2 = (slot >> 1) * 2;
    a_hi[j] = (unsigned)(row * 128 + ((kg2 ^ sw) << 4) + (slot & 1) * 8 + sub);
    a_lo[j] = (unsigned)(row * 128 + (((kg2 + 1) ^ sw) << 4) + (slot & 1) * 8 + sub);
  }
  // ---- B (paired weights) by DMA: slot p = tid + 512 j of a [256 rows][8 x 16 B] tile, SOURCE slot swizzled
  // (piece j covers rows 64 j + (tid >> 3): the swizzle only depends on row & 15, so the pieces share one per-lane offset)
  const unsigned boff = (unsigned)((((tid >> 3)) * g.D + ((tid & 7) ^ mt_swz(tid >> 3)) * 4) * 4);
  const char* bbase = reinterpret_cast<const char*>(H.wp + n0 * g.D);
  // `live` false (past the last tile): the same four DMA pieces are issued from ONE address (a single cache line, into a stage
  // nobody reads any more), so that every iteration has the same VMEM count and the hand-written vmcnt waits need no branch.
  auto issue_b = [&](int t, bool live) {
    char* sb = smem + (t % PNST) * PSTAGE + PA_BYTES + wave * 1024;
    const int64_t k0 = (int64_t)t * PBK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const char* bj = bbase + ((int64_t)j * 64 * g.D + k0) * 4;      // uniform
      __builtin_amdgcn_global_load_lds((gptr_f)(live ? bj + boff : reinterpret_cast<const char*>(g.X)), (lptr_f)(sb + j * 8192), 16, 0, 0);
    }
  };
  // The A loads of the loop are inline asm, waited for by hand: the compiler's own wait-count pass would put a vmcnt(0) in front of
  // their first use (it cannot count the DMA pieces issued behind them across the back-edge), draining everything in flight.
  // Two register sets alternate (the loop is unrolled by two), every load is unconditional: no phi, no register copy between a
  // load and its wait.
  struct ARegs { f32x4 v0, v1; pj_f2 v2; };
  auto load_a_async = [&](int t, ARegs& r) {
    const float* xk = g.X + (int64_t)t * PBK;                 // uniform: an SGPR pair
    asm volatile("global_load_dwordx4 %0, %3, %6\n\tglobal_load_dwordx4 %1, %4, %6\n\tglobal_load_dwordx2 %2, %5, %6"
                 : "=&v"(r.v0), "=&v"(r.v1), "=&v"(r.v2)
                 : "v"(aoff[0]), "v"(aoff[1]), "v"(aoff[2]), "s"(xk)
                 : "memory");
  };
  auto split4 = [&](const f32x4& v, char* hi_p, char* lo_p) {
    pj_bf4 hi, lo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const __bf16 h = (__bf16)v[q];
      hi[q] = h;
      lo[q] = (__bf16)(v[q] - (float)h);
    }
    *reinterpret_cast<pj_bf4*>(hi_p) = hi;
    *reinterpret_cast<pj_bf4*>(lo_p) = lo;
  };
  auto store_a = [&](int t, const ARegs& r) {                 // registers -> bf16 hi / lo -> the paired row image of stage t % 3
    char* sa = smem + (t % PNST) * PSTAGE;
    split4(r.v0, sa + a_hi[0], sa + a_lo[0]);
    split4(r.v1, sa + a_hi[1], sa + a_lo[1]);
    pj_bf2 hi, lo;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const __bf16 h = (__bf16)r.v2[q];
      hi[q] = h;
      lo[q] = (__bf16)(r.v2[q] - (float)h);
    }
    *reinterpret_cast<pj_bf2*>(sa + a_hi[2]) = hi;
    *reinterpret_cast<pj_bf2*>(sa + a_lo[2]) = lo;
  };

  // fragment addresses (stage 0): row r = lane & 15 of a 16-row block, k-group kg = lane >> 4 -> slots 2kg (hi), 2kg+1 (lo)
  const int r16 = lane & 15, kg = lane >> 4;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_f)smem;
  const int ra = wm * 80 + r16, rb = wn * 64 + r16;
  const unsigned fa_hi = lds0 + ra * 128 + (((2 * kg) ^ mt_swz(ra)) << 4);
  const unsigned fa_lo = lds0 + ra * 128 + (((2 * kg + 1) ^ mt_swz(ra)) << 4);
  const unsigned fb_hi = lds0 + PA_BYTES + rb * 128 + (((2 * kg) ^ mt_swz(rb)) << 4);
  const unsigned fb_lo = lds0 + PA_BYTES + rb * 128 + (((2 * kg + 1) ^ mt_swz(rb)) << 4);

  f32x4 acc[NRA][NRB];
#pragma unroll
  for (int i = 0; i < NRA; ++i)
#pragma unroll
    for (int j = 0; j < NRB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (int)(g.D / PBK);
  ARegs rga, rgb;
#ifndef PJ_PINGPONG
  // prologue: B(0), B(1) in flight, A(0) -> stage 0 (the compiler's wait in front of the conversion drains all three), A(1) in flight
  issue_b(0, true);
  issue_b(1, nk > 1);
  {
    ARegs r0;
    const char* xb = reinterpret_cast<const char*>(g.X);
    r0.v0 = *reinterpret_cast<const f32x4*>(xb + aoff[0]);
    r0.v1 = *reinterpret_cast<const f32x4*>(xb + aoff[1]);
    r0.v2 = *reinterpret_cast<const pj_f2*>(xb + aoff[2]);
    store_a(0, r0);
  }
  load_a_async(nk > 1 ? 1 : 0, rga);
#else
  // prologue: tiles 0 and 1 complete in LDS (the compiler's wait in front of the first conversion drains the DMA pieces too), A(2), A(3)
  // in flight in the two register sets
  issue_b(0, true);
  issue_b(1, nk > 1);
  {
    ARegs r0, r1;
    const char* xb = reinterpret_cast<const char*>(g.X);
    const char* xb1 = xb + (nk > 1 ? PBK * 4 : 0);
    r0.v0 = *reinterpret_cast<const f32x4*>(xb + aoff[0]);
    r0.v1 = *reinterpret_cast<const f32x4*>(xb + aoff[1]);
    r0.v2 = *reinterpret_cast<const pj_f2*>(xb + aoff[2]);
    r1.v0 = *reinterpret_cast<const f32x4*>(xb1 + aoff[0]);
    r1.v1 = *reinterpret_cast<const f32x4*>(xb1 + aoff[1]);
    r1.v2 = *reinterpret_cast<const pj_f2*>(xb1 + aoff[2]);
    store_a(0, r0);
    if (nk > 1) store_a(1, r1);
  }
  load_a_async(nk > 2 ? 2 : nk - 1, rga);
  load_a_async(nk > 3 ? 3 : nk - 1, rgb);
#endif

#ifdef PJ_PINGPONG
  // ---- (opt-in experiment, -DPJ_PINGPONG; the default is the lock-step loop below) the k loop as a PING-PONG of the two waves of every SIMD (waves w and w + 4 share a SIMD; group = wave >> 2 = the wave's M half).
  // In lock-step (all eight waves read LDS, then all eight issue MFMAs: the default form below) the matrix pipe idles while the
  // fragments are read and the LDS idles under the MFMAs - measured: MFMA-only loop 27 us, data movement only 35-38 us, together 57 us.
  // Here every k-step is two slots with a workgroup barrier after each; in a slot one group is in its COMPUTE phase (the 60 MFMAs of a
  // k-step, every fragment already in registers: nothing but matrix instructions) while the other is in its LOAD phase for the next tile
  // it will multiply (its 18 fragment reads, its share of the split + LDS stores of the following A tile, the global loads and weight
  // DMA two tiles ahead).  Slot 2t: group 0 loads tile t | group 1 computes tile t-1;  slot 2t+1: group 0 computes tile t | group 1
  // loads tile t.  Per SIMD the matrix pipe always has one wave issuing MFMAs and the LDS / VMEM traffic of the partner runs under it.
  //   load(s):     read the 18 fragments of tile s (stage s % 3); issue this wave's DMA pieces of B(s+2) (stage (s+2) % 3 = (s-1) % 3: both
  //                groups are past their reads of tile s-1); drain LDS; wait until only the 7 youngest requests are in flight: A(s+2) -
  //                requested two iterations ago - is in its registers and B(s+1) has landed.
  //   compute(s):  the 60 MFMAs of tile s with the split of A(s+2) to bf16 hi / lo and its LDS stores (stage (s+2) % 3) interleaved into
  //                their issue gaps (two VALU per MFMA fit under the 16 cycles the matrix pipe needs); then A(s+4) is requested into the
  //                registers just emptied.
  //   Tile s+2 is complete when both groups have run compute(s) (slots 2s+1 and 2s+2) and load(s+1) (B landed: slots 2s+2, 2s+3); its
  //   first reader is group 0's load(s+2) in slot 2s+4.  Lead of the global requests: A two iterations, B one and a half.
  //   MEASURED (round 3, same box, p = 0.25, us): lock-step loop 72.4-73.8 | ping-pong with split / stores / DMA in the LOAD phase 71.0-71.8
  //   (any order of its pieces) | this form (split + stores interleaved into the MFMAs) 75.4 | ping-pong without any global traffic 53.6,
  //   without MFMAs 57.1, without the epilogue 55.2 (the epilogue is 18 us: 51 MB of stores leave every CU at the same moment).
  //   s_memtime stamps (PJ_PP_PROF): a load phase that also splits / stores / issues DMA takes ~1.8x the 60-MFMA phase (fragment reads ~480
  //   cycles for 4 x 18 KB, split + 6 stores ~400, four DMA issues ~320, waits ~650) and sets the slot length; moved into the compute phase the
  //   VALU chain of the split stalls the in-order MFMA issue instead.  The k-step moves 144 KB of fragment reads + 52 KB of operand writes
  //   through the LDS against 2 x 1020 MFMA cycles: ~80 % LDS occupancy whatever the phase structure - the lever is fewer fragment bytes per
  //   MFMA (larger wave tiles: 4 waves x 80 x 128 with 512 registers), not the phase order.  Kept out of the default build.
  f32x4 x[NFR];
#if defined(PJ_PP_NOREAD)
  for (int q = 0; q < NFR; ++q) x[q] = f32x4{1.f, 2.f, 3.f, 4.f};
#endif
#ifdef PJ_PP_PROF
  uint32_t pf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t pt = 0;
#define PF_MARK(i) { const uint64_t now_ = __builtin_amdgcn_s_memtime(); pf[i] += (uint32_t)(now_ - pt); pt = now_; }
#define PF_START() { pt = __builtin_amdgcn_s_memtime(); }
#else
#define PF_MARK(i)
#define PF_START()
#endif
  auto load_phase = [&](int s, ARegs& r_next) {               // r_next: the registers A(s+2) arrives in (stored by compute(s))
    const unsigned so = (unsigned)((s % PNST) * PSTAGE);
    PF_START();
    MT_READ9(x, 5, 10, fa_lo + so, fb_hi + so);
    MT_READ9(x, 0, 14, fa_hi + so, fb_lo + so);
#ifndef PJ_PP_NODMA
    issue_b(s + 2, s + 2 < nk);
#endif
    MT_WAIT9(0, x, 5, 10);
    MT_WAIT9(0, x, 0, 14);
    PF_MARK(0);
#ifndef PJ_PP_NODMA
    asm volatile("s_waitcnt vmcnt(7)" : "+v"(r_next.v0), "+v"(r_next.v1), "+v"(r_next.v2) : : "memory");
#else
    asm volatile("s_waitcnt vmcnt(3)" : "+v"(r_next.v0), "+v"(r_next.v1), "+v"(r_next.v2) : : "memory");
#endif
    PF_MARK(2);
  };
  auto compute_phase = [&](int s, ARegs& r) {                 // r: A(s+2), arrived
    PF_START();
    __builtin_amdgcn_sched_barrier(0);
    mt_term(x, 5, 10, acc);                                   // lo*hi
    mt_term(x, 0, 14, acc);                                   // hi*lo
    mt_term(x, 0, 10, acc);                                   // hi*hi
#ifndef PJ_PP_NOSTORE
    store_a(s + 2, r);            // (unconditional - one basic block with the MFMAs: past the last tile it writes clamped rows into a stage nobody reads)
#ifndef PJ_PP_NOINTERLEAVE
#pragma unroll
    for (int q = 0; q < 30; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);      // two VALU
      if (q % 4 == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // a DS write
    }
#endif
#endif
    __builtin_amdgcn_sched_barrier(0);
    load_a_async(s + 4 < nk ? s + 4 : nk - 1, r);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // my A(s+2) stores are in LDS before the slot's barrier
    PF_MARK(4);
  };
  int pf_slot = 3;
  auto slot_end = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#ifdef PJ_PP_PROF
    PF_MARK(pf_slot);
    pf_slot = pf_slot == 3 ? 5 : 3;          // 3: barrier wait after a load phase, 5: after a compute phase (the loop alternates them)
#endif
  };
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // my A(0) stores to LDS are done
  slot_end();
  // (one copy of the loop for both groups: group 1 runs it one slot later - it waits out slot 0, group 0 waits out the last slot)
  const bool late = __builtin_amdgcn_readfirstlane(wm) != 0;
  if (late) slot_end();
  {
    int t = 0;
#pragma unroll 1
    for (; t + 1 < nk; t += 2) {
      load_phase(t, rga);          slot_end();
      compute_phase(t, rga);       slot_end();
      load_phase(t + 1, rgb);      slot_end();
      compute_phase(t + 1, rgb);   slot_end();
    }
    if (t < nk) {
      load_phase(t, rga);          slot_end();
      compute_phase(t, rga);       slot_end();
    }
  }
  if (!late) slot_end();
#else
  // Iteration t (3-stage ring, A and B both two tiles ahead):  [barrier: tile t complete in stage t%3]
  //   reads g1(t) = {A lo, B hi};  A(t+2) loads -> the free register set;  B(t+2) DMA -> stage (t+2)%3 (= (t-1)%3: every wave is
  //   past its reads);  20 MFMAs hi*lo of tile t-1 (operands still in registers: they cover the latency of g1);
  //   reads g2(t) = {A hi, B lo} (their registers are free now);  20 MFMAs lo*hi of tile t (cover g2);  20 MFMAs hi*hi of tile t;
  //   wait until only this iteration's 7 VMEM operations are in flight (A(t+1) is in its registers, B(t+1) has landed), split A(t+1)
  //   to bf16 hi / lo and store it into stage (t+1)%3 (last read in iteration t-2).
  f32x4 x[NFR];
#if defined(PJ_NOREAD)
  for (int q = 0; q < NFR; ++q) x[q] = f32x4{1.f, 2.f, 3.f, 4.f};
#endif
  auto body = [&](int t, ARegs& r_load, ARegs& r_use) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // my A(t) stores to LDS are done
    __builtin_amdgcn_s_barrier();
    const unsigned so = (unsigned)((t % PNST) * PSTAGE);
#ifndef PJ_NOREAD
    MT_READ9(x, 5, 10, fa_lo + so, fb_hi + so);
#endif
#ifndef PJ_NOLOAD
#ifndef PJ_NOA
    load_a_async(t + 2 < nk ? t + 2 : nk - 1, r_load);
#endif
#ifndef PJ_NODMA
    issue_b(t + 2, t + 2 < nk);
#endif
#endif
    __builtin_amdgcn_sched_barrier(0);
#ifndef PJ_NOMMA
    if (t > 0) mt_term(x, 0, 14, acc);                        // hi*lo of tile t-1
#endif
    __builtin_amdgcn_sched_barrier(0);
#ifndef PJ_NOREAD
    MT_WAIT9(0, x, 5, 10);
    MT_READ9(x, 0, 14, fa_hi + so, fb_lo + so);
#endif
    __builtin_amdgcn_sched_barrier(0);
#ifndef PJ_NOMMA
    mt_term(x, 5, 10, acc);                                   // lo*hi
#endif
    __builtin_amdgcn_sched_barrier(0);
#ifndef PJ_NOREAD
    MT_WAIT9(0, x, 0, 14);
#endif
#ifndef PJ_NOLOAD
    // A(t+1) has been in flight for more than an iteration; its split and LDS stores are independent of the hi*hi MFMAs below
    // and issue in their shadow (one MFMA occupies the matrix pipe for 16 cycles and an issue slot for 4)
#if defined(PJ_NOA)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#elif defined(PJ_NODMA)
    asm volatile("s_waitcnt vmcnt(3)" : "+v"(r_use.v0), "+v"(r_use.v1), "+v"(r_use.v2) : : "memory");
#elif !defined(PJ_NOWAIT)
    asm volatile("s_waitcnt vmcnt(7)" : "+v"(r_use.v0), "+v"(r_use.v1), "+v"(r_use.v2) : : "memory");
#endif
#endif
    __builtin_amdgcn_sched_barrier(0);
#ifndef PJ_NOMMA
    mt_term(x, 0, 10, acc);                                   // hi*hi
#endif
#if !defined(PJ_NOLOAD) && !defined(PJ_NOWAIT) && !defined(PJ_NOA)
    if (t + 1 < nk) store_a(t + 1, r_use);
#endif
#if !defined(PJ_NOMMA) && !defined(PJ_NOLOAD) && !defined(PJ_NOINTERLEAVE)
#pragma unroll
    for (int q = 0; q < 20; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      // three VALU
      if (q % 3 == 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // a DS write
    }
#endif
  };
  int t = 0;
#pragma unroll 1
  for (; t + 1 < nk; t += 2) {
    body(t, rgb, rga);
    body(t + 1, rga, rgb);
  }
  if (t < nk) body(t, rgb, rga);
#ifndef PJ_NOMMA
  mt_term(x, 0, 14, acc);                                     // hi*lo of the last tile
#endif
#endif
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(rga.v0), "+v"(rga.v1), "+v"(rga.v2), "+v"(rgb.v0), "+v"(rgb.v1), "+v"(rgb.v2) : : "memory");
#ifdef PJ_NOEPI
  {
    float sacc = 0.f;
    for (int i = 0; i < NRA; ++i)
      for (int j = 0; j < NRB; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sacc != 1.2345e-30f) return;
  }
#endif

  // ---- epilogue, two 80-row halves through LDS (the ring is free)
  float* tile = reinterpret_cast<float*>(smem);
  uint32_t* rkeys = reinterpret_cast<uint32_t*>(smem + 80 * PTP * 4);
  const int c4 = (tid & 63) * 4, r0 = tid >> 6;               // this thread's 4 columns are fixed
  const int64_t n = n0 + c4;
  float bias[4] = {0.f, 0.f, 0.f, 0.f};
  if (H.bias) { const f32x4 b = *reinterpret_cast<const f32x4*>(H.bias + n); bias[0] = b[0]; bias[1] = b[1]; bias[2] = b[2]; bias[3] = b[3]; }
  const bool hashed = H.drop_p > 0.f && !H.drop_mask;
  const uint64_t dseed = hashed ? eff_seed(H.drop_seed, g.drop_tick) : 0;
  const uint32_t thr16 = (uint32_t)(H.drop_p * 65536.f + 0.5f);
  const float inv_keep = H.drop_mask ? 1.f / (1.f - H.drop_p) : 65536.f / (float)(65536u - thr16);
  _Float16* dact = reinterpret_cast<_Float16*>(H.dact);
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    __syncthreads();                                          // fragment reads / the previous half's tile reads are over
    if (wm == half) {
      const int cl = lane & 15, rq = lane >> 4;
#pragma unroll
      for (int i = 0; i < NRA; ++i)
#pragma unroll
        for (int j = 0; j < NRB; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) tile[(i * 16 + rq * 4 + e) * PTP + wn * 64 + j * 16 + cl] = acc[i][j][e];
    }
    if (hashed && tid < 80) rkeys[tid] = drop_row_key(dseed, (uint64_t)(m0 + half * 80 + tid));
    __syncthreads();
#ifndef PJ_EPI_UNROLL
#define PJ_EPI_UNROLL 1
#endif
#pragma unroll PJ_EPI_UNROLL
    for (int r = r0; r < 80; r += 8) {
      const int64_t m = m0 + half * 80 + r;
      if (m >= g.N) break;
      const f32x4 a = *reinterpret_cast<const f32x4*>(tile + r * PTP + c4);
      float v[4] = {a[0] + bias[0], a[1] + bias[1], a[2] + bias[2], a[3] + bias[3]};
      float ks[4] = {1.f, 1.f, 1.f, 1.f};
      if (H.drop_mask) {
        const uchar4 mk = *reinterpret_cast<const uchar4*>(H.drop_mask + m * g.E + n);
        ks[0] = mk.x ? inv_keep : 0.f; ks[1] = mk.y ? inv_keep : 0.f; ks[2] = mk.z ? inv_keep : 0.f; ks[3] = mk.w ? inv_keep : 0.f;
      } else if (hashed) {
        const uint32_t rk = rkeys[r];
        const uint32_t h0 = pj_pair_hash(rk, (uint32_t)(n >> 1)), h1 = pj_pair_hash(rk, (uint32_t)(n >> 1) + 1u);
        ks[0] = (h0 & 0xffffu) >= thr16 ? inv_keep : 0.f;
        ks[1] = (h0 >> 16) >= thr16 ? inv_keep : 0.f;
        ks[2] = (h1 & 0xffffu) >= thr16 ? inv_keep : 0.f;
        ks[3] = (h1 >> 16) >= thr16 ? inv_keep : 0.f;
      }
      if (dact) {
        pj_h4 d;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float y, gq;
          act_fwd_grad(v[q], g.act, y, gq);                   // d out / d pre for the backward (shares the erf)
          v[q] = y * ks[q];
          d[q] = (_Float16)(gq * ks[q]);
        }
#if defined(PJ_EPI_NOSTORE)
        if (d[0] == (_Float16)1234.5f) *reinterpret_cast<pj_h4*>(dact + m * g.E + n) = d;
#elif defined(PJ_EPI_NT)
        __builtin_nontemporal_store(d, reinterpret_cast<pj_h4*>(dact + m * g.E + n));
#else
        *reinterpret_cast<pj_h4*>(dact + m * g.E + n) = d;
#endif
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = act_fwd(v[q], g.act) * ks[q];
      }
      if (H.resid) {
        const f32x4 rr = *reinterpret_cast<const f32x4*>(H.resid + m * H.ldr + n);
        v[0] += rr[0]; v[1] += rr[1]; v[2] += rr[2]; v[3] += rr[3];
      }
#if defined(PJ_EPI_NOSTORE)
      if (v[0] == 1234.5f) *reinterpret_cast<f32x4*>(H.H + m * H.ldh + n) = f32x4{v[0], v[1], v[2], v[3]};
#elif defined(PJ_EPI_NT)
      __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(H.H + m * H.ldh + n));
#else
      *reinterpret_cast<f32x4*>(H.H + m * H.ldh + n) = f32x4{v[0], v[1], v[2], v[3]};
#endif
    }
  }
#ifdef PJ_PP_PROF
  __syncthreads();
  if (m_tile == 1 && n_tile == 0 && lane == 0 && (wave == 0 || wave == 4))
    for (int i = 0; i < 8; ++i) H.H[(m0 + wave) * H.ldh + i] = (float)pf[i];
#endif
}

int bag_project_ws(hipStream_t st, const mhimx_bag_project_args* bags, int n_bags);       // bag_project_ws.hip

static int bag_project_check(const mhimx_bag_project_args& g) {
  MHIMX_CHECK_ARG(g.X && g.N >= 1 && g.D >= PBK && g.D % PBK == 0, "bag_project: X [N,D] with D a multiple of 32");
  MHIMX_CHECK_ARG(g.E >= PBN && g.E % PBN == 0, "bag_project: E must be a multiple of 256");
  MHIMX_CHECK_ARG(g.n_heads >= 1 && g.n_heads <= MHIMX_PROJ_MAX_HEADS, "bag_project: 1..%d models", MHIMX_PROJ_MAX_HEADS);
  MHIMX_CHECK_ARG(g.ldx % 4 == 0 && g.ldx >= g.D && aligned16(g.X), "bag_project: X rows must be 16-byte aligned");
  for (int h = 0; h < g.n_heads; ++h) {
    const mhimx_proj_head& H = g.head[h];
    const bool scored = h == 0 && g.score0 != nullptr;          // (the scored model 0 may leave its feature rows unwritten)
    MHIMX_CHECK_ARG(H.wp && (H.H || scored) && aligned16(H.wp) && aligned16(H.H) && (!H.H || (H.ldh % 4 == 0 && H.ldh >= g.E)),
                    "bag_project: model %d: null / unaligned weight image or output", h);
    MHIMX_CHECK_ARG(!H.resid || (aligned16(H.resid) && H.ldr % 4 == 0 && H.ldr >= g.E && H.resid != H.H), "bag_project: model %d: unaligned residual rows / the output itself", h);
    MHIMX_CHECK_ARG(!H.bias || aligned16(H.bias), "bag_project: model %d: unaligned bias", h);
    MHIMX_CHECK_ARG(!H.dact || (reinterpret_cast<uintptr_t>(H.dact) & 7) == 0, "bag_project: model %d: unaligned dact", h);
    MHIMX_CHECK_ARG(H.drop_p >= 0.f && H.drop_p < 1.f, "bag_project: model %d: dropout probability outside [0,1)", h);
    MHIMX_CHECK_ARG(!H.drop_mask || (reinterpret_cast<uintptr_t>(H.drop_mask) & 3) == 0, "bag_project: model %d: unaligned mask", h);
  }
  return 0;
}

int bag_project(hipStream_t st, const mhimx_bag_project_args* bags, int n_bags) {
  const mhimx_bag_project_args& g = bags[0];
  for (int b = 0; b < n_bags; ++b) {
    if (int rc = bag_project_check(bags[b])) return rc;
    if (b == 0) continue;
    const mhimx_bag_project_args& q = bags[b];
    bool same = q.N == g.N && q.D == g.D && q.E == g.E && q.ldx == g.ldx && q.act == g.act && q.n_heads == g.n_heads && q.drop_tick == g.drop_tick;
    for (int h = 0; same && h < g.n_heads; ++h)
      same = q.head[h].wp == g.head[h].wp && q.head[h].bias == g.head[h].bias && q.head[h].ldh == g.head[h].ldh && q.head[h].drop_p == g.head[h].drop_p &&
             !q.head[h].drop_mask && !g.head[h].drop_mask && !q.head[h].resid && !g.head[h].resid && (q.head[h].dact != nullptr) == (g.head[h].dact != nullptr);
    MHIMX_CHECK_ARG(same, "bag_project_multi: bag %d: the bags of one launch share shapes, weights, activation and dropout law (no masks, no residual rows)", b);
  }
  // the specialised-wave form (bag_project_ws.hip: 8 ping-pong consumer waves + 4 producer waves) is the default; MHIMX_PROJ_LOCKSTEP=1
  // selects this file's uniform 8-wave kernel (same tiles, same arithmetic, same bits)
  static const bool lockstep = getenv("MHIMX_PROJ_LOCKSTEP") != nullptr;
  if (!lockstep || g.score0) return bag_project_ws(st, bags, n_bags);
  MHIMX_CHECK_ARG(n_bags == 1, "bag_project_multi: only the default (specialised-wave) kernel takes several bags");
  MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)bag_project_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PNST * PSTAGE)));
  const int nN = (int)(g.n_heads * g.E / PBN), nM = (int)cdiv(g.N, PBM);
  dim3 grid((unsigned)(8 * nN * cdiv(nM, 8)));
  hipLaunchKernelGGL(bag_project_kernel, grid, dim3(PTHREADS), PNST * PSTAGE, st, g);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

}  // namespace mhimx

extern "C" int mhimx_bag_project(void* stream, const mhimx_bag_project_args* a) {
  if (!a) return mhimx::fail(-1, "bag_project: null argument block");
  return mhimx::bag_project((hipStream_t)stream, a, 1);
}
extern "C" int mhimx_bag_project_multi(void* stream, const mhimx_bag_project_args* bags, int32_t n_bags) {
  using namespace mhimx;
  MHIMX_CHECK_ARG(bags && n_bags >= 1 && n_bags <= 8, "bag_project_multi: 1..8 bags");
  return bag_project((hipStream_t)stream, bags, n_bags);
}
