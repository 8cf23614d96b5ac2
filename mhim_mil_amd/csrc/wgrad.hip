// wgrad.hip — the weight gradient of the bag projection,  dW1[E,D] = dPRE[L,E]^T . X[rows[0..L)][D]   (mhim.py:69-76 backward),
// as two launches that share one operand format:
//
//   rows_dpre_image : dPRE[p,:] = dH[rows[p],:] * dact16[rows[p],:]  (activation + dropout backward of the rows that took part),
//                     its column sums (the bias gradient) and — instead of an fp32 [L,E] matrix — the MATRIX-CORE IMAGE of dPRE^T:
//                     per 32-row k-step and 128-column block one contiguous 16 KiB tile [k-octet 4][hi|lo 2][column slot 128][8 bf16].
//   bag_wgrad       : the TN product on v_mfma_f32_16x16x32_bf16, 3-term bf16 (hi*hi + hi*lo + lo*hi), split over the L reduction.
//
// Why: the generic TN kernel (gemm_dma.hip) keeps both operands fp32 in LDS, reads every k-contiguous fragment as 8 strided
// ds_read_b32 and splits it to bf16 hi/lo in EVERY consuming wave (~400 VALU per 24 MFMAs: the loop is VALU-bound, 50 us).
// Here the fp32 -> bf16 hi/lo split happens once per element: dPRE's inside the elementwise kernel that makes it (and lays
// it out k-contiguous, so the TN product needs no transposing read at all), X's on the way into LDS —
//   * X (raw fp32, gathered by row id) goes through registers: a wave loads 8 rows x 512 B (lanes 0-31: rows 0-3 of its octet,
//     lanes 32-63: rows 4-7, 16 B per lane, full 512-B segments), v_permlane32_swap gives every lane the 8 k-consecutive values of
//     two columns, which are split and written as two ds_write_b128 (hi, lo) per column: the transposition costs 8 swaps.
//   * LDS images are [k-octet][hi|lo][column slot][16 B]: the 16 lanes of a fragment read and the 32 lanes of a half-wave write hit
//     consecutive 16-byte slots — no bank conflicts, no swizzle.  The column slot of column c is (c % 4) * (cols/4) + c / 4, i.e. an
//     MFMA tile covers every 4th column: the four tiles of a lane are four CONSECUTIVE output columns (one 16-byte store each).
//   * dPRE^T tiles go L2 -> LDS by direct DMA, 16 KiB linear per k-step.
//   * workgroup tile 128 (E) x 256 (D), 8 waves as 2 x 4, 64 x 64 per wave = 48 MFMAs against 16 ds_read_b128 per k-step;
//     one s_barrier per k-step; the schedule of bag_project.hip (MFMAs of tile t-1 under the reads of tile t).  Both operands are
//     requested THREE k-steps ahead (144 KB in flight per CU: at ~2 us of loaded-memory latency two tiles ahead starve the ~70 KB/us a
//     CU can ingest): the image in a 5-deep LDS ring, X in three rotating register sets in front of a 2-deep LDS ring.
//   * 16 reduction slabs x 16 output tiles = 256 workgroups (one per CU); the tiles of a slab run on ONE XCD (its X rows and dPRE
//     tiles are shared through that L2); the slab sum is a job of the step's deferred reduction launch, as before.
#include <stdlib.h>
#include <string.h>

#include "mma_tile.hpp"
#include "mca2_side.hpp"
#include "reduce_jobs.hpp"

namespace mhimx {

constexpr int WBI = 128, WBN = 256, WBK = 32, WTHREADS = 512;
constexpr int WA_BYTES = WBI * 128, WB_BYTES = WBN * 128;                                         // 16 KiB, 32 KiB per k-step
constexpr int WNA = 5, WNB = 2;                      // ring depths: dPRE^T image tiles (DMA, three tiles ahead), X tiles (through registers)
constexpr int WRING = WNA * WA_BYTES + WNB * WB_BYTES;                                            // 144 KiB
constexpr int W_MAX_CHUNK = 3072;                                                                 // rows of one slab (row table: 12 KiB)
constexpr int WNF = 16;                              // fragments of a k-step: x[0..3] A hi, x[4..7] A lo, x[8..11] B hi, x[12..15] B lo

// four 16-row blocks of the E-side operand (512 B apart) into x[oa..oa+3], four of the D-side operand (1 KiB apart) into x[ob..ob+3]
#define WG_READ8(x, oa, ob, a, b)                                                                                       \
  asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:512\n\tds_read_b128 %2, %8 offset:1024\n\t"           \
               "ds_read_b128 %3, %8 offset:1536\n\t"                                                                    \
               "ds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:1024\n\tds_read_b128 %6, %9 offset:2048\n\t"          \
               "ds_read_b128 %7, %9 offset:3072"                                                                        \
               : "=&v"(x[oa]), "=&v"(x[oa + 1]), "=&v"(x[oa + 2]), "=&v"(x[oa + 3]), "=&v"(x[ob]), "=&v"(x[ob + 1]),       \
                 "=&v"(x[ob + 2]), "=&v"(x[ob + 3])                                                                      \
               : "v"(a), "v"(b)                                                                                         \
               : "memory")
#define WG_WAIT8(n, x, oa, ob)                                                                                          \
  asm volatile("s_waitcnt lgkmcnt(" #n ")"                                                                              \
               : "+v"(x[oa]), "+v"(x[oa + 1]), "+v"(x[oa + 2]), "+v"(x[oa + 3]), "+v"(x[ob]), "+v"(x[ob + 1]),             \
                 "+v"(x[ob + 2]), "+v"(x[ob + 3])                                                                        \
               :                                                                                                        \
               : "memory")

MHIMX_DEV void wg_term(const f32x4 (&x)[WNF], int oa, int ob, f32x4 (&acc)[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = mt_mfma(x[oa + i], x[ob + j], acc[i][j]);
}

// a's lanes 32..63 <-> b's lanes 0..31.  Inline asm: this compiler's __builtin_amdgcn_permlane32_swap drops the second result (both
// elements of the returned pair read back as the first operand).  The s_nop covers the VALU-write -> permlane-swap hazard the
// assembler does not see inside an asm block.
MHIMX_DEV void wg_swap(float& a, float& b) { asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }

MHIMX_DEV void wg_split8(const float (&v)[8], f32x4& hi_o, f32x4& lo_o) {
  bf8 hi, lo;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const __bf16 h = (__bf16)v[q];
    hi[q] = h;
    lo[q] = (__bf16)(v[q] - (float)h);
  }
  hi_o = __builtin_bit_cast(f32x4, hi);
  lo_o = __builtin_bit_cast(f32x4, lo);
}

struct WgradArgs {
  const char* img;            // dPRE^T image (rows_dpre_image)
  const float* X;
  int64_t ldx;
  const int64_t* rows;        // [L] row ids into X, or null
  int64_t L, E, D;
  int ksteps, kps, splits;    // k-steps in all, per slab, slabs
  float* out;                 // slabs [splits][E][D]
};

// Trailing workgroups of a launch (mhimx_bag_wgrad_args.ride_tail): blocks [first, first + n_reduce) run the step's deferred reductions that
// were already queued (reduce_jobs.hpp), the stage3 blocks in front of them the last stage of the Merge backward's tail - behind a gate: the stage-2
// blocks at the FRONT of the same grid count their arrivals in side.w.gate (relaxed agent-scope atomics; the one matrix that crosses, dQ,
// travels as write-through stores and cache-bypassing loads) and a stage-3 block starts when all of them have arrived.  Blocks are dispatched in order, so the stage-2 blocks are resident or done long before a trailing block
// starts: the wait cannot deadlock, and with a grid of more than one round of tiles it is over before it begins.
struct WgradTail { int first, n_reduce, stage3, gate_want; ReduceTable t; };      // blocks: [first, first + stage3) stage 3, then n_reduce reductions
constexpr unsigned WG_GATE_SPINS = 1u << 20;

// the rider workgroups of a weight-gradient launch (see WgradTail): true if this block was one
MHIMX_DEV bool wg_rider_block(char* smem, int side_blocks, const Merge2Side& side, const WgradTail& tail) {
  if ((int)blockIdx.x < side_blocks) {
    if (threadIdx.x < M2_THREADS) {
      merge2_side_stage(2, (int)blockIdx.x, reinterpret_cast<float*>(smem), side);
      if (tail.stage3) {                         // (the waves above M2_THREADS have left: the barrier counts the four that remain)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's write-through stores of dQ are acknowledged ...
        __syncthreads();                         // ... before the barrier that precedes the arrival
        if (threadIdx.x == 0) __hip_atomic_fetch_add(side.w.gate, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return true;
  }
  if (tail.first && (int)blockIdx.x >= tail.first) {
    const int tb = (int)blockIdx.x - tail.first;
    if (tb >= tail.stage3) {
      reduce_jobs_block<WTHREADS>(tail.t, tb - tail.stage3, reinterpret_cast<float(*)[33]>(smem));
    } else if (threadIdx.x < M2_THREADS) {       // (the tail's stage-3 workgroups come first: the longest chain of the trailing blocks)
      if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(side.w.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)tail.gate_want && ++spins < WG_GATE_SPINS)
          __builtin_amdgcn_s_sleep(4);
      }
      __syncthreads();                           // (what stage 3 reads of stage 2 - dQ - it reads past the caches: mca2_side.hpp)
      merge2_side_stage(3, tb, reinterpret_cast<float*>(smem), side);
    }
    return true;
  }
  return false;
}

constexpr int W_MAX_BAGS = 8;
struct WgradBags {            // the bags of one launch (bag_wgrad_ws_kernel): bag b owns slabs [b * spb, (b + 1) * spb)
  const char* img[W_MAX_BAGS];
  const float* X[W_MAX_BAGS];
  const int64_t* rows[W_MAX_BAGS];
  int spb;
};

__global__ __launch_bounds__(WTHREADS) void bag_wgrad_kernel(WgradArgs g, int side_blocks, Merge2Side side) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.x < side_blocks) {          // a parked Merge-backward tail rides along (stage 2, 256 of the 512 threads): its few
    if (threadIdx.x < M2_THREADS) merge2_side_stage(2, (int)blockIdx.x, reinterpret_cast<float*>(smem), side);   // short workgroups go first
    return;
  }
  const unsigned bx = blockIdx.x - (unsigned)side_blocks;      // (side_blocks % 8 == 0: the XCD of a tile does not move)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int nJ = (int)(g.D / WBN), nIT = (int)(g.E / WBI), nT = nIT * nJ;
  const int xcd = bx & 7, sidx = bx >> 3;
  const int slab = (sidx / nT) * 8 + xcd, tile = sidx % nT;
  if (slab >= g.splits) return;
  const int itile = tile / nJ;
  const int64_t i0 = (int64_t)itile * WBI, n0 = (int64_t)(tile % nJ) * WBN;
  const int ks0 = slab * g.kps;
  const int nk = (ks0 + g.kps < g.ksteps ? ks0 + g.kps : g.ksteps) - ks0;       // >= 1 by the host's choice of splits

  // byte offsets of this slab's X rows (a bag is < 4 GiB); rows past L repeat the last one (their dPRE image rows are zero)
  unsigned* rowtab = reinterpret_cast<unsigned*>(smem + WRING);

  // ---- X: wave -> (row octet o of the k-step, column half ch); lane -> (rows 4 half .. 4 half + 3 of the octet, columns 4c .. 4c+3)
  const int oct = wave & 3, ch = wave >> 2, half = lane >> 5, c = lane & 31;
  const unsigned colb = (unsigned)((n0 + ch * 128 + 4 * c) * 4);
  const unsigned rt_lds = (unsigned)(uintptr_t)(lptr_f)rowtab + (unsigned)((oct * 8 + half * 4) * 4);    // + t * 128
  struct XRegs { f32x4 v[4]; };
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  auto load_x_async = [&](const u32x4& ro, XRegs& r) {
    asm volatile("global_load_dwordx4 %0, %4, %8\n\tglobal_load_dwordx4 %1, %5, %8\n\t"
                 "global_load_dwordx4 %2, %6, %8\n\tglobal_load_dwordx4 %3, %7, %8"
                 : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3])
                 : "v"(ro[0] + colb), "v"(ro[1] + colb), "v"(ro[2] + colb), "v"(ro[3] + colb), "s"(g.X)
                 : "memory");
  };
  // LDS slot of this lane's two columns after the half-wave swap: column 4c + j, j = s + 2 half (s = 0, 1) -> slot j * 64 + ch * 32 + c
  const unsigned xs0 = (unsigned)(WNA * WA_BYTES + ((oct * 2) * 256 + (2 * half) * 64 + ch * 32 + c) * 16);
  auto store_x = [&](int t, XRegs& r) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float a0 = r.v[q][0], a1 = r.v[q][1], a2 = r.v[q][2], a3 = r.v[q][3];
      wg_swap(a0, a2);
      wg_swap(a1, a3);
      r.v[q] = f32x4{a0, a1, a2, a3};
    }
    char* sb = smem + (t % WNB) * WB_BYTES + xs0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float kv[8] = {r.v[0][s], r.v[1][s], r.v[2][s], r.v[3][s], r.v[0][s + 2], r.v[1][s + 2], r.v[2][s + 2], r.v[3][s + 2]};
      f32x4 hi, lo;
      wg_split8(kv, hi, lo);
      *reinterpret_cast<f32x4*>(sb + s * 1024) = hi;
      *reinterpret_cast<f32x4*>(sb + s * 1024 + 4096) = lo;
    }
  };
  // ---- dPRE^T image tiles by DMA: 16 KiB linear, 2 x 16 B per thread.  `live` false: the pieces come from ONE address (a stage nobody
  // reads any more) so that every iteration has the same VMEM count and the hand-written vmcnt waits need no branch.
  const char* abase = g.img + ((int64_t)ks0 * nIT + itile) * WA_BYTES + tid * 16;
  auto issue_a = [&](int t, bool live) {
    char* sa = smem + (t % WNA) * WA_BYTES + wave * 1024;
    const char* src = live ? abase + (int64_t)t * nIT * WA_BYTES : g.img;
    __builtin_amdgcn_global_load_lds((gptr_f)src, (lptr_f)sa, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_f)(live ? src + 8192 : src), (lptr_f)(sa + 8192), 16, 0, 0);
  };

  issue_a(0, true);                                           // (before the row table: it does not depend on it)
  for (int q = tid; q < nk * WBK; q += WTHREADS) {
    int64_t l = (int64_t)ks0 * WBK + q;
    if (l >= g.L) l = g.L - 1;
    rowtab[q] = (unsigned)((g.rows ? g.rows[l] : l) * g.ldx * 4);
  }
  __syncthreads();
  // fragment addresses (stage 0): slot r = lane & 15 of a 16-slot block, k-octet kg = lane >> 4
  const int r16 = lane & 15, kg = lane >> 4;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_f)smem;
  const unsigned fa_hi = lds0 + ((kg * 2) * 128 + 16 * wm + r16) * 16, fa_lo = fa_hi + 2048;
  const unsigned fb_hi = lds0 + WNA * WA_BYTES + ((kg * 2) * 256 + 16 * wn + r16) * 16, fb_lo = fb_hi + 4096;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: image tile 0, X(0) -> stage 0 (plain loads: the compiler's wait drains both); then, in the loop's own order,
  // {X(1), image 1}, {X(2), image 2} in flight; row offsets of tile 3 in registers
  XRegs rg0, rg1, rg2;
  auto row_offsets = [&](int t) { return *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(rowtab) + (oct * 8 + half * 4) * 4 + (t < nk ? t : nk - 1) * 128); };
  {
    const u32x4 r0o = row_offsets(0);
    XRegs r0;
    const char* xb = reinterpret_cast<const char*>(g.X);
#pragma unroll
    for (int q = 0; q < 4; ++q) r0.v[q] = *reinterpret_cast<const f32x4*>(xb + r0o[q] + colb);
    store_x(0, r0);
  }
  load_x_async(row_offsets(1), rg1);
  issue_a(1, nk > 1);
  load_x_async(row_offsets(2), rg2);
  issue_a(2, nk > 2);
  u32x4 ro = row_offsets(3);

  f32x4 x[WNF];
#ifdef WG_NOREAD
  for (int q = 0; q < WNF; ++q) x[q] = f32x4{1.f, 2.f, 3.f, 4.f};
#endif
  // Iteration t:  [barrier: tile t complete in its stages]  X(t+3) loads -> the free register set, image DMA(t+3) -> stage (t+3)%5
  // (tile t-2's: every wave is past its reads); reads g1 = {A lo, B hi}; row offsets of tile t+4; 16 MFMAs hi*lo of tile t-1 (operands
  // still in registers); reads g2 = {A hi, B lo}; 16 MFMAs lo*hi; wait until only the last two iterations' 12 VMEM operations are in
  // flight (X(t+1) is in its registers, image t+1 has landed); 16 MFMAs hi*hi with the swap / split / LDS stores of X(t+1) into stage
  // (t+1)%2 (tile t-1's) in their shadow.
  auto body = [&](int t, XRegs& r_load, XRegs& r_use) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ro) : : "memory");       // my X(t) stores are done; the row offsets are here
    __builtin_amdgcn_s_barrier();
    const unsigned soa = (unsigned)((t % WNA) * WA_BYTES), sob = (unsigned)((t % WNB) * WB_BYTES);
#ifndef WG_NOX
    load_x_async(ro, r_load);
#endif
#ifndef WG_NODMA
    issue_a(t + 3, t + 3 < nk);
#endif
#ifndef WG_NOREAD
    WG_READ8(x, 4, 8, fa_lo + soa, fb_hi + sob);
#endif
    {
      const unsigned ra = rt_lds + (unsigned)((t + 4 < nk ? t + 4 : nk - 1) * 128);
      asm volatile("ds_read_b128 %0, %1" : "=&v"(ro) : "v"(ra) : "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
#ifndef WG_NOMMA
    if (t > 0) wg_term(x, 0, 12, acc);                        // hi*lo of tile t-1
#endif
    __builtin_amdgcn_sched_barrier(0);
#ifndef WG_NOREAD
    WG_WAIT8(0, x, 4, 8);
    WG_READ8(x, 0, 12, fa_hi + soa, fb_lo + sob);
#endif
    __builtin_amdgcn_sched_barrier(0);
#ifndef WG_NOMMA
    wg_term(x, 4, 8, acc);                                    // lo*hi
#endif
    __builtin_amdgcn_sched_barrier(0);
#ifndef WG_NOREAD
    WG_WAIT8(0, x, 0, 12);
#endif
#if defined(WG_NOX) && defined(WG_NODMA)
#elif defined(WG_NOX)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#elif defined(WG_NODMA)
    asm volatile("s_waitcnt vmcnt(8)" : "+v"(r_use.v[0]), "+v"(r_use.v[1]), "+v"(r_use.v[2]), "+v"(r_use.v[3]) : : "memory");
#else
    asm volatile("s_waitcnt vmcnt(12)" : "+v"(r_use.v[0]), "+v"(r_use.v[1]), "+v"(r_use.v[2]), "+v"(r_use.v[3]) : : "memory");
#endif
    __builtin_amdgcn_sched_barrier(0);
#ifndef WG_NOMMA
    wg_term(x, 0, 8, acc);                                    // hi*hi
#endif
#if !defined(WG_NOX) && !defined(WG_NOSTORE)
    if (t + 1 < nk) store_x(t + 1, r_use);
#endif
#if !defined(WG_NOINTERLEAVE) && !defined(WG_NOMMA) && !defined(WG_NOX) && !defined(WG_NOSTORE)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // four VALU
      if (q % 4 == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // a DS write
    }
#endif
  };
  int t = 0;
#pragma unroll 1
  for (; t + 2 < nk; t += 3) {
    body(t, rg0, rg1);
    body(t + 1, rg1, rg2);
    body(t + 2, rg2, rg0);
  }
  if (t < nk) body(t, rg0, rg1);
  if (t + 1 < nk) body(t + 1, rg1, rg2);
#ifndef WG_NOMMA
  wg_term(x, 0, 12, acc);                                     // hi*lo of the last tile
#endif
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
               : "+v"(rg0.v[0]), "+v"(rg0.v[1]), "+v"(rg0.v[2]), "+v"(rg0.v[3]), "+v"(rg1.v[0]), "+v"(rg1.v[1]), "+v"(rg1.v[2]), "+v"(rg1.v[3]),
                 "+v"(rg2.v[0]), "+v"(rg2.v[1]), "+v"(rg2.v[2]), "+v"(rg2.v[3]), "+v"(ro)
               :
               : "memory");

#ifdef WG_NOEPI
  {
    float sacc = 0.f;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sacc != 1.2345e-30f) return;
  }
#endif
  // ---- epilogue: tile (ja, jb) of a lane is row 4 (16 wm + 4 (lane >> 4) + e) + ja, column 4 (16 wn + (lane & 15)) + jb: 16-byte stores
  float* out = g.out + (int64_t)slab * g.E * g.D;
#pragma unroll
  for (int ja = 0; ja < 4; ++ja)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t i = i0 + 4 * (16 * wm + 4 * kg + e) + ja;
      const int64_t n = n0 + 4 * (16 * wn + r16);
      *reinterpret_cast<f32x4*>(out + i * g.D + n) = f32x4{acc[ja][0][e], acc[ja][1][e], acc[ja][2][e], acc[ja][3][e]};
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same product with SPECIALISED waves.  In bag_wgrad_kernel every wave loads, splits, stores, reads fragments and multiplies; the
// counters say its waves wait 46 % of their time with the matrix pipe 44 % busy (two waves per SIMD in lock-step phases: when one
// waits on memory so does the other).  Here waves 0-3 (one per SIMD) are PRODUCERS - X loads, half-wave swap, bf16 split, LDS stores,
// the image DMA - and waves 4-7 CONSUMERS that touch no global memory in the loop: 64 x 128 outputs each (32 accumulator tiles), 24
// fragment reads and 96 MFMAs per k-step, fragments of tile t+1 requested while tile t multiplies.  A producer's instructions issue in
// the shadow of its SIMD's consumer MFMAs; fragment reads per MFMA drop by a quarter (96 KB instead of 128 KB per k-step per CU).
// One s_barrier per k-step: it publishes tile t+2 and retires tile t; 3-deep rings for both operands (144 KB).
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int SNA = 3, SNB = 3, SRING = SNA * WA_BYTES + SNB * WB_BYTES;

// four fragment blocks of the E-side operand (512 B apart) / of one column half of the D-side operand (1 KiB apart)
#define WS_READ4A(d, p)                                                                                                  \
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:512\n\tds_read_b128 %2, %4 offset:1024\n\tds_read_b128 %3, %4 offset:1536" \
               : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]) : "v"(p) : "memory")
#define WS_READ4B(d, p)                                                                                                  \
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072" \
               : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]) : "v"(p) : "memory")
// wait until at most n LDS operations are outstanding; names the two groups the following MFMAs read
#define WS_WAIT8(n, a, b)                                                                                                \
  asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : : "memory")

// 16 MFMAs: four E-side blocks x four D-side blocks of column half `cb`
template <int CB>
MHIMX_DEV void ws_unit(const f32x4 (&a)[4], const f32x4 (&b)[4], f32x4 (&acc)[4][8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][4 * CB + j] = mt_mfma(a[i], b[j], acc[i][4 * CB + j]);
}

__global__ __launch_bounds__(WTHREADS) void bag_wgrad_ws_kernel(WgradArgs g, WgradBags mb, int side_blocks, Merge2Side side, WgradTail tail) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (wg_rider_block(smem, side_blocks, side, tail)) return;
  const unsigned bx = blockIdx.x - (unsigned)side_blocks;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef WG_PROF                                                  // shader cycles: entry -> loop, loop, epilogue (tools/exp_wgrad.py WG_PROF=1)
  const uint64_t wp_t0 = __builtin_readcyclecounter();
  const uint64_t wp_rt0 = wall_clock64();                        // (round 6: the 100 MHz clock beside it -> the shader clock the launch holds)
#endif
  const int nJ = (int)(g.D / WBN), nIT = (int)(g.E / WBI), nT = nIT * nJ;
  const int xcd = bx & 7, sidx = bx >> 3;
  const int slab = (sidx / nT) * 8 + xcd, tile = sidx % nT;
  if (slab >= g.splits) return;
  const int itile = tile / nJ;
  const int64_t i0 = (int64_t)itile * WBI, n0 = (int64_t)(tile % nJ) * WBN;
  // an accumulation window's bags in ONE launch (mhimx_bag_wgrad_multi): slab -> (bag, slab of that bag); every bag has the same L
  const int bag = slab / mb.spb, ks0 = (slab - bag * mb.spb) * g.kps;
  const char* const gimg = mb.img[bag];
  const float* const gX = mb.X[bag];
  const int64_t* const grows = mb.rows[bag];
  const int nk = (ks0 + g.kps < g.ksteps ? ks0 + g.kps : g.ksteps) - ks0;
  unsigned* rowtab = reinterpret_cast<unsigned*>(smem + SRING);
  for (int q = tid; q < nk * WBK; q += WTHREADS) {
    int64_t l = (int64_t)ks0 * WBK + q;
    if (l >= g.L) l = g.L - 1;
    rowtab[q] = (unsigned)((grows ? grows[l] : l) * g.ldx * 4);
  }
  __syncthreads();
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_f)smem;

#if defined(WG_PROD_PRIO) || defined(WG_CONS_PRIO)
#ifndef WG_PROD_PRIO
#define WG_PROD_PRIO 0
#endif
#ifndef WG_CONS_PRIO
#define WG_CONS_PRIO 0
#endif
  if (wave < 4) __builtin_amdgcn_s_setprio(WG_PROD_PRIO);
  else __builtin_amdgcn_s_setprio(WG_CONS_PRIO);
#endif
  if (wave < 4) {
    // =========================================================== producers: wave = row octet of the k-step
    const int oct = wave, half = lane >> 5, c = lane & 31;
    const unsigned colb0 = (unsigned)((n0 + 4 * c) * 4), colb1 = colb0 + 128 * 4;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    struct XSet { f32x4 v[8]; };                               // [column half][row of the lane's four]
    auto row_offsets = [&](int t) {
      return *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(rowtab) + (oct * 8 + half * 4) * 4 + (t < nk ? t : nk - 1) * 128);
    };
    auto load_x_async = [&](const u32x4& ro, XSet& r) {
      asm volatile("global_load_dwordx4 %0, %8, %16\n\tglobal_load_dwordx4 %1, %9, %16\n\tglobal_load_dwordx4 %2, %10, %16\n\t"
                   "global_load_dwordx4 %3, %11, %16\n\tglobal_load_dwordx4 %4, %12, %16\n\tglobal_load_dwordx4 %5, %13, %16\n\t"
                   "global_load_dwordx4 %6, %14, %16\n\tglobal_load_dwordx4 %7, %15, %16"
                   : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7])
                   : "v"(ro[0] + colb0), "v"(ro[1] + colb0), "v"(ro[2] + colb0), "v"(ro[3] + colb0), "v"(ro[0] + colb1), "v"(ro[1] + colb1),
                     "v"(ro[2] + colb1), "v"(ro[3] + colb1), "s"(gX)
                   : "memory");
    };
    const unsigned xs0 = (unsigned)(SNA * WA_BYTES + ((oct * 2) * 256 + (2 * half) * 64 + c) * 16);
    auto store_x = [&](int t, XSet& r) {
      char* sb = smem + (t % SNB) * WB_BYTES + xs0;
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        float a[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          a[q][0] = r.v[ch * 4 + q][0]; a[q][1] = r.v[ch * 4 + q][1]; a[q][2] = r.v[ch * 4 + q][2]; a[q][3] = r.v[ch * 4 + q][3];
          wg_swap(a[q][0], a[q][2]);
          wg_swap(a[q][1], a[q][3]);
        }
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
          const float kv[8] = {a[0][sx], a[1][sx], a[2][sx], a[3][sx], a[0][sx + 2], a[1][sx + 2], a[2][sx + 2], a[3][sx + 2]};
          f32x4 hi, lo;
          wg_split8(kv, hi, lo);
          *reinterpret_cast<f32x4*>(sb + ch * 512 + sx * 1024) = hi;
          *reinterpret_cast<f32x4*>(sb + ch * 512 + sx * 1024 + 4096) = lo;
        }
      }
    };
    const char* abase = gimg + ((int64_t)ks0 * nIT + itile) * WA_BYTES + (wave * 64 + lane) * 16;
    auto issue_a = [&](int t, bool live) {                     // 16 KiB by 256 threads: four 4 KiB pieces
      char* sa = smem + (t % SNA) * WA_BYTES + wave * 1024;
      const char* src = live ? abase + (int64_t)t * nIT * WA_BYTES : gimg;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_global_load_lds((gptr_f)(live ? src + j * 4096 : src), (lptr_f)(sa + j * 4096), 16, 0, 0);
    };
    XSet s0, s1, s2;
    issue_a(0, true);
    issue_a(1, nk > 1);
    {                                                         // (requesting X(0), X(1), X(2) together shortens entry -> loop from 9.3 k to 7.7 k
      const char* xb = reinterpret_cast<const char*>(gX);    //  cycles - stamped, -DWG_PROF - and the launch not at all: 32.1 vs 32.0 us)
      for (int t = 0; t < 2 && t < nk; ++t) {
        const u32x4 ro = row_offsets(t);
        XSet r0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          r0.v[q] = *reinterpret_cast<const f32x4*>(xb + ro[q] + colb0);
          r0.v[4 + q] = *reinterpret_cast<const f32x4*>(xb + ro[q] + colb1);
        }
        store_x(t, r0);
      }
    }
    // in the loop's own order from here on: {4 image pieces, 8 X loads} per tile
    load_x_async(row_offsets(2), s2);
    issue_a(2, false);                                        // (four dummy pieces: keeps the loop's VMEM count)
    load_x_async(row_offsets(3), s0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                             // #0: tiles 0 and 1 are in LDS
    auto body = [&](int t, XSet& r_load, XSet& r_use) {
      issue_a(t + 2, t + 2 < nk);
      load_x_async(row_offsets(t + 4), r_load);
      asm volatile("s_waitcnt vmcnt(24)" : "+v"(r_use.v[0]), "+v"(r_use.v[1]), "+v"(r_use.v[2]), "+v"(r_use.v[3]), "+v"(r_use.v[4]),
                   "+v"(r_use.v[5]), "+v"(r_use.v[6]), "+v"(r_use.v[7]) : : "memory");
      if (t + 2 < nk) store_x(t + 2, r_use);
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");      // image tile t+2 landed, my X(t+2) stores are done
      __builtin_amdgcn_s_barrier();                           // #t+1
    };
#ifdef WG_PROF
    const uint64_t wp_t1 = __builtin_readcyclecounter();
#endif
    int t = 0;
#pragma unroll 1
    for (; t + 2 < nk; t += 3) {
      body(t, s1, s2);                                        // X(t+4) -> s1, X(t+2) from s2
      body(t + 1, s2, s0);
      body(t + 2, s0, s1);
    }
    if (t < nk) body(t, s1, s2);
    if (t + 1 < nk) body(t + 1, s2, s0);
    // EVERY register of the three sets is named: the loads of the last (clamped, unused) prefetches are still in flight here, and a register the
    // compiler believes dead is handed out again while its load has not landed - the in-flight data then overwrote the row offsets of the
    // next asm loads (GPU memory access fault when a second process stretched the latency; tools/asm_lint.py)
#define WS_ALL(s) "+v"(s.v[0]), "+v"(s.v[1]), "+v"(s.v[2]), "+v"(s.v[3]), "+v"(s.v[4]), "+v"(s.v[5]), "+v"(s.v[6]), "+v"(s.v[7])
    asm volatile("s_waitcnt vmcnt(0)" : WS_ALL(s0), WS_ALL(s1), WS_ALL(s2) : : "memory");
#undef WS_ALL
#ifdef WG_PROF
    if (bx == 0 && lane == 0) {
      const uint64_t wp_t2 = __builtin_readcyclecounter();
      float* pr = reinterpret_cast<float*>(const_cast<char*>(gimg) + (int64_t)g.ksteps * nIT * WA_BYTES) + wave * 4;   // behind the image (the script allocates it)
      pr[0] = (float)(wp_t1 - wp_t0); pr[1] = (float)(wp_t2 - wp_t1); pr[2] = 0.f; pr[3] = (float)nk;
    }
#endif
    return;
  }

  // =============================================================== consumers: 2 x 2 waves of 64 (E) x 128 (D)
  const int cw = wave - 4, wm = cw >> 1, wn = cw & 1;
  const int r16 = lane & 15, kg = lane >> 4;
  const unsigned fa_hi = lds0 + ((kg * 2) * 128 + 16 * wm + r16) * 16, fa_lo = fa_hi + 2048;
  const unsigned fb_hi = lds0 + SNA * WA_BYTES + ((kg * 2) * 256 + 32 * wn + r16) * 16, fb_lo = fb_hi + 4096;
  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Fragments: A lo, A hi and two D-side slots P, Q (four blocks each).  Per k-step six units of 16 MFMAs, ordered so that a slot is
  // refilled as soon as its last reader has issued and is needed again >= 256 cycles later:
  //   u1 Alo x P(Bhi0)   u2 Alo x Q(Bhi1)   [Alo <- tile t+1]   u3 Ahi x P(Bhi0)   [P <- Blo0]   u4 Ahi x Q(Bhi1)   [Q <- Blo1]
  //   u5 Ahi x P(Blo0)   [P <- Bhi0 of t+1]   u6 Ahi x Q(Blo1)   [Q <- Bhi1 of t+1, Ahi <- t+1]
  f32x4 alo[4], ahi[4], P[4], Q[4];
  __builtin_amdgcn_s_barrier();                               // #0
#ifdef WG_PROF
  const uint64_t wp_t1 = __builtin_readcyclecounter();
#endif
  WS_READ4A(alo, fa_lo);
  WS_READ4B(P, fb_hi);
  WS_READ4B(Q, fb_hi + 256);
  WS_READ4A(ahi, fa_hi);
#pragma unroll 1
  for (int t = 0; t < nk; ++t) {
    const unsigned ca = (unsigned)((t % SNA) * WA_BYTES), cbo = (unsigned)((t % SNB) * WB_BYTES);
    const unsigned na = (unsigned)(((t + 1) % SNA) * WA_BYTES), nb = (unsigned)(((t + 1) % SNB) * WB_BYTES);
    const bool more = t + 1 < nk;                             // (past the end the prefetches re-read the last tile: harmless, keeps the counts)
    const unsigned pa = more ? na : ca, pb = more ? nb : cbo;
    WS_WAIT8(8, alo, P);
    __builtin_amdgcn_sched_barrier(0);
    ws_unit<0>(alo, P, acc);                                  // u1  lo*hi, column half 0
    __builtin_amdgcn_sched_barrier(0);
    WS_WAIT8(4, alo, Q);
    __builtin_amdgcn_sched_barrier(0);
    ws_unit<1>(alo, Q, acc);                                  // u2  lo*hi, half 1
    __builtin_amdgcn_sched_barrier(0);
    WS_READ4A(alo, fa_lo + pa);                               // tile t+1 was published by barrier #t
    WS_WAIT8(4, ahi, P);
    __builtin_amdgcn_sched_barrier(0);
    ws_unit<0>(ahi, P, acc);                                  // u3  hi*hi, half 0
    __builtin_amdgcn_sched_barrier(0);
    WS_READ4B(P, fb_lo + cbo);
    ws_unit<1>(ahi, Q, acc);                                  // u4  hi*hi, half 1
    __builtin_amdgcn_sched_barrier(0);
    WS_READ4B(Q, fb_lo + cbo + 256);
    WS_WAIT8(4, ahi, P);
    __builtin_amdgcn_sched_barrier(0);
    ws_unit<0>(ahi, P, acc);                                  // u5  hi*lo, half 0
    __builtin_amdgcn_sched_barrier(0);
    WS_READ4B(P, fb_hi + pb);
    WS_WAIT8(4, ahi, Q);
    __builtin_amdgcn_sched_barrier(0);
    ws_unit<1>(ahi, Q, acc);                                  // u6  hi*lo, half 1
    __builtin_amdgcn_sched_barrier(0);
    WS_READ4B(Q, fb_hi + pb + 256);
    WS_READ4A(ahi, fa_hi + pa);
    __builtin_amdgcn_s_barrier();                             // #t+1: tile t is retired, tile t+2 published
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef WG_PROF
  const uint64_t wp_t2 = __builtin_readcyclecounter();
#endif
  // epilogue: block (ja, jq = 4 cb + jb) of a lane is row 4 (16 wm + 4 kg + e) + ja, column 4 (16 (2 wn + cb) + r16) + jb
  float* out = g.out + (int64_t)slab * g.E * g.D;
#pragma unroll
  for (int ja = 0; ja < 4; ++ja)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t i = i0 + 4 * (16 * wm + 4 * kg + e) + ja;
        const int64_t n = n0 + 4 * (16 * (2 * wn + cb) + r16);
#ifdef MHIMX_SLAB_WT
        st_f4_wt(out + i * g.D + n, f32x4{acc[ja][4 * cb + 0][e], acc[ja][4 * cb + 1][e], acc[ja][4 * cb + 2][e], acc[ja][4 * cb + 3][e]});
#else
        *reinterpret_cast<f32x4*>(out + i * g.D + n) =
            f32x4{acc[ja][4 * cb + 0][e], acc[ja][4 * cb + 1][e], acc[ja][4 * cb + 2][e], acc[ja][4 * cb + 3][e]};
#endif
      }
#ifdef WG_PROF
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (bx == 0 && lane == 0) {
    const uint64_t wp_t3 = __builtin_readcyclecounter();
    float* pr = reinterpret_cast<float*>(const_cast<char*>(gimg) + (int64_t)g.ksteps * nIT * WA_BYTES) + wave * 4;
    pr[0] = (float)(wp_t1 - wp_t0); pr[1] = (float)(wp_t2 - wp_t1); pr[2] = (float)(wp_t3 - wp_t2); pr[3] = (float)nk;
    if (wave == 4) { pr[32 - 16] = (float)(wall_clock64() - wp_rt0); pr[33 - 16] = (float)(wp_t3 - wp_t0); }   // (wave 4 = the first consumer wave: pr = base + 16) 10 ns ticks | shader cycles, entry -> end
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------
// Specialised waves AND a ping-pong of the consumers (the form of bag_project_ws.hip): 4 producer waves (one per SIMD: the X loads,
// half-wave swap, bf16 split, LDS stores and the image DMA, half of a k-step's work in each of its two slots) and 8 consumer waves as
// 2 x 4 of 64 x 64 outputs that only read fragments and issue MFMAs.  Consumers cw and cw + 4 share a SIMD and run half a k-step apart:
// slot 2s: group 0 reads the 16 fragments of tile s | group 1 issues the 48 MFMAs of tile s-1;  slot 2s+1: the other way round - the
// matrix pipe of every SIMD always has a wave feeding it and the fragment reads run under the partner's MFMAs (bag_wgrad_ws_kernel's four
// consumers interleave their own reads with their own MFMAs: every fragment wait stalls that SIMD's matrix pipe).  A workgroup barrier
// after each slot; 3-deep rings: in k-step s the producers store X(s+1) (stage of tile s-2), issue the DMA of image s+2 (stage of tile
// s-1: both groups are past it) and request X(s+4); at its end they wait until X(s+2) is in registers and image s+1 has landed.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int PP_THREADS = 768;
__global__ __launch_bounds__(PP_THREADS) void bag_wgrad_pp_kernel(WgradArgs g, int side_blocks, Merge2Side side) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.x < side_blocks) {
    if (threadIdx.x < M2_THREADS) merge2_side_stage(2, (int)blockIdx.x, reinterpret_cast<float*>(smem), side);
    return;
  }
  const unsigned bx = blockIdx.x - (unsigned)side_blocks;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nJ = (int)(g.D / WBN), nIT = (int)(g.E / WBI), nT = nIT * nJ;
  const int xcd = bx & 7, sidx = bx >> 3;
  const int slab = (sidx / nT) * 8 + xcd, tile = sidx % nT;
  if (slab >= g.splits) return;
  const int itile = tile / nJ;
  const int64_t i0 = (int64_t)itile * WBI, n0 = (int64_t)(tile % nJ) * WBN;
  const int ks0 = slab * g.kps;
  const int nk = (ks0 + g.kps < g.ksteps ? ks0 + g.kps : g.ksteps) - ks0;
  unsigned* rowtab = reinterpret_cast<unsigned*>(smem + SRING);
  for (int q = tid; q < nk * WBK; q += PP_THREADS) {
    int64_t l = (int64_t)ks0 * WBK + q;
    if (l >= g.L) l = g.L - 1;
    rowtab[q] = (unsigned)((g.rows ? g.rows[l] : l) * g.ldx * 4);
  }
  __syncthreads();
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_f)smem;
  auto slot_end = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  if (wave < 4) {
    // =========================================================== producers: wave = row octet of the k-step
    const int oct = wave, half = lane >> 5, c = lane & 31;
    const unsigned colb0 = (unsigned)((n0 + 4 * c) * 4), colb1 = colb0 + 128 * 4;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    struct XSet { f32x4 v[8]; };                               // [column half][row of the lane's four]
    auto row_offsets = [&](int t) {
      return *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(rowtab) + (oct * 8 + half * 4) * 4 + (t < nk ? t : nk - 1) * 128);
    };
    auto load_half = [&](const u32x4& ro, unsigned colb, f32x4* d) {          // four rows of one column half
      asm volatile("global_load_dwordx4 %0, %4, %8\n\tglobal_load_dwordx4 %1, %5, %8\n\tglobal_load_dwordx4 %2, %6, %8\n\t"
                   "global_load_dwordx4 %3, %7, %8"
                   : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
                   : "v"(ro[0] + colb), "v"(ro[1] + colb), "v"(ro[2] + colb), "v"(ro[3] + colb), "s"(g.X)
                   : "memory");
    };
    const unsigned xs0 = (unsigned)(SNA * WA_BYTES + ((oct * 2) * 256 + (2 * half) * 64 + c) * 16);
    auto store_half = [&](int t, int ch, const f32x4* d) {
      char* sb = smem + (t % SNB) * WB_BYTES + xs0;
      float a[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a[q][0] = d[q][0]; a[q][1] = d[q][1]; a[q][2] = d[q][2]; a[q][3] = d[q][3];
        wg_swap(a[q][0], a[q][2]);
        wg_swap(a[q][1], a[q][3]);
      }
#pragma unroll
      for (int sx = 0; sx < 2; ++sx) {
        const float kv[8] = {a[0][sx], a[1][sx], a[2][sx], a[3][sx], a[0][sx + 2], a[1][sx + 2], a[2][sx + 2], a[3][sx + 2]};
        f32x4 hi, lo;
        wg_split8(kv, hi, lo);
        *reinterpret_cast<f32x4*>(sb + ch * 512 + sx * 1024) = hi;
        *reinterpret_cast<f32x4*>(sb + ch * 512 + sx * 1024 + 4096) = lo;
      }
    };
    const char* abase = g.img + ((int64_t)ks0 * nIT + itile) * WA_BYTES + (wave * 64 + lane) * 16;
    auto issue_a = [&](int t, bool live) {                     // 16 KiB by 256 threads: four 4 KiB pieces
      char* sa = smem + (t % SNA) * WA_BYTES + wave * 1024;
      const char* src = live ? abase + (int64_t)t * nIT * WA_BYTES : g.img;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_global_load_lds((gptr_f)(live ? src + j * 4096 : src), (lptr_f)(sa + j * 4096), 16, 0, 0);
    };
#define PP_ALL(s) "+v"(s.v[0]), "+v"(s.v[1]), "+v"(s.v[2]), "+v"(s.v[3]), "+v"(s.v[4]), "+v"(s.v[5]), "+v"(s.v[6]), "+v"(s.v[7])
    // prologue: images 0 and 1 requested, X(0) -> stage 0 (plain loads: the compiler's wait drains the DMA pieces too), X(1), X(2), X(3)
    // requested into the three register sets, X(1) waited for
    XSet sa_, sb_, sc_;
    issue_a(0, true);
    issue_a(1, nk > 1);
    {
      const char* xb = reinterpret_cast<const char*>(g.X);
      const u32x4 ro = row_offsets(0);
      XSet r0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        r0.v[q] = *reinterpret_cast<const f32x4*>(xb + ro[q] + colb0);
        r0.v[4 + q] = *reinterpret_cast<const f32x4*>(xb + ro[q] + colb1);
      }
      store_half(0, 0, &r0.v[0]);
      store_half(0, 1, &r0.v[4]);
    }
    { const u32x4 ro = row_offsets(1); load_half(ro, colb0, &sa_.v[0]); load_half(ro, colb1, &sa_.v[4]); }
    { const u32x4 ro = row_offsets(2); load_half(ro, colb0, &sb_.v[0]); load_half(ro, colb1, &sb_.v[4]); }
    { const u32x4 ro = row_offsets(3); load_half(ro, colb0, &sc_.v[0]); load_half(ro, colb1, &sc_.v[4]); }
    asm volatile("s_waitcnt vmcnt(16)" : PP_ALL(sa_) : : "memory");          // X(1) is here (the images are older: landed)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    slot_end();                                               // ---- tile 0 complete
    // k-step s: r = X(s+1) (arrived), r_next = X(s+2) (in flight, waited for at the end)
    auto kstep = [&](int s, XSet& r, XSet& r_next) {
      const bool st_ok = s + 1 < nk;
      const u32x4 ro = row_offsets(s + 4);
      // ---- slot 2s: the image DMA first (the end-of-k-step wait leaves everything younger than it in flight), column half 0
#ifdef PPW_NOPROD
      slot_end(); slot_end(); return;
#endif
      issue_a(s + 2, s + 2 < nk);
      if (st_ok) store_half(s + 1, 0, &r.v[0]);
      __builtin_amdgcn_sched_barrier(0);
      load_half(ro, colb0, &r.v[0]);                          // X(s+4), column half 0 (issued after the split read the registers)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      slot_end();
      // ---- slot 2s+1: column half 1
      if (st_ok) store_half(s + 1, 1, &r.v[4]);
      __builtin_amdgcn_sched_barrier(0);
      load_half(ro, colb1, &r.v[4]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // in flight, oldest first: X(s+2) [8] | image s+1 [4], X(s+3) [8] | image s+2 [4], X(s+4) [8]: leave the 20 youngest
      asm volatile("s_waitcnt vmcnt(20)" : PP_ALL(r_next) : : "memory");
      slot_end();
    };
    // (the remainder k-steps sit INSIDE the rotation loop: separate tail copies are laid out where tools/asm_lint.py's linear scan cannot
    // see the waits that precede them)
#pragma unroll 1
    for (int s = 0; s < nk; s += 3) {
      kstep(s, sa_, sb_);
      if (s + 1 < nk) kstep(s + 1, sb_, sc_);
      if (s + 2 < nk) kstep(s + 2, sc_, sa_);
    }
    slot_end();                                               // slot 2 nk: group 1's last compute phase
    asm volatile("s_waitcnt vmcnt(0)" : PP_ALL(sa_), PP_ALL(sb_), PP_ALL(sc_) : : "memory");
#undef PP_ALL
    return;
  }

  // =============================================================== consumers: 2 x 4 waves of 64 (E) x 64 (D)
  const int cw = wave - 4, wm = cw >> 2, wn = cw & 3;
  const int r16 = lane & 15, kg = lane >> 4;
  const unsigned fa_hi = lds0 + ((kg * 2) * 128 + 16 * wm + r16) * 16, fa_lo = fa_hi + 2048;
  const unsigned fb_hi = lds0 + SNA * WA_BYTES + ((kg * 2) * 256 + 16 * wn + r16) * 16, fb_lo = fb_hi + 4096;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 x[WNF];
  slot_end();                                                 // ---- tile 0 complete (producers' prologue)
  const bool late = wm != 0;                                  // group 1 runs the same loop one slot later
  if (late) slot_end();
#pragma unroll 1
  for (int s = 0; s < nk; ++s) {
    const unsigned soa = (unsigned)((s % SNA) * WA_BYTES), sob = (unsigned)((s % SNB) * WB_BYTES);
    WG_READ8(x, 4, 8, fa_lo + soa, fb_hi + sob);
    WG_READ8(x, 0, 12, fa_hi + soa, fb_lo + sob);
    WG_WAIT8(0, x, 4, 8);
    WG_WAIT8(0, x, 0, 12);
    slot_end();
#ifndef PPW_NOMMA
    wg_term(x, 4, 8, acc);                                    // lo*hi
    wg_term(x, 0, 12, acc);                                   // hi*lo
    wg_term(x, 0, 8, acc);                                    // hi*hi
#endif
    slot_end();
  }
  if (!late) slot_end();
  // ---- epilogue: tile (ja, jb) of a lane is row 4 (16 wm + 4 (lane >> 4) + e) + ja, column 4 (16 wn + (lane & 15)) + jb: 16-byte stores
  float* out = g.out + (int64_t)slab * g.E * g.D;
#pragma unroll
  for (int ja = 0; ja < 4; ++ja)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t i = i0 + 4 * (16 * wm + 4 * kg + e) + ja;
      const int64_t n = n0 + 4 * (16 * wn + r16);
      *reinterpret_cast<f32x4*>(out + i * g.D + n) = f32x4{acc[ja][0][e], acc[ja][1][e], acc[ja][2][e], acc[ja][3][e]};
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// BOTH operands as images (round 4).  The kernels above read X raw and split it on its way into LDS: four E-side tiles do that work for
// every X element, and it is half of the loop (stamped: the consumers alone and the producers alone take the same time).  Here the bag
// has been laid down ONCE as the D-side operand image (prep job kind 9: per 32-row k-step and 256-column block one 32 KiB tile, byte for
// byte the LDS stage of the kernels above), by workgroups that ride in a launch of the forward that leaves the chip idle, and dPRE^T's
// image is in BAG order (rows that did not take part: zeros) - so a k-step is 48 KiB of linear LDS-DMA, nothing goes through registers,
// every wave only reads fragments and multiplies.  8 waves as 2 x 4 of 64 x 64 (bag_wgrad_kernel's consumer schedule: the hi*lo term of
// tile t-1 under the first fragment reads of tile t), 3 stages of [A 16 KiB | B 32 KiB]: tile t+2 is requested when tile t-1 retires.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int DNS = 3, DSTAGE = WA_BYTES + WB_BYTES, DRING = DNS * DSTAGE;                        // 144 KiB

__global__ __launch_bounds__(WTHREADS) void bag_wgrad_dma_kernel(WgradArgs g, const char* ximg, int side_blocks, Merge2Side side, WgradTail tail) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (wg_rider_block(smem, side_blocks, side, tail)) return;
  const unsigned bx = blockIdx.x - (unsigned)side_blocks;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int nJ = (int)(g.D / WBN), nIT = (int)(g.E / WBI), nT = nIT * nJ;
  const int xcd = bx & 7, sidx = bx >> 3;
  const int slab = (sidx / nT) * 8 + xcd, tile = sidx % nT;
  if (slab >= g.splits) return;
  const int itile = tile / nJ, jtile = tile % nJ;
  const int64_t i0 = (int64_t)itile * WBI, n0 = (int64_t)jtile * WBN;
  const int ks0 = slab * g.kps;
  const int nk = (ks0 + g.kps < g.ksteps ? ks0 + g.kps : g.ksteps) - ks0;

  // a k-step: 16 KiB of dPRE^T + 32 KiB of X, 6 x 16 B per thread.  `live` false: the pieces come from ONE address and land in a stage nobody
  // reads any more, so that every iteration has the same VMEM count and the vmcnt waits need no branch.
  const char* abase = g.img + ((int64_t)ks0 * nIT + itile) * WA_BYTES + tid * 16;
  const char* bbase = ximg + ((int64_t)ks0 * nJ + jtile) * WB_BYTES + tid * 16;
  auto issue = [&](int t, bool live) {
    char* sa = smem + (t % DNS) * DSTAGE + wave * 1024;
    const char* a = live ? abase + (int64_t)t * nIT * WA_BYTES : g.img;
    const char* b = live ? bbase + (int64_t)t * nJ * WB_BYTES : g.img;
    __builtin_amdgcn_global_load_lds((gptr_f)a, (lptr_f)sa, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_f)(live ? a + 8192 : a), (lptr_f)(sa + 8192), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds((gptr_f)(live ? b + j * 8192 : b), (lptr_f)(sa + WA_BYTES + j * 8192), 16, 0, 0);
  };
  issue(0, true);
  issue(1, nk > 1);

  const int r16 = lane & 15, kg = lane >> 4;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_f)smem;
  const unsigned fa_hi = lds0 + ((kg * 2) * 128 + 16 * wm + r16) * 16, fa_lo = fa_hi + 2048;
  const unsigned fb_hi = lds0 + WA_BYTES + ((kg * 2) * 256 + 16 * wn + r16) * 16, fb_lo = fb_hi + 4096;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 x[WNF];
#if defined(WGD_NOREAD) || defined(WGD_HALFREAD)
  for (int q = 0; q < WNF; ++q) x[q] = f32x4{1.f, 2.f, 3.f, 4.f};
#endif
#pragma unroll 1
  for (int t = 0; t < nk; ++t) {
#ifndef WGD_NODMA
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");           // my pieces of tile t have landed (tile t+1's six may be in flight)
#endif
    __builtin_amdgcn_s_barrier();                              // ... and everybody's; every wave is past its reads of tile t-1
    const unsigned so = (unsigned)((t % DNS) * DSTAGE);
#ifndef WGD_NODMA
    issue(t + 2, t + 2 < nk);                                  // -> the stage of tile t-1
#endif
#ifndef WGD_NOREAD
    WG_READ8(x, 4, 8, fa_lo + so, fb_hi + so);
#endif
    __builtin_amdgcn_sched_barrier(0);
#ifndef WGD_NOMMA
    if (t > 0) wg_term(x, 0, 12, acc);                         // hi*lo of tile t-1 (operands still in registers)
#endif
    __builtin_amdgcn_sched_barrier(0);
#ifndef WGD_NOREAD
    WG_WAIT8(0, x, 4, 8);
#ifndef WGD_HALFREAD
    WG_READ8(x, 0, 12, fa_hi + so, fb_lo + so);
#endif
#endif
    __builtin_amdgcn_sched_barrier(0);
#ifndef WGD_NOMMA
    wg_term(x, 4, 8, acc);                                     // lo*hi
#endif
    __builtin_amdgcn_sched_barrier(0);
#ifndef WGD_NOREAD
    WG_WAIT8(0, x, 0, 12);
#endif
    __builtin_amdgcn_sched_barrier(0);
#ifndef WGD_NOMMA
    wg_term(x, 0, 8, acc);                                     // hi*hi
#endif
  }
  wg_term(x, 0, 12, acc);                                      // hi*lo of the last tile
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (the last dummy pieces: nothing may land after the workgroup's LDS is handed on)
  float* out = g.out + (int64_t)slab * g.E * g.D;
#pragma unroll
  for (int ja = 0; ja < 4; ++ja)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t i = i0 + 4 * (16 * wm + 4 * kg + e) + ja;
      const int64_t n = n0 + 4 * (16 * wn + r16);
      *reinterpret_cast<f32x4*>(out + i * g.D + n) = f32x4{acc[ja][0][e], acc[ja][1][e], acc[ja][2][e], acc[ja][3][e]};
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// dPRE -> matrix-core image + column-sum partials.  One workgroup per (32-row k-step, 256 columns); thread -> (4 columns, one row
// octet): 8 x (16 B of dH + 8 B of dact16) in flight per thread, ~10 waves per CU.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rows_dpre_image_kernel(const float* __restrict__ dH, const _Float16* __restrict__ dact,
                                                             const int64_t* __restrict__ rows, int64_t L, int E, char* __restrict__ img,
                                                             float* __restrict__ part, int side_blocks, Merge2Side side, int dh_compact,
                                                             const uint8_t* __restrict__ keep, BagBatch bb) {
  __shared__ __attribute__((aligned(16))) float lds[M2_PARTIALS_LDS > 3 * 256 * 4 ? M2_PARTIALS_LDS : 3 * 256 * 4];
  if (blockIdx.z) {      // (common.hpp: a bag of an accumulation window)
    MHIMX_BAG(dH); MHIMX_BAG(dact); MHIMX_BAG(rows); MHIMX_BAG(img); MHIMX_BAG(part); MHIMX_BAG(keep);
    if (side_blocks > 0) bag_move(side, bb);
  }
  if ((int)blockIdx.x < side_blocks) {          // a parked Merge-backward tail rides along (stage 1), its workgroups first
    merge2_side_stage(1, (int)blockIdx.x, lds, side);
    return;
  }
  typedef _Float16 h4v __attribute__((ext_vector_type(4)));
  const int nCB = E / 256 > 0 ? (E + 255) / 256 : 1;            // column blocks of 256 (E % 128 == 0: the last one may be half)
  const int bid = (int)blockIdx.x - side_blocks;
  const int ks = bid / nCB, cb = bid % nCB;
  const int nIT = E / WBI;
  const int tid = threadIdx.x, koct = tid >> 6, cq = cb * 64 + (tid & 63);      // columns 4 cq .. 4 cq + 3
  const bool live = cq < E / 4;
  f32x4 cs = f32x4{0.f, 0.f, 0.f, 0.f};
  if (live) {
    const int itile = cq >> 5, cl = cq & 31;                    // slots j * 32 + cl of tile itile
    int64_t r[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int64_t l = (int64_t)ks * WBK + koct * 8 + q;
      r[q] = l < L ? (rows ? rows[l] : l) : -1;
      if (keep && l < L && !keep[l]) r[q] = -1;                 // bag order: a row that did not take part is a zero row of the image
    }
    f32x4 gv[8];
    h4v dv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int64_t rr = r[q] < 0 ? 0 : r[q];
      const int64_t lq = (int64_t)ks * WBK + koct * 8 + q;                        // dh_compact: dH row = list position (rows address dact only)
      gv[q] = reinterpret_cast<const f32x4*>(dH + (dh_compact ? (lq < L ? lq : 0) : rr) * E)[cq];
      dv[q] = dact ? reinterpret_cast<const h4v*>(dact + rr * E)[cq] : h4v{(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      gv[q] = r[q] < 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : gv[q] * f32x4{(float)dv[q][0], (float)dv[q][1], (float)dv[q][2], (float)dv[q][3]};
      cs += gv[q];
    }
    char* tile = img + ((int64_t)ks * nIT + itile) * WA_BYTES + (koct * 2 * 128 + cl) * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float kv[8] = {gv[0][j], gv[1][j], gv[2][j], gv[3][j], gv[4][j], gv[5][j], gv[6][j], gv[7][j]};
      f32x4 hi, lo;
      wg_split8(kv, hi, lo);
#ifdef MHIMX_IMG_WT
      st_f4_wt(reinterpret_cast<float*>(tile + j * 512), hi);
      st_f4_wt(reinterpret_cast<float*>(tile + j * 512 + 2048), lo);
#else
      *reinterpret_cast<f32x4*>(tile + j * 512) = hi;
      *reinterpret_cast<f32x4*>(tile + j * 512 + 2048) = lo;
#endif
    }
  }
  if (part) {                                                   // the four octets' column sums -> one partial row per k-step
    if (koct > 0) reinterpret_cast<f32x4*>(lds)[(koct - 1) * 64 + (tid & 63)] = cs;
    __syncthreads();
    if (koct == 0 && live) {
      const f32x4* o = reinterpret_cast<const f32x4*>(lds) + (tid & 63);
      cs = (cs + o[0]) + (o[64] + o[128]);
      reinterpret_cast<f32x4*>(part + (int64_t)ks * E)[cq] = cs;
    }
  }
}

static bool wgrad_shape_ok(int64_t L, int64_t E, int64_t D, int64_t ldx, int64_t n_bag_rows) {
  return L >= 1 && E >= WBI && E % WBI == 0 && D >= WBN && D % WBN == 0 && ldx % 4 == 0 && ldx >= D &&
         n_bag_rows * ldx * 4 < ((int64_t)1 << 32);
}

}  // namespace mhimx

using namespace mhimx;

extern "C" int64_t mhimx_wgrad_image_bytes(int64_t L, int64_t E) { return cdiv(L, WBK) * WBK * E * 4; }

// slabs the product will use for L rows, and the workspace (floats) they take
// (Round 3: slab counts chosen to fill whole rounds of 256 workgroups - 24 tiles x 21 slabs of 66 k-steps instead of 24 x 15 of 96 for
// the [1536 x 512] gradient of c3 - measured the SAME launch time, 264 vs 242 us on a slower box: with 12 row tiles re-reading X and 2
// column tiles re-reading the image the launch moves ~1.6 GB through L2 and is bound there, not by how its workgroups fill the CUs.)
static int wgrad_plan(int64_t L, int64_t E, int64_t D, int* kps_out) {
  const int ksteps = (int)cdiv(L, WBK);
  const int64_t tiles = (E / WBI) * (D / WBN);
  int64_t want = cdiv(256, tiles);                            // one workgroup per CU
  if (want > cdiv(ksteps, 4)) want = cdiv(ksteps, 4);         // at least 4 k-steps each
  if (want < 1) want = 1;
  int kps = (int)cdiv(ksteps, want);
  if (kps * WBK > W_MAX_CHUNK) kps = W_MAX_CHUNK / WBK;
  *kps_out = kps;
  return (int)cdiv(ksteps, kps);
}
extern "C" int64_t mhimx_wgrad_ws_floats(int64_t L, int64_t E, int64_t D) {
  int kps;
  return (int64_t)wgrad_plan(L, E, D, &kps) * E * D;
}

static int rows_dpre_image_impl(void* stream, const float* dH, const void* dact16, const int64_t* rows, int64_t L, int64_t E, void* img,
                                float* colsum_out, int32_t accumulate, void* ws, int64_t ws_bytes, mhimx_reduce_list* defer, int dh_compact,
                                const uint8_t* keep = nullptr) {
  MHIMX_CHECK_ARG(dH && img && L >= 1 && E >= WBI && E % WBI == 0 && aligned16(dH) && aligned16(img) &&
                      (reinterpret_cast<uintptr_t>(dact16) & 7) == 0,
                  "rows_dpre_image: E must be a multiple of 128, buffers aligned");
  const int64_t ksteps = cdiv(L, WBK), nblk = ksteps, ncb = cdiv(E, 256);
  MHIMX_CHECK_ARG(!colsum_out || (ws && ws_bytes >= nblk * E * 4), "rows_dpre_image: workspace too small (%lld bytes)", (long long)(nblk * E * 4));
  Merge2Side side = {};
  int side_blocks = 0;
  if (defer && defer->side.pending == 1) {       // a parked Merge-backward tail: stage 1 rides in this launch
    memcpy(&side, defer->side.blob, sizeof(side));
    side_blocks = merge2_side_blocks(1, side);
    defer->side.pending = 2;
  }
  MHIMX_CHECK_ARG(cur_batch().n == 0 || !colsum_out || defer, "rows_dpre_image: a bag-batched launch queues its column sums");
  hipLaunchKernelGGL(rows_dpre_image_kernel, bgrid((unsigned)(ksteps * ncb + side_blocks)), dim3(256), 0, (hipStream_t)stream, dH, (const _Float16*)dact16,
                     rows, L, (int)E, (char*)img, colsum_out ? (float*)ws : nullptr, side_blocks, side, dh_compact, keep, cur_batch());
  MHIMX_LAUNCH_CHECK();
  if (colsum_out && !defer_push(defer, reduce_job_parts((const float*)ws, nblk, E, E, colsum_out, accumulate))) {
    const int rc = reduce_parts_now((hipStream_t)stream, (const float*)ws, nblk, E, E, colsum_out, accumulate);
    if (rc) return rc;
  }
  return 0;
}

extern "C" int mhimx_rows_dpre_image(void* stream, const float* dH, const void* dact16, const int64_t* rows, int64_t L, int64_t E, void* img,
                                     float* colsum_out, int32_t accumulate, void* ws, int64_t ws_bytes, mhimx_reduce_list* defer) {
  return rows_dpre_image_impl(stream, dH, dact16, rows, L, E, img, colsum_out, accumulate, ws, ws_bytes, defer, 0);
}
extern "C" int mhimx_rows_dpre_image_k(void* stream, const float* dH, const void* dact16, const uint8_t* keep, int64_t N, int64_t E, void* img,
                                       float* colsum_out, int32_t accumulate, void* ws, int64_t ws_bytes, mhimx_reduce_list* defer) {
  return rows_dpre_image_impl(stream, dH, dact16, nullptr, N, E, img, colsum_out, accumulate, ws, ws_bytes, defer, 0, keep);
}
extern "C" int mhimx_rows_dpre_image_c(void* stream, const float* dH_compact, const void* dact16, const int64_t* rows, int64_t L, int64_t E,
                                       void* img, float* colsum_out, int32_t accumulate, void* ws, int64_t ws_bytes, mhimx_reduce_list* defer) {
  return rows_dpre_image_impl(stream, dH_compact, dact16, rows, L, E, img, colsum_out, accumulate, ws, ws_bytes, defer, 1);
}

// slabs PER BAG of a launch over n_bags bags of L rows each: enough workgroups for the chip, a row table that fits (W_MAX_CHUNK rows)
static int wgrad_plan_multi(int64_t L, int64_t E, int64_t D, int n_bags, int* kps_out) {
  const int ksteps = (int)cdiv(L, WBK);
  const int64_t tiles = (E / WBI) * (D / WBN);
  int64_t spb = cdiv(256, tiles * n_bags);
  const int64_t need = cdiv((int64_t)ksteps * WBK, W_MAX_CHUNK);
  if (spb < need) spb = need;
  if (spb > ksteps) spb = ksteps;
  int kps = (int)cdiv(ksteps, spb);
  *kps_out = kps;
  return (int)cdiv(ksteps, kps);
}
extern "C" int64_t mhimx_wgrad_multi_ws_floats(int64_t L, int64_t E, int64_t D, int32_t n_bags) {
  int kps;
  return (int64_t)wgrad_plan_multi(L, E, D, n_bags, &kps) * n_bags * E * D;
}

static int bag_wgrad_impl(void* stream, const mhimx_bag_wgrad_args* bags, int n_bags) {
  const mhimx_bag_wgrad_args* a = bags;
  MHIMX_CHECK_ARG(a->img && a->X && a->C && a->ws && aligned16(a->img) && aligned16(a->X) && aligned16(a->C) && aligned16(a->ws) && a->ldc % 4 == 0,
                  "bag_wgrad: null / unaligned operand");
  MHIMX_CHECK_ARG(wgrad_shape_ok(a->L, a->E, a->D, a->ldx, a->n_bag_rows),
                  "bag_wgrad: needs E %% 128 == 0, D %% 256 == 0, 16-byte aligned rows and a bag below 4 GiB");
  WgradArgs g;
  g.img = (const char*)a->img; g.X = a->X; g.ldx = a->ldx; g.rows = a->rows; g.L = a->L; g.E = a->E; g.D = a->D;
  g.ksteps = (int)cdiv(a->L, WBK);
  WgradBags mb = {};
  if (n_bags == 1) {
    g.splits = wgrad_plan(a->L, a->E, a->D, &g.kps);
    mb.spb = g.splits;
  } else {
    mb.spb = wgrad_plan_multi(a->L, a->E, a->D, n_bags, &g.kps);
    g.splits = mb.spb * n_bags;
  }
  for (int b = 0; b < n_bags; ++b) {
    const mhimx_bag_wgrad_args& q = bags[b];
    MHIMX_CHECK_ARG(q.img && q.X && aligned16(q.img) && aligned16(q.X) && q.L == a->L && q.E == a->E && q.D == a->D && q.ldx == a->ldx &&
                        q.n_bag_rows * q.ldx * 4 < ((int64_t)1 << 32) && (q.rows != nullptr) == (a->rows != nullptr),
                    "bag_wgrad_multi: bag %d: null / unaligned operand, or not the shape of bag 0 (L, E, D, ldx, row ids)", b);
    mb.img[b] = (const char*)q.img; mb.X[b] = q.X; mb.rows[b] = q.rows;
  }
  MHIMX_CHECK_ARG(a->ws_floats >= (int64_t)g.splits * a->E * a->D, "bag_wgrad: workspace too small (%lld floats)", (long long)((int64_t)g.splits * a->E * a->D));
  g.out = a->ws;
  const size_t smem = WRING + (size_t)g.kps * WBK * 4;
  MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)bag_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WRING + W_MAX_CHUNK * 4)));
  Merge2Side side = {};
  int side_blocks = 0;
  mhimx_reduce_list* defer = a->defer;
  if (defer && defer->side.pending == 2 && smem >= M2_SIDE_LDS * sizeof(float)) {
    memcpy(&side, defer->side.blob, sizeof(side));           // a parked Merge-backward tail: stage 2 rides in this launch
    side_blocks = merge2_side_blocks(2, side);
    if (side_blocks % 8 == 0) defer->side.pending = 3;
    else side_blocks = 0;
  }
  const int64_t tiles = (a->E / WBI) * (a->D / WBN);
  dim3 grid((unsigned)(8 * tiles * cdiv(g.splits, 8) + side_blocks));
  static const bool ws_form = getenv("MHIMX_WGRAD_UNIFORM") == nullptr;
  static const bool pp_form = ws_form && getenv("MHIMX_WGRAD_PP") != nullptr;
  WgradTail tail = {};
  MHIMX_CHECK_ARG(!a->ximg || (n_bags == 1 && !a->rows && aligned16(a->ximg) && a->L == a->n_bag_rows),
                  "bag_wgrad: the bag image (ximg) goes with a bag-ordered dPRE image: one bag, no row list, L = n_bag_rows");
  if (a->ride_tail && defer && ((ws_form && !pp_form) || a->ximg) && n_bags == 1 && (defer->n > 0 || defer->side.pending == 3)) {
    // the reductions queued so far and (behind the stage-2 gate) the tail's last stage: trailing workgroups of this launch
    for (int i = 0; i < defer->n; ++i) {
      const mhimx_reduce_job& j = defer->j[i];
      MHIMX_CHECK_ARG(j.parts && j.out && j.G > 0 && (j.kind == 0 ? j.W > 0 : (j.kind == 1 && j.K1 > 0 && j.K2 > 0)), "bag_wgrad: bad queued reduction %d", i);
    }
    tail.first = (int)grid.x;
    tail.n_reduce = reduce_table_fill(tail.t, defer->j, defer->n, WTHREADS);
    defer->n = 0;
    if (defer->side.pending == 3 && side_blocks > 0) {          // (its stage 2 rides at the front of THIS grid: the gate counts those blocks)
      tail.stage3 = merge2_side_blocks(3, side);
      tail.gate_want = side_blocks;
      defer->side.pending = 0;
    }
    grid.x += (unsigned)(tail.n_reduce + tail.stage3);
  }
  // The specialised-wave form (bag_wgrad_ws_kernel, ~5 % faster) is the default again.  It was opt-in for a while: with two processes
  // time-slicing one GPU it ended in a GPU memory access fault on the long TransMIL-shaped launches (tools/two_proc_c3.sh: 3 of 6 runs
  // died).  Cause: its last asm wait named 6 of the 24 prefetch registers, so the compiler handed the other 18 out again while the
  // (unused, clamped) last prefetch loads were still in flight, and their data landed on the row offsets of the next asm loads - only
  // when latency was stretched.  Fixed there, checked by tools/asm_lint.py (tests/test_isa_lint_cpu.py); 12 of 12 two-process runs
  // finish.  MHIMX_WGRAD_UNIFORM=1 selects the uniform kernel (experiments).
  // Round 3: bag_wgrad_pp_kernel (specialised waves + ping-pong consumers, 768 threads: what took the projection from 73 to 63 us)
  // measures the SAME as the specialised-wave kernel here (34.7 vs 34.5 us same-box; consumers alone 27.8, producers alone 27.8, both 37:
  // ~13 us of the launch are the row table, the first tiles and the 33 MB of slab stores, outside the loop either form pipelines), so the
  // round-2 kernel stays the default; MHIMX_WGRAD_PP=1 selects the ping-pong form, MHIMX_WGRAD_UNIFORM=1 the uniform one.
  MHIMX_CHECK_ARG(n_bags == 1 || (ws_form && !pp_form), "bag_wgrad_multi: only the default (specialised-wave) kernel takes several bags");
  if (a->ximg) {                                               // both operands as images: nothing but linear DMA and fragments in the loop
    MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)bag_wgrad_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DRING)));
    hipLaunchKernelGGL(bag_wgrad_dma_kernel, grid, dim3(WTHREADS), DRING, (hipStream_t)stream, g, (const char*)a->ximg, side_blocks, side, tail);
  } else if (pp_form) {
    const size_t smem2 = SRING + (size_t)g.kps * WBK * 4;
    MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)bag_wgrad_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SRING + W_MAX_CHUNK * 4)));
    hipLaunchKernelGGL(bag_wgrad_pp_kernel, grid, dim3(PP_THREADS), smem2, (hipStream_t)stream, g, side_blocks, side);
  } else if (ws_form) {
    const size_t smem2 = SRING + (size_t)g.kps * WBK * 4;
    MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)bag_wgrad_ws_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SRING + W_MAX_CHUNK * 4)));
    hipLaunchKernelGGL(bag_wgrad_ws_kernel, grid, dim3(WTHREADS), smem2, (hipStream_t)stream, g, mb, side_blocks, side, tail);
  } else
  hipLaunchKernelGGL(bag_wgrad_kernel, grid, dim3(WTHREADS), smem, (hipStream_t)stream, g, side_blocks, side);
  MHIMX_LAUNCH_CHECK();
  if (!defer_push(defer, reduce_job_slabs(a->ws, g.splits, a->E, a->D, a->ldc, a->C, a->accumulate))) {
    const int rc = reduce_slabs_now((hipStream_t)stream, a->ws, a->C, a->E, a->D, a->ldc, g.splits, a->accumulate);
    if (rc) return rc;
  }
  return 0;
}

extern "C" int mhimx_bag_wgrad(void* stream, const mhimx_bag_wgrad_args* a) {
  if (!a) return fail(-1, "bag_wgrad: null argument block");
  return bag_wgrad_impl(stream, a, 1);
}
extern "C" int mhimx_bag_wgrad_multi(void* stream, const mhimx_bag_wgrad_args* bags, int32_t n_bags) {
  MHIMX_CHECK_ARG(bags && n_bags >= 1 && n_bags <= W_MAX_BAGS, "bag_wgrad_multi: 1..%d bags", W_MAX_BAGS);
  return bag_wgrad_impl(stream, bags, n_bags);
}
