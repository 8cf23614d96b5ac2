// gemm_dma.hip — the fast GEMM core: fp32 operand tiles go HBM/L2 -> LDS by direct DMA
// (global_load_lds_dwordx4: no VGPR staging, 16 B per lane, 1 KiB per wave instruction), two LDS stages,
// two workgroups per CU; the fp32 -> (hi, lo) 16-bit split happens when the MFMA fragments are built.
//
//   gemm_nt_dma : C[M,N] = epi(A[rows?][M,K] . B[N,K]^T)     LDS image [row][32 k] fp32, 128-B rows.
//       The DMA writes LDS lane-linearly, so the bank-conflict swizzle is applied to the per-lane SOURCE
//       address and again on the fragment read (same involution: 16-B slot ^= (row>>1)&7): a 16-lane
//       ds_read_b128 group then covers 16 distinct 16-B slots of the 256-B bank row.
//   gemm_tn_dma : C[K1,K2] = A[M,K1]^T . B[rows?][M,K2]       LDS image [32 m][128 cols] fp32 exactly as in HBM
//       (coalesced 512-B row segments); the k-contiguous fragment a lane needs is a strided column of that
//       image: 8 x ds_read_b32 with consecutive lanes on consecutive banks.  No transposing stores.
// Requirements (else the register-staged kernels in gemm.hip run): K % 32 == 0 (nt); K1,K2 % 128 == 0 (tn);
// 16-byte aligned rows.
#include <string.h>

#include <stdlib.h>
#include "common.hpp"
#include "mca2_rows.hpp"
#include "prep_jobs.hpp"

namespace mhimx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int DBM = 128, DBN = 128, DBK = 32, DTHREADS = 256;
constexpr int TILE_BYTES = 128 * DBK * 4;          // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
#ifndef MHIMX_NT_STAGES
#define MHIMX_NT_STAGES 2
#endif
constexpr int NSTAGE = MHIMX_NT_STAGES;           // LDS ring depth of the NT kernel (x 32 KiB); 2 => two workgroups per CU
// dynamic LDS of the NT kernel: the ring, or the epilogue's 128 x 136 fp32 output tile (68 KiB), whichever is larger
constexpr int NT_LDS_BYTES = NSTAGE * STAGE_BYTES > DBM * (DBN + 8) * 4 ? NSTAGE * STAGE_BYTES : DBM * (DBN + 8) * 4;
// 8 x ds_read_b128 from per-lane LDS byte addresses a[0..7] + off (issue only; pair with LDS_WAIT8 before use)
#define LDS_READ8(x, a, off)                                                                                         \
  asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"       \
               "ds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\tds_read_b128 %6, %14\n\tds_read_b128 %7, %15"          \
               : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7])    \
               : "v"(a[0] + (off)), "v"(a[1] + (off)), "v"(a[2] + (off)), "v"(a[3] + (off)), "v"(a[4] + (off)),            \
                 "v"(a[5] + (off)), "v"(a[6] + (off)), "v"(a[7] + (off))                                                   \
               : "memory")
#define LDS_WAIT8(x)                                                                                                  \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                 \
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])            \
               :                                                                                                      \
               : "memory")

constexpr int TN_STAGES = 2;                       // LDS ring depth of the TN kernel (x 32 KiB)
constexpr int MAX_TN_CHUNK = 4096;                 // rows of the reduction one workgroup may own (row-id table in LDS)

__device__ float g_zero_row[256];                  // 1 KiB of zeros: source for out-of-range reduction rows

template <int PREC> struct Frag;
template <> struct Frag<MHIMX_PREC_BF16X3> {
  using V8 = b8;
  static constexpr int NA = 2, NB = 2;
  static MHIMX_DEV void split2(const f4& a, const f4& b, V8& hi, V8& lo) {
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    split(x, hi, lo);
  }
  static MHIMX_DEV void split(const float (&x)[8], V8& hi, V8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const __bf16 h = (__bf16)x[i];
      hi[i] = h;
      lo[i] = (__bf16)(x[i] - (float)h);
    }
  }
};
template <> struct Frag<MHIMX_PREC_F16S> {
  using V8 = h8;
  static constexpr int NA = 1, NB = 2;
  static MHIMX_DEV void split2(const f4& a, const f4& b, V8& hi, V8& lo) {
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    split(x, hi, lo);
  }
  static MHIMX_DEV void split(const float (&x)[8], V8& hi, V8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const _Float16 h = (_Float16)x[i];
      hi[i] = h;
      lo[i] = (_Float16)(x[i] - (float)h);
    }
  }
};

template <int PREC>
MHIMX_DEV void mma12(const typename Frag<PREC>::V8 (&ah)[2], const typename Frag<PREC>::V8 (&al)[2],
                     const typename Frag<PREC>::V8 (&bh)[2], const typename Frag<PREC>::V8 (&bl)[2], f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      if constexpr (PREC == MHIMX_PREC_BF16X3) {
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
      } else {
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
      }
    }
}

// N x ds_read_b128 from per-lane LDS byte addresses a[i] + off (issue only; pair with lds_wait<N> before use).
// Written as asm so the compiler does not tie them to the in-flight LDS-DMA (which would cost a vmcnt(0) drain).
template <int N>
MHIMX_DEV void lds_read(f4 (&x)[N], const unsigned (&a)[N], unsigned off, unsigned flip) {
  static_assert(N == 6 || N == 8, "6 or 8 reads");
  if constexpr (N == 8) {
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"
                 "ds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\tds_read_b128 %6, %14\n\tds_read_b128 %7, %15"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7])
                 : "v"((a[0] + off) ^ flip), "v"((a[1] + off) ^ flip), "v"((a[2] + off) ^ flip), "v"((a[3] + off) ^ flip), "v"((a[4] + off) ^ flip), "v"((a[5] + off) ^ flip),
                   "v"((a[6] + off) ^ flip), "v"((a[7] + off) ^ flip)
                 : "memory");
  } else {
    asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %8\n\tds_read_b128 %3, %9\n\t"
                 "ds_read_b128 %4, %10\n\tds_read_b128 %5, %11"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5])
                 : "v"((a[0] + off) ^ flip), "v"((a[1] + off) ^ flip), "v"((a[2] + off) ^ flip), "v"((a[3] + off) ^ flip), "v"((a[4] + off) ^ flip), "v"((a[5] + off) ^ flip)
                 : "memory");
  }
}
template <int N>
MHIMX_DEV void lds_wait(f4 (&x)[N]) {
  if constexpr (N == 8) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
                 :
                 : "memory");
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]) : : "memory");
  }
}

template <int PREC, int NT>
MHIMX_DEV void mma_tile(const typename Frag<PREC>::V8 (&ah)[2], const typename Frag<PREC>::V8 (&al)[2],
                        const typename Frag<PREC>::V8 (&bh)[NT], const typename Frag<PREC>::V8 (&bl)[NT], f32x16 (&acc)[2][NT]) {
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if constexpr (PREC == MHIMX_PREC_BF16X3) {
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
      } else {
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
      }
    }
}

MHIMX_DEV void dma16(const float* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

// =================================================================================================
// NT
// =================================================================================================
// NW = 4: 2x2 waves, 64x64 per wave (12 MFMAs per 16-deep k slice);  NW = 8: 2x4 waves, 64x32 per wave (6 MFMAs).
// Eight waves put two waves on every SIMD even when a CU holds a single workgroup: one wave's DMA issue (~100
// cycles per LDS-DMA instruction), fragment reads and fp32->bf16 splitting then overlap the other wave's MFMAs.
// PAIRED: both operands arrive as "paired planes" (mhimx_pair_planes): every 8 consecutive k of a row are stored as
// 16 B of bf16 hi followed by 16 B of bf16 lo, i.e. the same 128 B per row per 32-deep k-step and therefore the SAME LDS
// image, DMA pattern and swizzle as the fp32 form — but the two 16-B slots a lane reads ARE its (hi, lo) MFMA fragments:
// no VALU conversion in the loop at all.
// ksteps > 0: split-K form — blockIdx.y owns k-steps [y*ksteps, (y+1)*ksteps) and writes its raw partial tile to slab y
// of g.ws ([M,N] each); reduce_slabs_kernel sums the slabs in a fixed order.  Used when a GEMM has few output tiles and a
// long reduction (the K loop is a serial chain of DMA round trips; 32 tiles on 256 CUs leave the chip idle).
template <int PREC, int NW, int PAIRED = 0>
__global__ __launch_bounds__(64 * NW) void gemm_nt_dma_kernel(mhimx_gemm_nt_args g, int ksteps) {
  using FR = Frag<PREC>;
  using V8 = typename FR::V8;
  constexpr int NT = NW == 4 ? 2 : 1;               // 32-column tiles per wave
  constexpr int NTHR = 64 * NW;
  constexpr int NDMA = 1024 / NTHR;                 // DMA instructions per thread per operand per k-step
  constexpr int NRD = 4 + 2 * NT;                   // ds_read_b128 per 16-deep k slice
  constexpr bool EPI_LDS_OK = NT_LDS_BYTES >= DBM * (DBN + 8) * 4;     // room for the epilogue's staged output tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = NW == 4 ? (wave >> 1) : (wave >> 2), wn = NW == 4 ? (wave & 1) : (wave & 3);
  // XCD-aware tile order: the dispatcher places linear block b on XCD b % 8 (speed only, never correctness).
  // All N-tiles of one M-tile are given to the SAME XCD back to back, so the A (patch-feature) rows are fetched
  // from HBM once and re-served by that XCD's L2.
  const int nN = (int)((g.N + DBN - 1) / DBN), nM = (int)((g.M + DBM - 1) / DBM);
  const int xcd = blockIdx.x & 7, sidx = blockIdx.x >> 3;
  const int m_tile = (sidx / nN) * 8 + xcd, n_tile = sidx % nN;
  if (m_tile >= nM) return;
  const int64_t m0 = (int64_t)m_tile * DBM, n0 = (int64_t)n_tile * DBN;

  // slot p = tid + NTHR*j of a [128 rows][8 slots] tile; LDS position is linear in p, the SOURCE is swizzled
  const float* asrc[NDMA];
  const float* bsrc[NDMA];
#pragma unroll
  for (int j = 0; j < NDMA; ++j) {
    const int p = tid + NTHR * j;
    const int row = p >> 3, slot = (p & 7) ^ ((row >> 1) & 7);
    int64_t m = m0 + row, n = n0 + row;
    if (m >= g.M) m = g.M - 1;                      // clamped rows feed accumulators that are never stored
    if (n >= g.N) n = g.N - 1;
    asrc[j] = g.A + (g.rows ? g.rows[m] : m) * g.lda + slot * 4;
    bsrc[j] = g.B + n * g.ldb + slot * 4;
  }
  auto issue = [&](int64_t k0, int stage) {
    char* sa = smem + stage * STAGE_BYTES + wave * 1024;
    char* sb = sa + TILE_BYTES;
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
      dma16(asrc[j] + k0, sa + j * (NTHR * 16));
      dma16(bsrc[j] + k0, sb + j * (NTHR * 16));
    }
  };

  f32x16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // Fragment addresses (LDS byte offsets, stage 0, ks = 0): 16-B slots s0, s0+1 with s0 = kh*2, swizzled by
  // (row>>1)&7.  ks = 1 flips slot bit 2 (address ^ 64).  fa[0..3]: A rows wm*64 + q*32 + r; fa[4..]: B rows.
  const int r = lane & 31, kh = lane >> 5;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  unsigned fa[NRD];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int ra = wm * 64 + q * 32 + r;
    fa[q * 2 + 0] = lds0 + ra * 128 + (((kh * 2) ^ ((ra >> 1) & 7)) << 4);
    fa[q * 2 + 1] = lds0 + ra * 128 + (((kh * 2 + 1) ^ ((ra >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    const int rb = wn * (32 * NT) + q * 32 + r;
    fa[4 + q * 2 + 0] = lds0 + TILE_BYTES + rb * 128 + (((kh * 2) ^ ((rb >> 1) & 7)) << 4);
    fa[4 + q * 2 + 1] = lds0 + TILE_BYTES + rb * 128 + (((kh * 2 + 1) ^ ((rb >> 1) & 7)) << 4);
  }

  // NSTAGE-deep LDS ring, ONE barrier per k-step, counted vmcnt so younger tiles stay in flight across it.
  // LDS reads are inline asm: a compiler-visible ds_read after an LDS-DMA makes hipcc drain the queue (vmcnt(0)).
  const int nk_all = (int)(g.K / DBK);
  const int kb = ksteps > 0 ? (int)blockIdx.y * ksteps : 0;
  const int nk = ksteps > 0 ? (nk_all - kb < ksteps ? nk_all - kb : ksteps) : nk_all;
#pragma unroll
  for (int j = 0; j < NDMA; ++j) { asrc[j] += (int64_t)kb * DBK; bsrc[j] += (int64_t)kb * DBK; }
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) issue((int64_t)s * DBK, s);
  for (int t = 0; t < nk; ++t) {
    const int ahead = (nk - 1 - t) < (NSTAGE - 2) ? (nk - 1 - t) : (NSTAGE - 2);     // tiles issued after tile t
    if (ahead >= 2) {
      if constexpr (NDMA == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if (ahead == 1 && NSTAGE > 2) {
      if constexpr (NDMA == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                      // tile t landed for every wave; everyone left tile t-1
#ifndef MHIMX_DBG_NODMA
    if (t + NSTAGE - 1 < nk) issue((int64_t)(t + NSTAGE - 1) * DBK, (t + NSTAGE - 1) % NSTAGE);
#endif
#ifdef MHIMX_DBG_NOCOMPUTE
    continue;
#endif
    const unsigned so = (unsigned)((t % NSTAGE) * STAGE_BYTES);
    f4 x[NRD], y[NRD];
    lds_read<NRD>(x, fa, so, 0u);                         // ks = 0
    lds_wait<NRD>(x);
    lds_read<NRD>(y, fa, so, 64u);                     // ks = 1 in flight under the first MFMA batch
    auto frags = [&](const f4 (&z)[NRD]) {
      V8 ah[2], al[2], bh[NT], bl[NT];
      if constexpr (PAIRED) {
#pragma unroll
        for (int q = 0; q < 2; ++q) { ah[q] = __builtin_bit_cast(V8, z[q * 2]); al[q] = __builtin_bit_cast(V8, z[q * 2 + 1]); }
#pragma unroll
        for (int q = 0; q < NT; ++q) { bh[q] = __builtin_bit_cast(V8, z[4 + q * 2]); bl[q] = __builtin_bit_cast(V8, z[4 + q * 2 + 1]); }
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) FR::split2(z[q * 2], z[q * 2 + 1], ah[q], al[q]);
#pragma unroll
        for (int q = 0; q < NT; ++q) FR::split2(z[4 + q * 2], z[4 + q * 2 + 1], bh[q], bl[q]);
      }
      mma_tile<PREC, NT>(ah, al, bh, bl, acc);
    };
    frags(x);
    lds_wait<NRD>(y);
    frags(y);
  }

  const int cl = lane & 31, rh = lane >> 5;
  if (ksteps > 0) {                                      // split-K: raw partial tile -> slab blockIdx.y
    float* slab = g.ws + (int64_t)blockIdx.y * g.M * g.N;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int64_t n = n0 + wn * (32 * NT) + nt * 32 + cl;
        if (n >= g.N) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int64_t m = m0 + wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * rh;
          if (m < g.M) slab[m * g.N + n] = acc[mt][nt][e];
        }
      }
    return;
  }
  // ---- epilogue (same contract as gemm.hip).  Vector form: the accumulators go through LDS (the ring is free now) and a
  // compact loop applies the element-wise work on float4 rows — one copy of the code instead of 32 unrolled ones (cold
  // I-cache lines at every launch) and 512-B row stores.  Needs 16-byte aligned rows; otherwise the scalar form below.
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (EPI_LDS_OK && n0 + DBN <= g.N && g.ldc % 4 == 0 && al16(g.C) && (!g.pre || (g.ldpre % 4 == 0 && al16(g.pre))) &&
      (!g.bias || al16(g.bias)) && (!g.colv || al16(g.colv))) {
    constexpr int TP = DBN + 8;                            // row pitch: the two half-waves of a store land 32 banks apart
    float* tile = reinterpret_cast<float*>(smem);
    __builtin_amdgcn_s_barrier();                         // every wave has left its last fragment reads
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          tile[(wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * rh) * TP + wn * (32 * NT) + nt * 32 + cl] = acc[mt][nt][e];
    __syncthreads();
    const int c4 = (tid & 31) * 4, r0 = tid >> 5;
    const int64_t n = n0 + c4;
    f4 bias = f4{0.f, 0.f, 0.f, 0.f}, colv = f4{0.f, 0.f, 0.f, 0.f};
    if (g.bias) bias = *reinterpret_cast<const f4*>(g.bias + n);
    if (g.rowv) colv = *reinterpret_cast<const f4*>(g.colv + n);
    const uint64_t dseed = g.drop_p > 0.f ? eff_seed(g.drop_seed, g.drop_tick) : 0;
    const float inv_keep = 1.f / (1.f - g.drop_p);
#pragma unroll 1
    for (int r = r0; r < DBM; r += NTHR / 32) {
      const int64_t m = m0 + r;
      if (m >= g.M) break;
      f4 v = *reinterpret_cast<const f4*>(tile + r * TP + c4) + bias;
      if (g.rowv) v += g.rowv[m] * colv;
      if (g.pre) *reinterpret_cast<f4*>(g.pre + m * g.ldpre + n) = v;
      if (g.act != MHIMX_ACT_NONE || g.drop_mask || g.drop_p > 0.f) {
        const uint32_t rkey = g.drop_p > 0.f ? drop_row_key(dseed, g.rows ? (uint64_t)g.rows[m] : (uint64_t)m) : 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float x = act_fwd(v[q], g.act);
          if (g.drop_mask) x = g.drop_mask[m * g.N + n + q] ? x * inv_keep : 0.f;
          else if (g.drop_p > 0.f) x = drop_keep_k(rkey, (uint32_t)(n + q), g.drop_p) ? x * inv_keep : 0.f;
          v[q] = x;
        }
      }
      f4* c = reinterpret_cast<f4*>(g.C + m * g.ldc + n);
      if (g.accumulate) v += *c;
      *c = v;
    }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int64_t n = n0 + wn * (32 * NT) + nt * 32 + cl;
      if (n >= g.N) continue;
      const float bias = g.bias ? g.bias[n] : 0.f;
      const float colv = g.rowv ? g.colv[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t m = m0 + wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * rh;
        if (m >= g.M) continue;
        float v = acc[mt][nt][e] + bias;
        if (g.rowv) v += g.rowv[m] * colv;
        if (g.pre) g.pre[m * g.ldpre + n] = v;
        v = act_fwd(v, g.act);
        if (g.drop_mask) {
          v = g.drop_mask[m * g.N + n] ? v / (1.f - g.drop_p) : 0.f;
        } else if (g.drop_p > 0.f) {
          const uint64_t rid = g.rows ? (uint64_t)g.rows[m] : (uint64_t)m;
          v = drop_keep(eff_seed(g.drop_seed, g.drop_tick), rid, (uint32_t)n, g.drop_p) ? v / (1.f - g.drop_p) : 0.f;
        }
        float* c = g.C + m * g.ldc + n;
        if (g.accumulate) v += *c;
        *c = v;
      }
    }
}

// =================================================================================================
// TN
// =================================================================================================
// ROWS: the riders are the row tiles of a Merge backward's rows pass (stage 4, mca2_rows.hpp; 161 KB of LDS: one workgroup per CU, the
// caller sizes the product so that tiles + riders fit the chip at once) - an instantiation of its own, so that the plain product keeps its
// register count and occupancy.
template <int PREC, int ROWS = 0>
__global__ __launch_bounds__(DTHREADS) void gemm_tn_dma_kernel(mhimx_gemm_tn_args g, int64_t mchunk, int side_blocks, int side_stage, Merge2Side side,
                                                               BagBatch bb) {
  using FR = Frag<PREC>;
  using V8 = typename FR::V8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (blockIdx.z) {
    g.A = bag_ptr(g.A, bb); g.B = bag_ptr(g.B, bb); g.C = bag_ptr(g.C, bb); g.ws = bag_ptr(g.ws, bb); g.rows = bag_ptr(g.rows, bb);
    if (side_blocks > 0) bag_move(side, bb);
  }
  if ((int)blockIdx.x < side_blocks) {          // a parked Merge-backward tail rides along (stage 2): its few short workgroups are
    if constexpr (ROWS != 0) {                   // dispatched first and free their slots early
      if ((int)blockIdx.x < side.w.T)
        merge2_rows_bwd_body<ROWS>((int)blockIdx.x, reinterpret_cast<float*>(smem), side.X, side.xrows, side.R, side.ln_w, side.ln_b, side.J, side.drop_p,
                             side.seed0, side.tick, side.dX, side.w);
    } else {
      merge2_side_stage(side_stage, (int)blockIdx.x, reinterpret_cast<float*>(smem), side);
    }
    return;
  }
  const unsigned bx = blockIdx.x - (unsigned)side_blocks;      // (side_blocks % 8 == 0: the XCD of a tile does not move)
  int64_t* rowtab = reinterpret_cast<int64_t*>(smem + TN_STAGES * STAGE_BYTES);       // [mchunk] B-row element offsets
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware order: all output tiles of one reduction slab run on one XCD (its rows of A and B are shared via L2)
  const int nJ = (int)(g.K2 / DBN), nT = (int)(g.K1 / DBM) * nJ;
  const int xcd = bx & 7, sidx = bx >> 3;
  const int zslab = (sidx / nT) * 8 + xcd, tile = sidx % nT;
  if (zslab >= g.splits && !(g.splits <= 1 && zslab == 0)) return;
  const int64_t i0 = (int64_t)(tile / nJ) * DBM, j0 = (int64_t)(tile % nJ) * DBN;
  const int64_t mbeg = (int64_t)zslab * mchunk;
  const int64_t mend = mbeg + mchunk < g.M ? mbeg + mchunk : g.M;
  const int nrows = (int)(mend > mbeg ? mend - mbeg : 0);

  for (int q = tid; q < nrows; q += DTHREADS) rowtab[q] = (g.rows ? g.rows[mbeg + q] : (mbeg + q)) * g.ldb;
  __syncthreads();

  // slot p = tid + 256 j of a [32 m][32 slots] tile
  auto issue = [&](int t, int stage) {
    char* sa = smem + stage * STAGE_BYTES + wave * 1024;
    char* sb = sa + TILE_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = tid + 256 * j;
      const int mm = t * DBK + (p >> 5), slot = p & 31;
      const bool in = mm < nrows;
      const float* a = in ? g.A + (mbeg + mm) * g.lda + i0 + slot * 4 : g_zero_row + slot * 4;
      const float* b = in ? g.B + rowtab[mm] + j0 + slot * 4 : g_zero_row + slot * 4;
      dma16(a, sa + j * 4096);
      dma16(b, sb + j * 4096);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int r = lane & 31, kh = lane >> 5;
  const int nk = (nrows + DBK - 1) / DBK;
  // TN_STAGES-deep ring: tiles t+1 .. t+TN_STAGES-1 are in flight while tile t is consumed.  Tile t must have landed for
  // EVERY wave before anyone reads it: each wave drains its own DMAs down to the younger tiles (8 instructions per tile),
  // THEN the barrier.  (Waiting after the barrier only covers a wave's own DMAs — on a cold first tile another wave's
  // rows could still be in flight: a race the 70 000-row test caught.)
  for (int t = 0; t < TN_STAGES - 1 && t < nk; ++t) issue(t, t);
  for (int t = 0; t < nk; ++t) {
    const int younger = (nk - 1 - t) < (TN_STAGES - 2) ? (nk - 1 - t) : (TN_STAGES - 2);
    if (younger >= 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // tile t landed everywhere; everyone has left tile t-1's stage
    if (t + TN_STAGES - 1 < nk) issue(t + TN_STAGES - 1, (t + TN_STAGES - 1) % TN_STAGES);
    asm volatile("" ::: "memory");
    const float* sa = reinterpret_cast<const float*>(smem + (t % TN_STAGES) * STAGE_BYTES);
    const float* sb = sa + TILE_BYTES / 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      V8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int ca = wm * 64 + q * 32 + r, cb = wn * 64 + q * 32 + r;
        const int mrow = ks * 16 + kh * 8;
        float xa[8], xb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          xa[u] = sa[(mrow + u) * 128 + ca];
          xb[u] = sb[(mrow + u) * 128 + cb];
        }
        FR::split(xa, ah[q], al[q]);
        FR::split(xb, bh[q], bl[q]);
      }
      mma12<PREC>(ah, al, bh, bl, acc);
    }
    asm volatile("" ::: "memory");
  }

  float* out = g.splits > 1 ? g.ws + (int64_t)zslab * g.K1 * g.K2 : g.C;
  const int64_t ldo = g.splits > 1 ? g.K2 : g.ldc;
  const bool accum = g.splits > 1 ? false : (g.accumulate != 0);
  const int cl = lane & 31, rh = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int64_t j = j0 + wn * 64 + nt * 32 + cl;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t i = i0 + wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * rh;
        float v = acc[mt][nt][e];
        float* p = out + i * ldo + j;
        if (accum) v += *p;
        *p = v;
      }
    }
}


// ---- host ------------------------------------------------------------------------------------------
// x[M,K] fp32 (row pitch ldx) -> out[M,K] "floats": per 8 consecutive k, 8 bf16 hi then 8 bf16 lo (32 B in, 32 B out)
__global__ void pair_planes_kernel(const float* __restrict__ x, int64_t ldx, int64_t M, int64_t K8, float* __restrict__ out) {
  const int64_t n = M * K8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / K8, g8 = i % K8;
    const f4 a = *reinterpret_cast<const f4*>(x + m * ldx + g8 * 8);
    const f4 b = *reinterpret_cast<const f4*>(x + m * ldx + g8 * 8 + 4);
    b8 hi, lo;
    Frag<MHIMX_PREC_BF16X3>::split2(a, b, hi, lo);
    f4* o = reinterpret_cast<f4*>(out + (m * K8 + g8) * 8);
    o[0] = __builtin_bit_cast(f4, hi);
    o[1] = __builtin_bit_cast(f4, lo);
  }
}
int pair_planes(hipStream_t st, const float* x, int64_t ldx, int64_t M, int64_t K, float* out) {
  MHIMX_CHECK_ARG(x && out && M > 0 && K > 0 && K % 8 == 0 && ldx % 4 == 0 && aligned16(x) && aligned16(out),
                  "pair_planes: K must be a multiple of 8, rows 16-byte aligned");
  const int64_t n = M * (K / 8);
  int64_t blocks = cdiv(n, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pair_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, ldx, M, K / 8, out);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

// One launch for the parameter-only preparation of a train step (a block range per job).  kind 0: out[c,r] = in[r,c];
// kind 1: paired planes of in[R,C]; kind 2: copy R*C floats; kind 3: *(uint64*)out += 1 (device step counters);
// kind 4: MFMA B-fragment image of in[R,C] (scorer_fused.hip); kind 5: the same image of in^T; kind 7: paired planes of in^T.
// (kind 6, several per launch - the bags of an accumulation window share the parameters but each owns a Merge workspace whose head
// holds the query-side images: job q's workspace starts m2_shift[q.R] floats behind the first one's; the layout is the same)
__global__ __launch_bounds__(256) void prep_batch_kernel(PrepJobs pj) {
  __shared__ __attribute__((aligned(16))) float lds[PREP_LDS_FLOATS];
  prep_job_block(pj, (int)blockIdx.x, lds);                    // (prep_jobs.hpp)
}
int merge2_prep_args(const mhimx_merge* m, int64_t R, void* ws, int64_t ws_bytes, Merge2PrepArgs* out);      // mca2.hip
// the job table of a launch (validated, block ranges assigned); returns the number of 256-thread blocks, < 0 on error
int prep_jobs_fill(const mhimx_prep_job* jobs, int n, PrepJobs* out) {
  MHIMX_CHECK_ARG(jobs && n >= 1 && n <= MHIMX_PREP_MAX, "prep_batch: 1..%d jobs", MHIMX_PREP_MAX);
  PrepJobs& pj = *out;
  pj.n = n;
  int n_merge = 0;
  for (int i = 0; i < n; ++i) {
    pj.j[i] = jobs[i];
    MHIMX_CHECK_ARG(jobs[i].out && (jobs[i].kind == 3 || jobs[i].kind == 10 || jobs[i].in), "prep_batch: null pointer in job %d", i);
    MHIMX_CHECK_ARG(jobs[i].kind >= 0 && jobs[i].kind <= 10, "prep_batch: unknown job kind");
    if (jobs[i].kind == 6) {
      MHIMX_CHECK_ARG(n_merge < PREP_MERGE_MAX, "prep_batch: at most %d Merge preparation jobs per launch", PREP_MERGE_MAX);
      Merge2PrepArgs m2;
      if (int r = merge2_prep_args(reinterpret_cast<const mhimx_merge*>(jobs[i].in), jobs[i].R, jobs[i].out, jobs[i].C, &m2)) return r;
      if (n_merge == 0) pj.m2 = m2;
      MHIMX_CHECK_ARG(m2.q_param == pj.m2.q_param && m2.wq == pj.m2.wq && m2.wkv == pj.m2.wkv && m2.ln_w == pj.m2.ln_w && m2.k == pj.m2.k,
                      "prep_batch: the Merge preparation jobs of one launch share their parameters");
      MHIMX_CHECK_ARG((m2.w.gq - pj.m2.w.gq) == (m2.w.gtf_aq - pj.m2.w.gtf_aq), "prep_batch: Merge workspace layouts differ");
      pj.m2_shift[n_merge] = m2.w.gq - pj.m2.w.gq;
      pj.j[i].R = n_merge++;
    }
    MHIMX_CHECK_ARG(jobs[i].kind != 5 || (jobs[i].C % 32 == 0 && jobs[i].R % 16 == 0 && aligned16(jobs[i].out)),
                    "prep_batch: the transposed fragment image needs C % 32 == 0, R % 16 == 0 and a 16-byte aligned output");
    MHIMX_CHECK_ARG(jobs[i].kind != 4 || (jobs[i].R % 32 == 0 && jobs[i].C % 16 == 0 && aligned16(jobs[i].in) && aligned16(jobs[i].out)),
                    "prep_batch: the fragment image needs R % 32 == 0, C % 16 == 0 and 16-byte aligned buffers");
    MHIMX_CHECK_ARG(jobs[i].kind != 8 || (jobs[i].C % 32 == 0 && aligned16(jobs[i].in) && aligned16(jobs[i].out)),
                    "prep_batch: the 16-row fragment image needs C % 32 == 0 and 16-byte aligned buffers");
    MHIMX_CHECK_ARG(jobs[i].kind != 9 || (jobs[i].C % 256 == 0 && jobs[i].R >= 1 && aligned16(jobs[i].in) && aligned16(jobs[i].out)),
                    "prep_batch: the bag image needs C % 256 == 0 and 16-byte aligned buffers");
    MHIMX_CHECK_ARG(jobs[i].kind != 7 || (jobs[i].R % 8 == 0 && aligned16(jobs[i].out)), "prep_batch: pairing the transpose needs R % 8 == 0 and a 16-byte aligned output");
    MHIMX_CHECK_ARG(jobs[i].kind != 1 || (jobs[i].C % 8 == 0 && aligned16(jobs[i].in) && aligned16(jobs[i].out)),
                    "prep_batch: pairing needs C % 8 == 0 and 16-byte aligned buffers");
  }
  int first = 0;
  for (int i = 0; i < n; ++i) {
    const int64_t items = (jobs[i].kind == 3 || jobs[i].kind == 10) ? 1 : (jobs[i].kind == 0 ? cdiv(jobs[i].R, 32) * cdiv(jobs[i].C, 32) :
                          (jobs[i].kind == 8 ? cdiv(jobs[i].R, 16) * 16 * jobs[i].C / 8 : jobs[i].R * jobs[i].C / 8));
    int64_t want = jobs[i].kind == 0 ? items : cdiv(items, 256);
    if (jobs[i].kind == 6) want = 64;                  // 8 heads x 8 column blocks
    if (jobs[i].kind == 9) want = cdiv(jobs[i].R, 32) * (jobs[i].C / 256);   // one 32 KiB tile per workgroup
    if (want < 1) want = 1;
    if (want > 4096) want = 4096;                      // every job loop is grid-stride
    pj.first[i] = first;
    first += (int)want;
  }
  pj.first[n] = first;
  return first;
}
int prep_batch(hipStream_t st, const mhimx_prep_job* jobs, int n) {
  PrepJobs pj;
  const int blocks = prep_jobs_fill(jobs, n, &pj);
  if (blocks < 0) return blocks;
  hipLaunchKernelGGL(prep_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, st, pj);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

bool nt_dma_ok(const mhimx_gemm_nt_args& g) {
  return (g.prec == MHIMX_PREC_BF16X3 || g.prec == MHIMX_PREC_F16S) && g.K % DBK == 0 && g.lda % 4 == 0 && g.ldb % 4 == 0 &&
         aligned16(g.A) && aligned16(g.B) && g.M > 16;
}

#ifndef MHIMX_NT_WAVES
#define MHIMX_NT_WAVES 8
#endif
int gemm_nt_dma(hipStream_t st, const mhimx_gemm_nt_args& g) {
  constexpr int NW = MHIMX_NT_WAVES;
  dim3 grid((unsigned)(8 * cdiv(g.N, DBN) * cdiv(cdiv(g.M, DBM), 8)));
  // split-K when the launch would leave most CUs idle behind a long serial K loop
  int ksteps = 0, ksplit = 1;
  {
    const int64_t tiles = cdiv(g.N, DBN) * cdiv(g.M, DBM), nk = g.K / DBK;
    const bool plain = !g.bias && !g.rowv && !g.pre && g.act == 0 && g.drop_p == 0.f && !g.drop_mask;
    if (plain && g.ws && tiles <= 128 && nk >= 8) {
      int64_t want = 256 / tiles;
      if (want > nk / 4) want = nk / 4;
      if (want > g.ws_floats / (g.M * g.N)) want = g.ws_floats / (g.M * g.N);
      if (want >= 2) {
        ksteps = (int)cdiv(nk, want);
        ksplit = (int)cdiv(nk, ksteps);
        grid.y = (unsigned)ksplit;
      }
    }
  }
  struct Reduce {
    hipStream_t st; const mhimx_gemm_nt_args& g; int ksplit;
    int run() const {
      if (ksplit <= 1) return 0;
      return reduce_slabs_now(st, g.ws, g.C, g.M, g.N, g.ldc, ksplit, g.accumulate);
    }
  } reduce{st, g, ksplit};
    MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)gemm_nt_dma_kernel<MHIMX_PREC_BF16X3, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS_BYTES)); MHIMX_HIP(hipFuncSetAttribute((const void*)gemm_nt_dma_kernel<MHIMX_PREC_F16S, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS_BYTES)));
  if (g.paired) {
        MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)gemm_nt_dma_kernel<MHIMX_PREC_BF16X3, NW, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS_BYTES)));
    hipLaunchKernelGGL((gemm_nt_dma_kernel<MHIMX_PREC_BF16X3, NW, 1>), grid, dim3(64 * NW), NT_LDS_BYTES, st, g, ksteps);
  } else if (g.prec == MHIMX_PREC_BF16X3)
    hipLaunchKernelGGL((gemm_nt_dma_kernel<MHIMX_PREC_BF16X3, NW>), grid, dim3(64 * NW), NT_LDS_BYTES, st, g, ksteps);
  else
    hipLaunchKernelGGL((gemm_nt_dma_kernel<MHIMX_PREC_F16S, NW>), grid, dim3(64 * NW), NT_LDS_BYTES, st, g, ksteps);
  MHIMX_LAUNCH_CHECK();
  return reduce.run();
}

bool tn_dma_ok(const mhimx_gemm_tn_args& g) {
  return g.prec != MHIMX_PREC_F32 && g.K1 % DBM == 0 && g.K2 % DBN == 0 && g.lda % 4 == 0 && g.ldb % 4 == 0 && aligned16(g.A) &&
         aligned16(g.B) && g.M > 16;
}

// picks the split count itself when the caller passes splits <= 0; returns the splits used through *splits_out
// rider (optional): a stage of a Merge backward's side work that runs as the launch's first workgroups (mca2_side.hpp); *rode says whether
// it got its ride (the block count must keep the tiles' XCD mapping: a multiple of 8)
int gemm_tn_dma(hipStream_t st, const mhimx_gemm_tn_args& g0, int64_t ws_floats_avail, const Merge2Side* rider, int rider_stage, bool* rode) {
  mhimx_gemm_tn_args g = g0;
  const int64_t tiles = (g.K1 / DBM) * (g.K2 / DBN);
  int splits = g.splits > 1 ? g.splits : 1;
  // enough workgroups to fill 256 CUs twice, at least 4 k-steps each, and a row table that fits LDS
  int64_t want = cdiv(512, tiles);
  if (want > cdiv(g.M, 4 * DBK)) want = cdiv(g.M, 4 * DBK);
  if (want < 1) want = 1;
  // (a bag-batched launch, common.hpp: the bags' products TOGETHER fill the chip - one round of 256 workgroups, at least 8 slabs per bag;
  // 8 bags at the single-bag slab count were 2048 workgroups in 8 rounds, 88 us, and 56 slabs per bag for the reduction to sum)
  if (cur_batch().n > 1) {
    int64_t share = 256 / (tiles * cur_batch().n) / 8 * 8;
    if (share < 8) share = 8;
    if (want > share) want = share;
  }
  if (g.ws && want > splits) splits = (int)want;
  while (splits > 1 && (int64_t)splits * g.K1 * g.K2 > ws_floats_avail) --splits;
  while (cdiv(g.M, splits) > MAX_TN_CHUNK) {
    ++splits;
    if ((int64_t)splits * g.K1 * g.K2 > ws_floats_avail) return -2;      // caller falls back to the register-staged kernel
  }
  // stage 4: the Merge backward's rows pass rides (one 161 KB-LDS workgroup per CU, as every tile of this launch then is): the product
  // takes the CUs the row tiles leave free, all of it in ONE round - 4 output tiles x 56 slabs = 224 tiles beside 31 row tiles at c2
  // (the product runs as fast on 32 slabs as on 64: it is latency, not slab traffic)
  // rider_stage 5 (round 6, the step as a DAG): nothing rides - the rows pass is a launch of its own on another branch - but the product is
  // SIZED as if rider->w.T row tiles rode (the same slab count = the same summation order = the bits of the chain form)
  const bool size_only = rider && rider_stage == 5;
  const bool rows_ride = rider && (rider_stage == 4 || size_only) && DTHREADS == M2_THREADS;
  int rows_blocks = 0;
  if (rows_ride) {
    rows_blocks = (int)align_up(rider->w.T, 8);
    // (a bag-batched launch - common.hpp: every bag of the window brings its own product and row tiles - sizes each bag's product for its share
    // of two rounds of the chip, at least 8 slabs)
    const int nbags = cur_batch().n > 0 ? cur_batch().n : 1;
    int64_t room = (nbags > 1 ? 512 / nbags : 256) - rows_blocks;
    if (nbags > 1 && room < 8 * tiles) room = 8 * tiles;
    int cap = (int)(room / tiles) / 8 * 8;
    // (MEASURED, 8 bags: with the rows riding every workgroup of the launch takes the rows pass's LDS - one per CU, 768 workgroups, three rounds,
    // 131 us; apart, the product keeps its own slab count and the rows pass is a launch of two rounds.  MHIMX_WINDOW_ROWS_RIDE=1: together)
    static const bool batched_ride = getenv("MHIMX_WINDOW_ROWS_RIDE") != nullptr && atoi(getenv("MHIMX_WINDOW_ROWS_RIDE")) != 0;
    if (cap < 8 || rows_blocks > 128 || (nbags > 1 && !batched_ride)) rows_blocks = 0;      // no room for the product beside the rows: they do not ride
    else if (splits > cap) splits = cap;
  }
  if (size_only) { rows_blocks = 0; rider = nullptr; }
  g.splits = splits;
  const int64_t mchunk = align_up(cdiv(g.M, splits), DBK);
  size_t smem = TN_STAGES * STAGE_BYTES + (size_t)mchunk * 8;
    MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)gemm_tn_dma_kernel<MHIMX_PREC_BF16X3>, hipFuncAttributeMaxDynamicSharedMemorySize, TN_STAGES * STAGE_BYTES + (MAX_TN_CHUNK + DBK) * 8)));
  Merge2Side side = {};
  int side_blocks = 0, side_stage = 2;
  if (rode) *rode = false;
  if (rows_ride && rows_blocks > 0 && cdiv(g.M, splits) <= MAX_TN_CHUNK) {
    constexpr size_t GEMM_MAX = TN_STAGES * STAGE_BYTES + (MAX_TN_CHUNK + DBK) * 8;
    constexpr size_t ROWS_SMEM = M2_BWD_SMEM > GEMM_MAX ? M2_BWD_SMEM : GEMM_MAX;
    MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)gemm_tn_dma_kernel<MHIMX_PREC_BF16X3, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ROWS_SMEM));
                          MHIMX_HIP(hipFuncSetAttribute((const void*)gemm_tn_dma_kernel<MHIMX_PREC_BF16X3, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ROWS_SMEM)));
    const size_t rows_smem = m2_bwd_smem(rider->w.rt);
    if (smem < rows_smem) smem = rows_smem;
    side = *rider;
    if (rode) *rode = true;
    dim3 grid = bgrid((unsigned)(8 * (g.K2 / DBN) * (g.K1 / DBM) * cdiv(splits, 8) + rows_blocks));
    if (side.w.rt == 16)
      hipLaunchKernelGGL((gemm_tn_dma_kernel<MHIMX_PREC_BF16X3, 16>), grid, dim3(DTHREADS), smem, st, g, mchunk, rows_blocks, 4, side, cur_batch());
    else
      hipLaunchKernelGGL((gemm_tn_dma_kernel<MHIMX_PREC_BF16X3, 32>), grid, dim3(DTHREADS), smem, st, g, mchunk, rows_blocks, 4, side, cur_batch());
    MHIMX_LAUNCH_CHECK();
    return splits;
  }
  if (rider && rider_stage != 4 && DTHREADS == M2_THREADS && smem >= M2_SIDE_LDS * sizeof(float) && merge2_side_blocks(rider_stage, *rider) % 8 == 0) {
    side = *rider;
    side_stage = rider_stage;
    side_blocks = merge2_side_blocks(rider_stage, side);
    if (rode) *rode = true;
  } else if (g.defer && g.defer->side.pending == 2 && DTHREADS == M2_THREADS && smem >= M2_SIDE_LDS * sizeof(float) && splits > 1) {
    memcpy(&side, g.defer->side.blob, sizeof(side));           // a parked Merge-backward tail: stage 2 rides in this launch
    side_blocks = merge2_side_blocks(2, side);
    if (side_blocks % 8 == 0) g.defer->side.pending = 3;
    else side_blocks = 0;
  }
  dim3 grid = bgrid((unsigned)(8 * (g.K2 / DBN) * (g.K1 / DBM) * cdiv(splits, 8) + side_blocks));
  hipLaunchKernelGGL(gemm_tn_dma_kernel<MHIMX_PREC_BF16X3>, grid, dim3(DTHREADS), smem, st, g, mchunk, side_blocks, side_stage, side, cur_batch());
  MHIMX_LAUNCH_CHECK();
  return splits;
}

}  // namespace mhimx
