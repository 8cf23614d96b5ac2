// feat_gemm.hip — the bag projection  H[M,N] = epi( X[rows?][M,K] . W[N,K]^T )  on paired-plane operands
// (mhimx_pair_planes: per 8 consecutive k, 16 B of bf16 hi then 16 B of bf16 lo), 3-term bf16 (hi*hi + hi*lo + lo*hi).
//
// Shape of the problem on MI355X: M = 10 000 patch rows, N = 512, K = 1024 gives 5.12 M outputs for 256 CUs, i.e. 20 000
// outputs per CU.  The workgroup tile is therefore 160 x 128 (20 480 outputs): ceil(10000/160) x 4 = 252 tiles — ONE
// round over the chip with every CU loaded equally (128 x 128 tiles need 316 = 1.23 rounds, the second one 23 % full).
//
//   * 4 waves as 2 x 2, 80 x 64 outputs per wave = 5 x 4 MFMA blocks of v_mfma_f32_16x16x32_bf16 (80 is not a
//     multiple of 32, so the 16-row form is used); 60 MFMAs per 32-deep k-step per wave against 18 ds_read_b128
//     (LDS bytes per MFMA cycle 0.6 of the 128 B/clk the LDS delivers).
//   * operands go HBM/L2 -> LDS by direct DMA (global_load_lds_dwordx4, 16 B per lane), 4-stage ring of
//     [160 + 128 rows][128 B]; the bank swizzle (16-B slot ^= (row>>1)&7) is applied to the per-lane SOURCE address
//     and again on the fragment read, as in gemm_dma.hip.
//   * the two 16-B slots a lane reads ARE its (hi, lo) MFMA fragments: no conversion, no VALU work in the loop.
//   * fragments are double-buffered in registers: the ds_reads of k-step t+1 are in flight under the 60 MFMAs of
//     k-step t; one s_barrier per k-step; counted vmcnt keeps two younger tiles in flight across it.
//   * XCD-aware tile order: the four N-tiles of one M-tile run on the same XCD back to back (X rows shared through
//     that XCD's L2).
#include "common.hpp"

namespace mhimx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void* gptr_f;
typedef __attribute__((address_space(3))) void* lptr_f;

constexpr int FBM = 160, FBN = 128, FBK = 32, FTHREADS = 256;
constexpr int FA_BYTES = FBM * 128, FB_BYTES = FBN * 128, FSTAGE = FA_BYTES + FB_BYTES;      // 20 KiB + 16 KiB
constexpr int FNST = 4;                                                                          // 144 KiB ring
constexpr int NRA = 5, NRB = 4, NFR = 2 * (NRA + NRB);                                           // 18 fragment reads / k-step

// 18 fragment reads of one k-step from FOUR per-lane base addresses: within an operand the 16-row blocks are 2 KiB
// apart and the bank swizzle (row>>1)&7 does not depend on the block index, so blocks are immediate offsets.
// x[0..4] A hi, x[5..9] A lo, x[10..13] B hi, x[14..17] B lo.
#define FG_READ18(x, ah, al, bh, bl)                                                                                    \
  asm volatile("ds_read_b128 %0, %18\n\tds_read_b128 %1, %18 offset:2048\n\tds_read_b128 %2, %18 offset:4096\n\t"      \
               "ds_read_b128 %3, %18 offset:6144\n\tds_read_b128 %4, %18 offset:8192\n\t"                              \
               "ds_read_b128 %5, %19\n\tds_read_b128 %6, %19 offset:2048\n\tds_read_b128 %7, %19 offset:4096\n\t"      \
               "ds_read_b128 %8, %19 offset:6144\n\tds_read_b128 %9, %19 offset:8192\n\t"                              \
               "ds_read_b128 %10, %20\n\tds_read_b128 %11, %20 offset:2048\n\tds_read_b128 %12, %20 offset:4096\n\t"   \
               "ds_read_b128 %13, %20 offset:6144\n\t"                                                                \
               "ds_read_b128 %14, %21\n\tds_read_b128 %15, %21 offset:2048\n\tds_read_b128 %16, %21 offset:4096\n\t"   \
               "ds_read_b128 %17, %21 offset:6144"                                                                     \
               : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]),    \
                 "=&v"(x[8]), "=&v"(x[9]), "=&v"(x[10]), "=&v"(x[11]), "=&v"(x[12]), "=&v"(x[13]), "=&v"(x[14]),           \
                 "=&v"(x[15]), "=&v"(x[16]), "=&v"(x[17])                                                                \
               : "v"(ah), "v"(al), "v"(bh), "v"(bl)                                                                     \
               : "memory")
#define FG_WAIT18(x)                                                                                                   \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                  \
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),           \
                 "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]),     \
                 "+v"(x[16]), "+v"(x[17])                                                                                 \
               :                                                                                                       \
               : "memory")

// fragment order in x[]: [0..4] A hi, [5..9] A lo, [10..13] B hi, [14..17] B lo
MHIMX_DEV void fg_mma(const f32x4 (&x)[NFR], f32x4 (&acc)[NRA][NRB]) {
  // term-major: 20 independent MFMAs between two that touch the same accumulator
#pragma unroll
  for (int i = 0; i < NRA; ++i)
#pragma unroll
    for (int j = 0; j < NRB; ++j)
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, x[5 + i]), __builtin_bit_cast(bf8, x[10 + j]), acc[i][j], 0, 0, 0);
#pragma unroll
  for (int i = 0; i < NRA; ++i)
#pragma unroll
    for (int j = 0; j < NRB; ++j)
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, x[i]), __builtin_bit_cast(bf8, x[14 + j]), acc[i][j], 0, 0, 0);
#pragma unroll
  for (int i = 0; i < NRA; ++i)
#pragma unroll
    for (int j = 0; j < NRB; ++j)
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, x[i]), __builtin_bit_cast(bf8, x[10 + j]), acc[i][j], 0, 0, 0);
}

__global__ __launch_bounds__(FTHREADS) void feat_gemm_kernel(mhimx_gemm_nt_args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nN = (int)(g.N / FBN), nM = (int)((g.M + FBM - 1) / FBM);
  const int xcd = blockIdx.x & 7, sidx = blockIdx.x >> 3;
  const int m_tile = (sidx / nN) * 8 + xcd, n_tile = sidx % nN;
  if (m_tile >= nM) return;
  const int64_t m0 = (int64_t)m_tile * FBM, n0 = (int64_t)n_tile * FBN;

  // DMA sources: slot p = tid + 256 j of a [rows][8 x 16 B] tile; LDS position linear in p, SOURCE slot swizzled
  const float* asrc[5];
  const float* bsrc[4];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int p = tid + FTHREADS * j;
    const int row = p >> 3, slot = (p & 7) ^ ((row >> 1) & 7);
    int64_t m = m0 + row;
    if (m >= g.M) m = g.M - 1;                              // clamped rows feed accumulators that are never stored
    asrc[j] = g.A + (g.rows ? g.rows[m] : m) * g.lda + slot * 4;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = tid + FTHREADS * j;
    const int row = p >> 3, slot = (p & 7) ^ ((row >> 1) & 7);
    bsrc[j] = g.B + (n0 + row) * g.ldb + slot * 4;
  }
  auto issue = [&](int t) {
    char* sa = smem + (t % FNST) * FSTAGE + wave * 1024;
    char* sb = sa + FA_BYTES;
    const int64_t k0 = (int64_t)t * FBK;
#pragma unroll
    for (int j = 0; j < 5; ++j) __builtin_amdgcn_global_load_lds((gptr_f)(asrc[j] + k0), (lptr_f)(sa + j * 4096), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds((gptr_f)(bsrc[j] + k0), (lptr_f)(sb + j * 4096), 16, 0, 0);
  };

  // fragment addresses (stage 0): row r = lane & 15 of a 16-row block, k-group kg = lane >> 4 -> slots 2kg (hi), 2kg+1 (lo)
  const int r16 = lane & 15, kg = lane >> 4;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_f)smem;
  const int ra = wm * 80 + r16, rb = wn * 64 + r16;          // block i adds 16 rows: (row>>1)&7 unchanged, address + 2048
  const unsigned fa_hi = lds0 + ra * 128 + (((2 * kg) ^ ((ra >> 1) & 7)) << 4);
  const unsigned fa_lo = lds0 + ra * 128 + (((2 * kg + 1) ^ ((ra >> 1) & 7)) << 4);
  const unsigned fb_hi = lds0 + FA_BYTES + rb * 128 + (((2 * kg) ^ ((rb >> 1) & 7)) << 4);
  const unsigned fb_lo = lds0 + FA_BYTES + rb * 128 + (((2 * kg + 1) ^ ((rb >> 1) & 7)) << 4);

  f32x4 acc[NRA][NRB];
#pragma unroll
  for (int i = 0; i < NRA; ++i)
#pragma unroll
    for (int j = 0; j < NRB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (int)(g.K / FBK);
  auto read_tile = [&](f32x4 (&x)[NFR], int t) {
    const unsigned so = (unsigned)((t % FNST) * FSTAGE);
    FG_READ18(x, fa_hi + so, fa_lo + so, fb_hi + so, fb_lo + so);
  };
  // wait until at most `younger` tiles issued after the wanted one are still in flight (9 DMA instructions per tile)
  auto wait_tiles = [&](int younger) {
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  // prologue: tiles 0..2 in flight; tile 0 landed -> its fragments into buffer x
  for (int t = 0; t < 3 && t < nk; ++t) issue(t);
  f32x4 x[NFR], y[NFR];
  wait_tiles(nk - 1 < 2 ? nk - 1 : 2);
  __builtin_amdgcn_s_barrier();
  read_tile(x, 0);
  FG_WAIT18(x);

  // steady state, unrolled by two (x holds tile t, y receives tile t+1, then the roles swap); single back-edge, the
  // last one or two tiles are peeled so that the accumulators keep one register assignment through the loop
  auto step = [&](f32x4 (&cur)[NFR], f32x4 (&nxt)[NFR], int t) {      // needs t + 1 < nk
    wait_tiles((nk - 1 < t + 3 ? nk - 1 : t + 2) - (t + 1));          // tile t+1 landed (tiles younger than it may fly on)
    __builtin_amdgcn_s_barrier();
#ifndef FG_NODMA
    if (t + 3 < nk) issue(t + 3);
#endif
#ifndef FG_NOLDS
    read_tile(nxt, t + 1);                                            // in flight under the 60 MFMAs below
#endif
    __builtin_amdgcn_sched_barrier(0);
#ifndef FG_NOMMA
    fg_mma(cur, acc);
#endif
    __builtin_amdgcn_sched_barrier(0);
    FG_WAIT18(nxt);
  };
  // hot loop: every step has a tile to issue (t+3 < nk) and exactly one younger tile in flight -> no branches
  auto hot = [&](f32x4 (&cur)[NFR], f32x4 (&nxt)[NFR], int t) {
    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifndef FG_NODMA
    issue(t + 3);
#endif
#ifndef FG_NOLDS
    read_tile(nxt, t + 1);
#endif
    __builtin_amdgcn_sched_barrier(0);
#ifndef FG_NOMMA
    fg_mma(cur, acc);
#endif
    __builtin_amdgcn_sched_barrier(0);
    FG_WAIT18(nxt);
  };
  int t = 0;
  for (; t + 4 < nk; t += 2) {
    hot(x, y, t);
    hot(y, x, t + 1);
  }
  for (; t + 2 < nk; t += 2) {
    step(x, y, t);
    step(y, x, t + 1);
  }
  if (nk - t == 2) {
    step(x, y, t);
    fg_mma(y, acc);
  } else {
    fg_mma(x, acc);
  }

  // ---- epilogue.  The accumulators go through LDS (the ring is free now) so that (a) the per-element work — bias,
  // pre-activation copy, exact GELU, counter-hash dropout — is ONE compact loop instead of 80 unrolled copies (the
  // unrolled form is ~70 KB of straight-line code: every instruction line a cold I-cache miss, ~15 us per launch), and
  // (b) every global store is a 512-B row segment (32 lanes x float4).
  // C layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + e.  Tile pitch 132 floats: the four row groups
  // of one ds_write land 16 banks apart -> conflict-free.
  constexpr int TP = FBN + 4;
  float* tile = reinterpret_cast<float*>(smem);
  __builtin_amdgcn_s_barrier();                          // every wave has left its last fragment reads
  {
    const int cl = lane & 15, rq = lane >> 4;
#pragma unroll
    for (int i = 0; i < NRA; ++i)
#pragma unroll
      for (int j = 0; j < NRB; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[(wm * 80 + i * 16 + rq * 4 + e) * TP + wn * 64 + j * 16 + cl] = acc[i][j][e];
  }
  __syncthreads();
  const int c4 = (tid & 31) * 4, r0 = tid >> 5;          // this thread's 4 columns are fixed: bias / colv loaded once
  const int64_t n = n0 + c4;
  float bias[4] = {0.f, 0.f, 0.f, 0.f}, colv[4] = {0.f, 0.f, 0.f, 0.f};
  if (g.bias) { const f32x4 b = *reinterpret_cast<const f32x4*>(g.bias + n); bias[0] = b[0]; bias[1] = b[1]; bias[2] = b[2]; bias[3] = b[3]; }
  if (g.rowv) { const f32x4 b = *reinterpret_cast<const f32x4*>(g.colv + n); colv[0] = b[0]; colv[1] = b[1]; colv[2] = b[2]; colv[3] = b[3]; }
  const uint64_t dseed = g.drop_p > 0.f ? eff_seed(g.drop_seed, g.drop_tick) : 0;
  const float inv_keep = 1.f / (1.f - g.drop_p);
#pragma unroll 1
  for (int r = r0; r < FBM; r += 8) {
    const int64_t m = m0 + r;
    if (m >= g.M) break;
    const f32x4 a = *reinterpret_cast<const f32x4*>(tile + r * TP + c4);
    float v[4] = {a[0] + bias[0], a[1] + bias[1], a[2] + bias[2], a[3] + bias[3]};
    if (g.rowv) {
      const float rv = g.rowv[m];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] += rv * colv[q];
    }
    if (g.pre) *reinterpret_cast<f32x4*>(g.pre + m * g.ldpre + n) = f32x4{v[0], v[1], v[2], v[3]};
    const uint32_t rkey = g.drop_p > 0.f ? drop_row_key(dseed, g.rows ? (uint64_t)g.rows[m] : (uint64_t)m) : 0u;
    float da[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float xq = v[q];
      float gq = 0.f;
      if (g.dact) act_fwd_grad(xq, g.act, v[q], gq);          // d out / d pre for the backward (shares the erf)
      else v[q] = act_fwd(xq, g.act);
      float ks = 1.f;
      if (g.drop_mask) ks = g.drop_mask[m * g.N + n + q] ? inv_keep : 0.f;
      else if (g.drop_p > 0.f) ks = drop_keep_k(rkey, (uint32_t)(n + q), g.drop_p) ? inv_keep : 0.f;
      da[q] = gq * ks;
      v[q] *= ks;
    }
    if (g.dact) *reinterpret_cast<f32x4*>(g.dact + m * g.lddact + n) = f32x4{da[0], da[1], da[2], da[3]};
    f32x4* c = reinterpret_cast<f32x4*>(g.C + m * g.ldc + n);
    f32x4 o = f32x4{v[0], v[1], v[2], v[3]};
    if (g.accumulate) { const f32x4 old = *c; o += old; }
#ifdef FG_NOSTORE
    if (v[0] == 1.2345e-30f)
#endif
    *c = o;
  }
}

bool feat_gemm_ok(const mhimx_gemm_nt_args& g) {
  return g.paired && g.prec == MHIMX_PREC_BF16X3 && g.N % FBN == 0 && g.K % FBK == 0 && g.M >= 64 &&
         g.lda % 4 == 0 && g.ldb % 4 == 0 && aligned16(g.A) && aligned16(g.B) &&
         g.ldc % 4 == 0 && aligned16(g.C) && (!g.pre || (g.ldpre % 4 == 0 && aligned16(g.pre))) && (!g.bias || aligned16(g.bias)) &&
         (!g.colv || aligned16(g.colv)) && (!g.dact || (g.lddact % 4 == 0 && aligned16(g.dact)));
}

int feat_gemm(hipStream_t st, const mhimx_gemm_nt_args& g) {
    MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)feat_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FNST * FSTAGE)));
  const int nN = (int)(g.N / FBN), nM = (int)cdiv(g.M, FBM);
  dim3 grid((unsigned)(8 * nN * cdiv(nM, 8)));
  hipLaunchKernelGGL(feat_gemm_kernel, grid, dim3(FTHREADS), FNST * FSTAGE, st, g);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

}  // namespace mhimx
