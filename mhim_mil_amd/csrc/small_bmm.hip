// small_bmm.hip — batches of SMALL square-ish products in one launch:  C_b = ident * I + alpha * op(A_b, B_b)  (+ C_b)
//
// The Nystrom pseudo-inverse (nystrom_attention.py:12-27, six iterations of four 256 x 256 x 256 products per head, forward and
// backward: ~200 launches per TransMIL train step) ran on the generic 128 x 128-tile kernels: 8 heads x 4 tiles = 32 workgroups on 256
// CUs, 18-57 us per launch.  Here a workgroup owns a 64 x 64 tile (8 heads x 16 tiles = 128 workgroups), 4 waves x one
// v_mfma_f32_32x32x16_bf16 block each, 3-term bf16 (hi*hi + hi*lo + lo*hi, ~2^-16), K in steps of 32 with the next step's global loads
// in flight; both operand tiles sit in LDS k-contiguous ([row][32 k], pitch 36), whatever their layout in memory - the operand that
// is not k-contiguous in memory is transposed by its LDS stores - so all three modes share the fragment path.  The affine epilogue
// (ident * I + alpha * product) fuses the "a I - M" steps of the iteration.   mode 0: A[M,K] B[N,K]^T, 1: A[M,K] B[K,N], 2: A[K,M]^T B[K,N].
#include "common.hpp"

namespace mhimx {

typedef float sb_f4 __attribute__((ext_vector_type(4)));
typedef float sb_f16 __attribute__((ext_vector_type(16)));
typedef __bf16 sb_b8 __attribute__((ext_vector_type(8)));
constexpr int SB_T = 64, SB_K = 32, SB_PITCH = 36, SB_THREADS = 256;

struct SmallBmm {
  const float* A; const float* B; float* C;
  int64_t lda, ldb, ldc, sA, sB, sC;
  int M, N, K;
  float alpha, ident;
  int accumulate;
  float* C2;              // optional second output of the same product: C2 = ident2 * I + alpha2 * op(A, B)  (whole-panel kernel only)
  float alpha2, ident2;
};

MHIMX_DEV void sb_split(const sb_f4& a, const sb_f4& b, sb_b8& hi, sb_b8& lo) {
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const __bf16 h = (__bf16)v[q];
    hi[q] = h;
    lo[q] = (__bf16)(v[q] - (float)h);
  }
}

// TA / TB: the operand is stored [K, rows] in memory (rows contiguous) and is transposed on its way into LDS
template <bool TA, bool TB>
__global__ __launch_bounds__(SB_THREADS) void small_bmm_kernel(SmallBmm g) {
  __shared__ __attribute__((aligned(16))) float As[SB_T * SB_PITCH];
  __shared__ __attribute__((aligned(16))) float Bs[SB_T * SB_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * SB_T, n0 = (int64_t)blockIdx.x * SB_T;
  const float* A = g.A + (int64_t)blockIdx.z * g.sA;
  const float* B = g.B + (int64_t)blockIdx.z * g.sB;
  float* C = g.C + (int64_t)blockIdx.z * g.sC;

  sb_f4 ra[2], rb[2];
  auto load = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int f = tid + SB_THREADS * j;
      if (TA) ra[j] = *reinterpret_cast<const sb_f4*>(A + (int64_t)(k0 + (f >> 4)) * g.lda + m0 + (f & 15) * 4);
      else ra[j] = *reinterpret_cast<const sb_f4*>(A + (m0 + (f >> 3)) * g.lda + k0 + (f & 7) * 4);
      if (TB) rb[j] = *reinterpret_cast<const sb_f4*>(B + (int64_t)(k0 + (f >> 4)) * g.ldb + n0 + (f & 15) * 4);
      else rb[j] = *reinterpret_cast<const sb_f4*>(B + (n0 + (f >> 3)) * g.ldb + k0 + (f & 7) * 4);
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int f = tid + SB_THREADS * j;
      if (TA) {
#pragma unroll
        for (int c = 0; c < 4; ++c) As[((f & 15) * 4 + c) * SB_PITCH + (f >> 4)] = ra[j][c];
      } else {
        *reinterpret_cast<sb_f4*>(As + (f >> 3) * SB_PITCH + (f & 7) * 4) = ra[j];
      }
      if (TB) {
#pragma unroll
        for (int c = 0; c < 4; ++c) Bs[((f & 15) * 4 + c) * SB_PITCH + (f >> 4)] = rb[j][c];
      } else {
        *reinterpret_cast<sb_f4*>(Bs + (f >> 3) * SB_PITCH + (f & 7) * 4) = rb[j];
      }
    }
  };

  sb_f16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int r = lane & 31, kh = lane >> 5;
  const float* ap = As + (wm * 32 + r) * SB_PITCH + 8 * kh;
  const float* bp = Bs + (wn * 32 + r) * SB_PITCH + 8 * kh;
  load(0);
  for (int k0 = 0; k0 < g.K; k0 += SB_K) {
    __syncthreads();                                       // the previous step's fragment reads are over
    store();
    __syncthreads();
    if (k0 + SB_K < g.K) load(k0 + SB_K);                  // in flight under the MFMAs
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      sb_b8 ah, al, bh, bl;
      sb_split(*reinterpret_cast<const sb_f4*>(ap + 16 * s), *reinterpret_cast<const sb_f4*>(ap + 16 * s + 4), ah, al);
      sb_split(*reinterpret_cast<const sb_f4*>(bp + 16 * s), *reinterpret_cast<const sb_f4*>(bp + 16 * s + 4), bh, bl);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
  }
  const int64_t n = n0 + wn * 32 + r;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int64_t m = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
    float v = g.alpha * acc[e];
    if (m == n) v += g.ident;
    float* p = C + m * g.ldc + n;
    if (g.accumulate) v += *p;
    *p = v;
  }
}

// The same product with the WHOLE K panel of both operands in LDS (K <= 256: the 256 x 256 x 256 products of the pseudo-inverse):
// the eight k-steps of small_bmm_kernel are eight global-load -> barrier -> store -> barrier rounds; here the panel is staged behind
// ONE barrier and the 8 x 6 MFMAs per wave run with no barrier between (8.5 -> 7.4 us per 8-head product; forcing all 32 loads of a
// thread in flight before the first store, or two accumulator chains, measured slower: 8.8 us).
constexpr int SBF_KMAX = 256, SBF_PITCH = SBF_KMAX + 4;
template <bool TA, bool TB>
MHIMX_DEV void small_bmm_full_body(const SmallBmm& g, int z, float* sbf) {
  float* As = sbf;
  float* Bs = sbf + SB_T * SBF_PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * SB_T, n0 = (int64_t)blockIdx.x * SB_T;
  const float* A = g.A + (int64_t)z * g.sA;
  const float* B = g.B + (int64_t)z * g.sB;
  float* C = g.C + (int64_t)z * g.sC;
  // K == SBF_KMAX (checked by the launcher): all 32 loads of the thread are in flight together, then the LDS stores
  constexpr int NKS = SBF_KMAX / SB_K;
  sb_f4 ra[NKS][2], rb[NKS][2];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int f = tid + SB_THREADS * j, k0 = ks * SB_K;
      if (TA) ra[ks][j] = *reinterpret_cast<const sb_f4*>(A + (int64_t)(k0 + (f >> 4)) * g.lda + m0 + (f & 15) * 4);
      else ra[ks][j] = *reinterpret_cast<const sb_f4*>(A + (m0 + (f >> 3)) * g.lda + k0 + (f & 7) * 4);
      if (TB) rb[ks][j] = *reinterpret_cast<const sb_f4*>(B + (int64_t)(k0 + (f >> 4)) * g.ldb + n0 + (f & 15) * 4);
      else rb[ks][j] = *reinterpret_cast<const sb_f4*>(B + (n0 + (f >> 3)) * g.ldb + k0 + (f & 7) * 4);
    }
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int f = tid + SB_THREADS * j, k0 = ks * SB_K;
      if (TA) {
#pragma unroll
        for (int c = 0; c < 4; ++c) As[((f & 15) * 4 + c) * SBF_PITCH + k0 + (f >> 4)] = ra[ks][j][c];
      } else {
        *reinterpret_cast<sb_f4*>(As + (f >> 3) * SBF_PITCH + k0 + (f & 7) * 4) = ra[ks][j];
      }
      if (TB) {
#pragma unroll
        for (int c = 0; c < 4; ++c) Bs[((f & 15) * 4 + c) * SBF_PITCH + k0 + (f >> 4)] = rb[ks][j][c];
      } else {
        *reinterpret_cast<sb_f4*>(Bs + (f >> 3) * SBF_PITCH + k0 + (f & 7) * 4) = rb[ks][j];
      }
    }
  __syncthreads();
  sb_f16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int r = lane & 31, kh = lane >> 5;
  const float* ap = As + (wm * 32 + r) * SBF_PITCH + 8 * kh;
  const float* bp = Bs + (wn * 32 + r) * SBF_PITCH + 8 * kh;
  for (int k0 = 0; k0 < SBF_KMAX; k0 += 16) {
    sb_b8 ah, al, bh, bl;
    sb_split(*reinterpret_cast<const sb_f4*>(ap + k0), *reinterpret_cast<const sb_f4*>(ap + k0 + 4), ah, al);
    sb_split(*reinterpret_cast<const sb_f4*>(bp + k0), *reinterpret_cast<const sb_f4*>(bp + k0 + 4), bh, bl);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  }
  const int64_t n = n0 + wn * 32 + r;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int64_t m = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
    float v = g.alpha * acc[e];
    if (m == n) v += g.ident;
    float* p = C + m * g.ldc + n;
    if (g.accumulate) v += *p;
    *p = v;
    if (g.C2) g.C2[(int64_t)z * g.sC + m * g.ldc + n] = g.alpha2 * acc[e] + (m == n ? g.ident2 : 0.f);
  }
}
template <bool TA, bool TB>
__global__ __launch_bounds__(SB_THREADS) void small_bmm_full_kernel(SmallBmm g) {
  extern __shared__ __attribute__((aligned(16))) float sbf[];
  small_bmm_full_body<TA, TB>(g, (int)blockIdx.z, sbf);
}
// TWO independent batches of products in one launch (same shapes, any two modes): blockIdx.z < batch -> the first.  The backward of a
// pseudo-inverse iteration is four pairs of independent 256^3 products (e.g. dzp = dz t3^T and dt3 = zp^T dz): 9 launches become 5.
MHIMX_DEV void small_bmm_full_any(const SmallBmm& g, int mode, int z, float* sbf) {
  if (mode == 0) small_bmm_full_body<false, false>(g, z, sbf);
  else if (mode == 1) small_bmm_full_body<false, true>(g, z, sbf);
  else small_bmm_full_body<true, true>(g, z, sbf);
}
__global__ __launch_bounds__(SB_THREADS) void small_bmm_pair_kernel(SmallBmm g0, int mode0, SmallBmm g1, int mode1, int batch) {
  extern __shared__ __attribute__((aligned(16))) float sbf[];
  if ((int)blockIdx.z < batch) small_bmm_full_any(g0, mode0, (int)blockIdx.z, sbf);
  else small_bmm_full_any(g1, mode1, (int)blockIdx.z - batch, sbf);
}

bool small_bmm_ok(int mode, const mhimx_gemm_nt_args& g, int batch, int64_t sA, int64_t sB, int64_t sC) {
  if (g.prec == MHIMX_PREC_F32 || g.rows || g.bias || g.M % SB_T || g.N % SB_T || g.K % SB_K) return false;
  if (g.M > 512 || g.N > 512 || g.K > 1024 || batch > 65535) return false;
  if ((g.M / SB_T) * (g.N / SB_T) * batch < 32) return false;                 // too few workgroups to be worth it
  return g.lda % 4 == 0 && g.ldb % 4 == 0 && sA % 4 == 0 && sB % 4 == 0 && aligned16(g.A) && aligned16(g.B) && mode >= 0 && mode <= 2;
}

int small_bmm2(hipStream_t st, int mode, const mhimx_gemm_nt_args& a, int batch, int64_t sA, int64_t sB, int64_t sC, float alpha, float ident,
               float* c2, float alpha2, float ident2) {
  SmallBmm g;
  g.A = a.A; g.B = a.B; g.C = a.C; g.lda = a.lda; g.ldb = a.ldb; g.ldc = a.ldc; g.sA = sA; g.sB = sB; g.sC = sC;
  g.M = (int)a.M; g.N = (int)a.N; g.K = (int)a.K; g.alpha = alpha; g.ident = ident; g.accumulate = a.accumulate;
  g.C2 = c2; g.alpha2 = alpha2; g.ident2 = ident2;
  if (c2 && g.K != SBF_KMAX) return fail(-1, "bmm_affine2: the second output needs the whole-panel kernel (K = 256)");
  dim3 grid((unsigned)(a.N / SB_T), (unsigned)(a.M / SB_T), (unsigned)batch);
  if (g.K == SBF_KMAX) {
    constexpr int SM = 2 * SB_T * SBF_PITCH * 4;
    MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)small_bmm_full_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SM));
                          MHIMX_HIP(hipFuncSetAttribute((const void*)small_bmm_full_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SM));
                          MHIMX_HIP(hipFuncSetAttribute((const void*)small_bmm_full_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SM)));
    if (mode == 0) hipLaunchKernelGGL((small_bmm_full_kernel<false, false>), grid, dim3(SB_THREADS), SM, st, g);
    else if (mode == 1) hipLaunchKernelGGL((small_bmm_full_kernel<false, true>), grid, dim3(SB_THREADS), SM, st, g);
    else hipLaunchKernelGGL((small_bmm_full_kernel<true, true>), grid, dim3(SB_THREADS), SM, st, g);
    MHIMX_LAUNCH_CHECK();
    return 0;
  }
  if (mode == 0) hipLaunchKernelGGL((small_bmm_kernel<false, false>), grid, dim3(SB_THREADS), 0, st, g);
  else if (mode == 1) hipLaunchKernelGGL((small_bmm_kernel<false, true>), grid, dim3(SB_THREADS), 0, st, g);
  else hipLaunchKernelGGL((small_bmm_kernel<true, true>), grid, dim3(SB_THREADS), 0, st, g);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

int small_bmm(hipStream_t st, int mode, const mhimx_gemm_nt_args& a, int batch, int64_t sA, int64_t sB, int64_t sC, float alpha, float ident) {
  return small_bmm2(st, mode, a, batch, sA, sB, sC, alpha, ident, nullptr, 0.f, 0.f);
}

static SmallBmm sb_args(const mhimx_gemm_nt_args& a, int64_t sA, int64_t sB, int64_t sC, float alpha, float ident) {
  SmallBmm g;
  g.A = a.A; g.B = a.B; g.C = a.C; g.lda = a.lda; g.ldb = a.ldb; g.ldc = a.ldc; g.sA = sA; g.sB = sB; g.sC = sC;
  g.M = (int)a.M; g.N = (int)a.N; g.K = (int)a.K; g.alpha = alpha; g.ident = ident; g.accumulate = a.accumulate;
  g.C2 = nullptr; g.alpha2 = g.ident2 = 0.f;
  return g;
}

}  // namespace mhimx

extern "C" int mhimx_bmm_affine_pair(void* stream, int32_t mode0, const mhimx_gemm_nt_args* a0, float alpha0, float ident0, int32_t mode1,
                                     const mhimx_gemm_nt_args* a1, float alpha1, float ident1, int32_t batch, int64_t stride) {
  using namespace mhimx;
  MHIMX_CHECK_ARG(a0 && a1 && a0->A && a0->B && a0->C && a1->A && a1->B && a1->C && batch >= 1, "bmm_affine_pair: null args");
  MHIMX_CHECK_ARG(a0->M == SBF_KMAX && a0->N == SBF_KMAX && a0->K == SBF_KMAX && a1->M == SBF_KMAX && a1->N == SBF_KMAX && a1->K == SBF_KMAX,
                  "bmm_affine_pair: 256 x 256 x 256 products only");
  MHIMX_CHECK_ARG(a0->C != a1->C, "bmm_affine_pair: the two products must write different outputs");
  MHIMX_CHECK_ARG(small_bmm_ok(mode0, *a0, 2 * batch, stride, stride, stride) && small_bmm_ok(mode1, *a1, 2 * batch, stride, stride, stride),
                  "bmm_affine_pair: 16-byte aligned contiguous operands, not the f32 mode");
  constexpr int SM = 2 * SB_T * SBF_PITCH * 4;
  MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)small_bmm_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SM)));
  hipLaunchKernelGGL(small_bmm_pair_kernel, dim3(SBF_KMAX / SB_T, SBF_KMAX / SB_T, (unsigned)(2 * batch)), dim3(SB_THREADS), SM, (hipStream_t)stream,
                     sb_args(*a0, stride, stride, stride, alpha0, ident0), (int)mode0, sb_args(*a1, stride, stride, stride, alpha1, ident1), (int)mode1,
                     (int)batch);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

extern "C" int mhimx_bmm_affine(void* stream, int32_t mode, const mhimx_gemm_nt_args* a, int32_t batch, int64_t strideA, int64_t strideB,
                                int64_t strideC, float alpha, float ident) {
  using namespace mhimx;
  MHIMX_CHECK_ARG(a && a->A && a->B && a->C && batch >= 1, "bmm_affine: null args");
  MHIMX_CHECK_ARG(ident == 0.f || a->M == a->N, "bmm_affine: ident * I needs square outputs");
  MHIMX_CHECK_ARG(small_bmm_ok(mode, *a, batch, strideA, strideB, strideC),
                  "bmm_affine: M, N multiples of 64 (<= 512), K a multiple of 32 (<= 1024), 16-byte aligned operands, not the f32 mode");
  return small_bmm((hipStream_t)stream, mode, *a, batch, strideA, strideB, strideC, alpha, ident);
}

extern "C" int mhimx_bmm_affine2(void* stream, int32_t mode, const mhimx_gemm_nt_args* a, int32_t batch, int64_t strideA, int64_t strideB,
                                 int64_t strideC, float alpha, float ident, float* C2, float alpha2, float ident2) {
  using namespace mhimx;
  MHIMX_CHECK_ARG(a && a->A && a->B && a->C && C2 && batch >= 1 && a->M == a->N, "bmm_affine2: null args / non-square output");
  MHIMX_CHECK_ARG(small_bmm_ok(mode, *a, batch, strideA, strideB, strideC),
                  "bmm_affine2: M, N multiples of 64 (<= 512), K = 256, 16-byte aligned operands, not the f32 mode");
  return small_bmm2((hipStream_t)stream, mode, *a, batch, strideA, strideB, strideC, alpha, ident, C2, alpha2, ident2);
}
