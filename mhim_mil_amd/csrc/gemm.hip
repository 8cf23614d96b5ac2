// gemm.hip — LDS-tiled MFMA GEMMs for gfx950 (wave64, v_mfma_f32_32x32x{2_f32,16_f16,16_bf16}).
//
// Two kernels share one 128x128 tile / 4-wave (2x2) compute core:
//   gemm_nt : C[M,N] = epi(A[M,K] (optionally row-gathered) . B[N,K]^T)      -- every Linear forward and dX
//   gemm_tn : C[K1,K2] = A[M,K1]^T . B[M,K2] (B optionally row-gathered)     -- every Linear weight gradient
// Operands live in HBM as fp32; they are converted on the way into LDS:
//   F32    : kept fp32, exact v_mfma_f32_32x32x2_f32 (1/16 of the 16-bit rate)
//   F16S   : A -> one fp16 term, B -> fp16 hi + lo   (2 MFMAs per tile step; the weight is the systematic error
//            source, the activation error averages out over rows: SURVEY.md §7 H1)
//   BF16X3 : A,B -> bf16 hi + lo, hi*hi + hi*lo + lo*hi (3 MFMAs; fp32 range, used for gradients)
// LDS rows are K-contiguous with a pad that makes the 16-byte fragment reads (ds_read_b128) conflict free:
// row pitch 80 B => the 16 rows of a lane group land on 16 distinct 16-B slots of the 256-B bank row.
#include "common.hpp"

namespace mhimx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, NTHREADS = 256;

// the DMA-staged fast core (gemm_dma.hip)
bool nt_dma_ok(const mhimx_gemm_nt_args& g);
int gemm_nt_dma(hipStream_t st, const mhimx_gemm_nt_args& g);
bool tn_dma_ok(const mhimx_gemm_tn_args& g);
struct Merge2Side;
int gemm_tn_dma(hipStream_t st, const mhimx_gemm_tn_args& g, int64_t ws_floats_avail, const Merge2Side* rider = nullptr, int rider_stage = 0,
                bool* rode = nullptr);
int pair_planes(hipStream_t st, const float* x, int64_t ldx, int64_t M, int64_t K, float* out);
int prep_batch(hipStream_t st, const mhimx_prep_job* jobs, int n);
bool feat_gemm_ok(const mhimx_gemm_nt_args& g);
int feat_gemm(hipStream_t st, const mhimx_gemm_nt_args& g);

// A launch may cover `batch` independent GEMMs (the heads of an attention product): blockIdx.z = b * splits + split,
// operand b lives at base + b * stride.  {0,0,0,splits} = a single GEMM.
struct Batch { int64_t sA, sB, sC; int splits; };

template <int PREC> struct Prec;
template <> struct Prec<MHIMX_PREC_F32> {
  using T = float;
  static constexpr int NA = 1, NB = 1, BK = 16, PITCH = 17;      // pitch in elements
};
template <> struct Prec<MHIMX_PREC_F16S> {
  using T = _Float16;
  static constexpr int NA = 1, NB = 2, BK = 32, PITCH = 40;
};
template <> struct Prec<MHIMX_PREC_BF16X3> {
  using T = __bf16;
  static constexpr int NA = 2, NB = 2, BK = 32, PITCH = 40;
};

// ---- fp32 -> (hi, lo) conversion -------------------------------------------------------------
template <typename T> MHIMX_DEV T cvt(float x);
template <> MHIMX_DEV float cvt<float>(float x) { return x; }
template <> MHIMX_DEV _Float16 cvt<_Float16>(float x) { return (_Float16)x; }
template <> MHIMX_DEV __bf16 cvt<__bf16>(float x) { return (__bf16)x; }

template <typename T, int NTERM>
MHIMX_DEV void split(float x, T& hi, T& lo) {
  hi = cvt<T>(x);
  if constexpr (NTERM == 2) lo = cvt<T>(x - (float)hi);
}

// ---- one k-step of MFMAs on the wave's 64x64 sub-tile ------------------------------------------
// As/Bs: [term][128][PITCH]; wave covers rows wm*64.. of As and rows wn*64.. of Bs.
template <int PREC>
MHIMX_DEV void mma_step(const typename Prec<PREC>::T* As, const typename Prec<PREC>::T* Bs, int wm, int wn, int lane,
                        f32x16 (&acc)[2][2]) {
  using PP = Prec<PREC>;
  using T = typename PP::T;
  constexpr int PITCH = PP::PITCH;
  constexpr int TSZ = 128 * PITCH;
  const int r = lane & 31, kh = lane >> 5;
  if constexpr (PREC == MHIMX_PREC_F32) {
#pragma unroll
    for (int ks = 0; ks < PP::BK / 2; ++ks) {
      float a[2], b[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = As[(wm * 64 + t * 32 + r) * PITCH + ks * 2 + kh];
        b[t] = Bs[(wn * 64 + t * 32 + r) * PITCH + ks * 2 + kh];
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
    }
  } else {
    using V8 = typename std::conditional<PREC == MHIMX_PREC_F16S, h8, b8>::type;
#pragma unroll
    for (int ks = 0; ks < PP::BK / 16; ++ks) {
      V8 a[PP::NA][2], b[PP::NB][2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int q = 0; q < PP::NA; ++q)
          a[q][t] = *reinterpret_cast<const V8*>(As + q * TSZ + (wm * 64 + t * 32 + r) * PITCH + ks * 16 + kh * 8);
#pragma unroll
        for (int q = 0; q < PP::NB; ++q)
          b[q][t] = *reinterpret_cast<const V8*>(Bs + q * TSZ + (wn * 64 + t * 32 + r) * PITCH + ks * 16 + kh * 8);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          if constexpr (PREC == MHIMX_PREC_F16S) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][mt], b[1][nt], acc[mt][nt], 0, 0, 0);   // lo first
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][mt], b[0][nt], acc[mt][nt], 0, 0, 0);
          } else {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][mt], b[0][nt], acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mt], b[1][nt], acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mt], b[0][nt], acc[mt][nt], 0, 0, 0);
          }
        }
    }
  }
}

// ---- write 4 consecutive-k fp32 values of one tile row into LDS (K-contiguous rows) --------------
template <int PREC, int NTERM>
MHIMX_DEV void lds_put4(typename Prec<PREC>::T* S, int row, int k, float4 v) {
  using PP = Prec<PREC>;
  using T = typename PP::T;
  constexpr int PITCH = PP::PITCH;
  constexpr int TSZ = 128 * PITCH;
  if constexpr (PREC == MHIMX_PREC_F32) {
    float* p = S + row * PITCH + k;
    p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
  } else {
    using V4 = typename std::conditional<PREC == MHIMX_PREC_F16S, h4, b4>::type;
    T hi[4], lo[4];
    split<T, NTERM>(v.x, hi[0], lo[0]); split<T, NTERM>(v.y, hi[1], lo[1]);
    split<T, NTERM>(v.z, hi[2], lo[2]); split<T, NTERM>(v.w, hi[3], lo[3]);
    V4 h = {hi[0], hi[1], hi[2], hi[3]};
    *reinterpret_cast<V4*>(S + row * PITCH + k) = h;
    if constexpr (NTERM == 2) {
      V4 l = {lo[0], lo[1], lo[2], lo[3]};
      *reinterpret_cast<V4*>(S + TSZ + row * PITCH + k) = l;
    }
  }
}

// =================================================================================================
// gemm_nt
// =================================================================================================
template <int PREC>
__global__ __launch_bounds__(NTHREADS) void gemm_nt_kernel(mhimx_gemm_nt_args g, Batch bt) {
  using PP = Prec<PREC>;
  g.A += blockIdx.z * bt.sA; g.B += blockIdx.z * bt.sB; g.C += blockIdx.z * bt.sC;
  using T = typename PP::T;
  constexpr int BK = PP::BK, PITCH = PP::PITCH, TSZ = 128 * PITCH;
  constexpr int F4R = BK / 4;                      // float4 per tile row
  constexpr int ITEMS = BM * F4R / NTHREADS;       // float4 per thread per operand
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* As = reinterpret_cast<T*>(smem_raw);
  T* Bs = As + PP::NA * TSZ;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;

  const float* ap[ITEMS];
  const float* bp[ITEMS];
  int rr[ITEMS], kk[ITEMS];
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    const int idx = tid + j * NTHREADS;
    rr[j] = idx / F4R;
    kk[j] = (idx % F4R) * 4;
    const int64_t m = m0 + rr[j], n = n0 + rr[j];
    ap[j] = nullptr;
    bp[j] = nullptr;
    if (m < g.M) ap[j] = g.A + (g.rows ? g.rows[m] : m) * g.lda + kk[j];
    if (n < g.N) bp[j] = g.B + n * g.ldb + kk[j];
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float4 ra[ITEMS], rb[ITEMS];
  auto fetch = [&](int64_t k0) {
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const bool kin = (k0 + kk[j]) < g.K;
      ra[j] = (ap[j] && kin) ? *reinterpret_cast<const float4*>(ap[j] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
      rb[j] = (bp[j] && kin) ? *reinterpret_cast<const float4*>(bp[j] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  fetch(0);
  for (int64_t k0 = 0; k0 < g.K; k0 += BK) {
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      lds_put4<PREC, PP::NA>(As, rr[j], kk[j], ra[j]);
      lds_put4<PREC, PP::NB>(Bs, rr[j], kk[j], rb[j]);
    }
    __syncthreads();
    if (k0 + BK < g.K) fetch(k0 + BK);
    mma_step<PREC>(As, Bs, wm, wn, lane, acc);
    __syncthreads();
  }

  // ---- epilogue
  const int cl = lane & 31, rh = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int64_t n = n0 + wn * 64 + nt * 32 + cl;
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
// skinny forms (M <= 16 rows: the k global-query tokens of Merge): exact fp32 FMA, wave per output column
// =================================================================================================
constexpr int SKINNY_M = 16;

// optional on-the-fly transform of the shared left operand of a skinny pair: a(m, col) = A[m, col] * keep(seed, m, col) / (1 - p)
// (the backward through an output dropout whose mask is the counter hash - no masked copy of A, no launch to make one), and
// colsum[col] (+)= sum_m a(m, col)
struct SkinnyADrop { float p; uint64_t seed; const uint64_t* tick; float* colsum; int accumulate; };

MHIMX_DEV float skinny_a(const SkinnyADrop& d, uint64_t seed, float v, int64_t m, int64_t col) {
  if (d.p <= 0.f) return v;
  const float ks = 1.f / (1.f - d.p);
  return drop_keep(seed, (uint64_t)m, (uint32_t)col, d.p) ? v * ks : 0.f;
}

template <bool AD>
MHIMX_DEV void skinny_nt_body(const mhimx_gemm_nt_args& g, int64_t block, const SkinnyADrop& ad) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n = block * 4 + wave;
  if (n >= g.N) return;
  float acc[SKINNY_M];
#pragma unroll
  for (int m = 0; m < SKINNY_M; ++m) acc[m] = 0.f;
  const float* brow = g.B + n * g.ldb;
  for (int64_t k = lane * 4; k < g.K; k += 256) {
    const float4 w = *reinterpret_cast<const float4*>(brow + k);
#pragma unroll
    for (int m = 0; m < SKINNY_M; ++m) {
      if (m < g.M) {
        float4 a = *reinterpret_cast<const float4*>(g.A + (g.rows ? g.rows[m] : (int64_t)m) * g.lda + k);
        if constexpr (AD) {
          const uint64_t sd = eff_seed(ad.seed, ad.tick);
          a.x = skinny_a(ad, sd, a.x, m, k); a.y = skinny_a(ad, sd, a.y, m, k + 1);
          a.z = skinny_a(ad, sd, a.z, m, k + 2); a.w = skinny_a(ad, sd, a.w, m, k + 3);
        }
        acc[m] += a.x * w.x + a.y * w.y + a.z * w.z + a.w * w.w;
      }
    }
  }
#pragma unroll
  for (int m = 0; m < SKINNY_M; ++m) acc[m] = wave_sum(acc[m]);
  if (lane != 0) return;
  const float bias = g.bias ? g.bias[n] : 0.f;
  for (int m = 0; m < g.M; ++m) {
    float v = acc[m] + bias;
    if (g.rowv) v += g.rowv[m] * g.colv[n];
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
__global__ __launch_bounds__(256) void skinny_nt_kernel(mhimx_gemm_nt_args g) { skinny_nt_body<false>(g, blockIdx.x, SkinnyADrop{}); }

template <bool AD>
MHIMX_DEV void skinny_tn_body(const mhimx_gemm_tn_args& g, int64_t i, int64_t jblock, const SkinnyADrop& ad) {
  const int64_t j = jblock * 256 + threadIdx.x;
  if (j >= g.K2) return;
  float acc = 0.f, cs = 0.f;
  uint64_t sd = 0;
  if constexpr (AD) sd = eff_seed(ad.seed, ad.tick);
  for (int64_t m = 0; m < g.M; ++m) {
    float a = g.A[m * g.lda + i];
    if constexpr (AD) a = skinny_a(ad, sd, a, m, i);
    cs += a;
    acc += a * g.B[(g.rows ? g.rows[m] : m) * g.ldb + j];
  }
  float* p = g.C + i * g.ldc + j;
  *p = g.accumulate ? *p + acc : acc;
  if constexpr (AD)
    if (ad.colsum && j == 0) ad.colsum[i] = ad.accumulate ? ad.colsum[i] + cs : cs;
}
__global__ __launch_bounds__(256) void skinny_tn_kernel(mhimx_gemm_tn_args g) { skinny_tn_body<false>(g, blockIdx.y, blockIdx.x, SkinnyADrop{}); }

// Two independent skinny products that read the same few rows (the Merge backward has two such pairs: d_W = d^T x and
// d_in = d W^T) as ONE launch: blocks [0, nt_blocks) run the NT form, the rest the TN form.  Each tiny kernel on the step's
// serial chain costs a ~5 us launch floor.
template <bool AD>
__global__ __launch_bounds__(256) void skinny_pair_kernel(mhimx_gemm_tn_args t, mhimx_gemm_nt_args g, int nt_blocks, int tn_jblocks,
                                                          SkinnyADrop ad) {
  if ((int)blockIdx.x < nt_blocks) {
    skinny_nt_body<AD>(g, blockIdx.x, ad);
  } else {
    const int b = (int)blockIdx.x - nt_blocks;
    skinny_tn_body<AD>(t, b / tn_jblocks, b % tn_jblocks, ad);
  }
}

template <int PREC>
static int launch_nt(hipStream_t st, const mhimx_gemm_nt_args& g, int batch = 1, Batch bt = Batch{0, 0, 0, 1}) {
  using PP = Prec<PREC>;
  const size_t smem = (size_t)(PP::NA + PP::NB) * 128 * PP::PITCH * sizeof(typename PP::T);
  dim3 grid((unsigned)cdiv(g.N, BN), (unsigned)cdiv(g.M, BM), (unsigned)batch);
  hipLaunchKernelGGL(gemm_nt_kernel<PREC>, grid, dim3(NTHREADS), smem, st, g, bt);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

// a_drop_*: the shared left operand is dropout(A) with the counter-hash mask (p, seed, tick); a_colsum (+)= its column sums
int skinny_pair(hipStream_t st, const mhimx_gemm_tn_args& t, const mhimx_gemm_nt_args& g, float a_drop_p, uint64_t a_seed,
                const uint64_t* a_tick, float* a_colsum, int a_accumulate, int use_a_drop) {
  MHIMX_CHECK_ARG(t.M > 0 && t.M <= SKINNY_M && g.M > 0 && g.M <= SKINNY_M && t.A && t.B && t.C && g.A && g.B && g.C,
                  "skinny_pair: both products need 1..%d rows and non-null operands", SKINNY_M);
  MHIMX_CHECK_ARG(g.K % 4 == 0 && g.lda % 4 == 0 && g.ldb % 4 == 0 && aligned16(g.A) && aligned16(g.B), "skinny_pair: NT operands must be 16-byte aligned rows");
  const int nt_blocks = (int)cdiv(g.N, 4), tn_jblocks = (int)cdiv(t.K2, 256);
  MHIMX_CHECK_ARG(!use_a_drop || (t.A == g.A && t.lda == g.lda && !g.rows && a_drop_p >= 0.f && a_drop_p < 1.f),
                  "skinny_pair: the operand transform needs the two products to share A");
  const SkinnyADrop ad{a_drop_p, a_seed, a_tick, a_colsum, a_accumulate};
  if (use_a_drop)
    hipLaunchKernelGGL(skinny_pair_kernel<true>, dim3((unsigned)(nt_blocks + tn_jblocks * t.K1)), dim3(256), 0, st, t, g, nt_blocks, tn_jblocks, ad);
  else
    hipLaunchKernelGGL(skinny_pair_kernel<false>, dim3((unsigned)(nt_blocks + tn_jblocks * t.K1)), dim3(256), 0, st, t, g, nt_blocks, tn_jblocks, ad);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

int gemm_nt(hipStream_t st, const mhimx_gemm_nt_args& g) {
  MHIMX_CHECK_ARG(g.M >= 0 && g.N > 0 && g.K > 0, "gemm_nt: bad dims M=%lld N=%lld K=%lld", (long long)g.M, (long long)g.N, (long long)g.K);
  if (g.M == 0) return 0;
  MHIMX_CHECK_ARG(g.K % 4 == 0 && g.lda % 4 == 0 && g.ldb % 4 == 0, "gemm_nt: K, lda, ldb must be multiples of 4");
  MHIMX_CHECK_ARG(aligned16(g.A) && aligned16(g.B), "gemm_nt: A and B must be 16-byte aligned");
  MHIMX_CHECK_ARG(g.A && g.B && g.C, "gemm_nt: null operand");
  MHIMX_CHECK_ARG(!g.rowv || g.colv, "gemm_nt: rowv needs colv");
  MHIMX_CHECK_ARG(g.drop_p >= 0.f && g.drop_p < 1.f, "gemm_nt: drop_p out of range");
  MHIMX_CHECK_ARG(!g.dact || (g.paired && feat_gemm_ok(g)), "gemm_nt: the dact output exists only on the paired-plane projection kernel");
  if (g.paired) {
    MHIMX_CHECK_ARG(g.M > SKINNY_M && g.K % 32 == 0 && nt_dma_ok(g), "gemm_nt: paired-plane operands need M > 16, K % 32 == 0, aligned rows");
    if (feat_gemm_ok(g)) return feat_gemm(st, g);          // 160 x 128 tiles: one balanced round over the chip
    return gemm_nt_dma(st, g);
  }
  if (g.M <= SKINNY_M) {
    hipLaunchKernelGGL(skinny_nt_kernel, dim3((unsigned)cdiv(g.N, 4)), dim3(256), 0, st, g);
    MHIMX_LAUNCH_CHECK();
    return 0;
  }
  if (nt_dma_ok(g)) return gemm_nt_dma(st, g);
  switch (g.prec) {
    case MHIMX_PREC_F32: return launch_nt<MHIMX_PREC_F32>(st, g);
    case MHIMX_PREC_F16S: return launch_nt<MHIMX_PREC_F16S>(st, g);
    case MHIMX_PREC_BF16X3: return launch_nt<MHIMX_PREC_BF16X3>(st, g);
    default: return fail(-1, "gemm_nt: unknown prec %d", g.prec);
  }
}

// =================================================================================================
// gemm_tn : C[i,j] = sum_m A[m,i] * B[rows[m], j]
// =================================================================================================
template <int PREC>
__global__ __launch_bounds__(NTHREADS) void gemm_tn_kernel(mhimx_gemm_tn_args g, int64_t mchunk, Batch bt) {
  using PP = Prec<PREC>;
  const int zb = blockIdx.z / bt.splits, zs = blockIdx.z % bt.splits;
  g.A += zb * bt.sA; g.B += zb * bt.sB; g.C += zb * bt.sC;
  using T = typename PP::T;
  constexpr int BK = PP::BK, PITCH = PP::PITCH, TSZ = 128 * PITCH;
  constexpr int HK = BK / 2;                         // m-values per thread per operand
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* As = reinterpret_cast<T*>(smem_raw);
  T* Bs = As + PP::NA * TSZ;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t i0 = (int64_t)blockIdx.y * BM, j0 = (int64_t)blockIdx.x * BN;
  const int64_t mbeg = (int64_t)zs * mchunk;
  const int64_t mend = mbeg + mchunk < g.M ? mbeg + mchunk : g.M;

  const int c = tid & 127, half = tid >> 7;          // this thread stages column c, m-range half
  const bool ain = (i0 + c) < g.K1, bin = (j0 + c) < g.K2;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float ra[HK], rb[HK];
  auto fetch = [&](int64_t m0) {
#pragma unroll
    for (int q = 0; q < HK; ++q) {
      const int64_t m = m0 + half * HK + q;
      const bool in = m < mend;
      ra[q] = (in && ain) ? g.A[m * g.lda + i0 + c] : 0.f;
      rb[q] = (in && bin) ? g.B[(g.rows ? g.rows[m] : m) * g.ldb + j0 + c] : 0.f;
    }
  };
  auto put = [&]() {
#pragma unroll
    for (int q = 0; q < HK; q += 4) {
      lds_put4<PREC, PP::NA>(As, c, half * HK + q, make_float4(ra[q], ra[q + 1], ra[q + 2], ra[q + 3]));
      lds_put4<PREC, PP::NB>(Bs, c, half * HK + q, make_float4(rb[q], rb[q + 1], rb[q + 2], rb[q + 3]));
    }
  };

  if (mbeg < mend) {
    fetch(mbeg);
    for (int64_t m0 = mbeg; m0 < mend; m0 += BK) {
      put();
      __syncthreads();
      if (m0 + BK < mend) fetch(m0 + BK);
      mma_step<PREC>(As, Bs, wm, wn, lane, acc);
      __syncthreads();
    }
  }

  float* out = g.splits > 1 ? g.ws + (int64_t)blockIdx.z * g.K1 * g.K2 : g.C;      // slab index = b * splits + split
  const int64_t ldo = g.splits > 1 ? g.K2 : g.ldc;
  const bool accum = g.splits > 1 ? false : (g.accumulate != 0);
  const int cl = lane & 31, rh = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int64_t j = j0 + wn * 64 + nt * 32 + cl;
      if (j >= g.K2) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t i = i0 + wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * rh;
        if (i >= g.K1) continue;
        float v = acc[mt][nt][e];
        float* p = out + i * ldo + j;
        if (accum) v += *p;
        *p = v;
      }
    }
}

__global__ void reduce_slabs_kernel(const float* __restrict__ ws, float* __restrict__ C, int64_t K1, int64_t K2,
                                    int64_t ldc, int splits, int accumulate, int64_t sC = 0) {
  const int64_t n = K1 * K2;
  ws += (int64_t)blockIdx.y * splits * n;          // batch element blockIdx.y
  C += (int64_t)blockIdx.y * sC;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
    // four independent partial sums keep four slab loads in flight (the slabs are n floats apart: every load is its own
    // cache line); the final combination order is fixed => deterministic
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = 0;
    for (; z + 4 <= splits; z += 4) {
      s0 += ws[(int64_t)(z + 0) * n + idx];
      s1 += ws[(int64_t)(z + 1) * n + idx];
      s2 += ws[(int64_t)(z + 2) * n + idx];
      s3 += ws[(int64_t)(z + 3) * n + idx];
    }
    for (; z < splits; ++z) s0 += ws[(int64_t)z * n + idx];
    const float s = (s0 + s1) + (s2 + s3);
    float* p = C + (idx / K2) * ldc + (idx % K2);
    *p = accumulate ? *p + s : s;
  }
}

int reduce_slabs_now(hipStream_t st, const float* ws, float* C, int64_t K1, int64_t K2, int64_t ldc, int splits, int accumulate, int batch,
                     int64_t sC) {
  const int64_t n = K1 * K2;
  MHIMX_CHECK_ARG(cur_batch().n == 0, "reduce: a bag-batched launch queues its slab sums (mhimx_reduce_list full?)");
  if (batch == 1 && reduce_slabs_as_parts(splits, K1, K2, ldc)) return reduce_parts_now(st, ws, splits, n, n, C, accumulate);
  const int blocks = (int)(cdiv(n, 256) < 2048 ? cdiv(n, 256) : 2048);
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks, batch), dim3(256), 0, st, ws, C, K1, K2, ldc, splits, accumulate, sC);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

template <int PREC>
static int launch_tn(hipStream_t st, const mhimx_gemm_tn_args& g, int batch = 1, Batch bt = Batch{0, 0, 0, 1}) {
  using PP = Prec<PREC>;
  const size_t smem = (size_t)(PP::NA + PP::NB) * 128 * PP::PITCH * sizeof(typename PP::T);
  const int splits = g.splits > 1 ? g.splits : 1;
  bt.splits = splits;
  const int64_t mchunk = align_up(cdiv(g.M, splits), PP::BK);
  dim3 grid((unsigned)cdiv(g.K2, BN), (unsigned)cdiv(g.K1, BM), (unsigned)(splits * batch));
  hipLaunchKernelGGL(gemm_tn_kernel<PREC>, grid, dim3(NTHREADS), smem, st, g, mchunk, bt);
  MHIMX_LAUNCH_CHECK();
  if (splits > 1) return reduce_slabs_now(st, g.ws, g.C, g.K1, g.K2, g.ldc, splits, g.accumulate, batch, bt.sC);
  return 0;
}

static int gemm_tn_impl(hipStream_t st, const mhimx_gemm_tn_args& g, const Merge2Side* rider, int rider_stage, bool* rode);
int gemm_tn(hipStream_t st, const mhimx_gemm_tn_args& g) { return gemm_tn_impl(st, g, nullptr, 0, nullptr); }
// the product with a stage of a Merge backward's side work riding along; returns 1 if it rode, 0 if the product ran without it, < 0 on error
int gemm_tn_rider(hipStream_t st, const mhimx_gemm_tn_args& g, const Merge2Side* rider, int stage) {
  bool rode = false;
  const int rc = gemm_tn_impl(st, g, rider, stage, &rode);
  return rc < 0 ? rc : (rode ? 1 : 0);
}
static int gemm_tn_impl(hipStream_t st, const mhimx_gemm_tn_args& g, const Merge2Side* rider, int rider_stage, bool* rode) {
  if (rode) *rode = false;
  MHIMX_CHECK_ARG(g.M >= 0 && g.K1 > 0 && g.K2 > 0, "gemm_tn: bad dims");
  MHIMX_CHECK_ARG(g.A && g.B && g.C, "gemm_tn: null operand");
  MHIMX_CHECK_ARG(g.splits <= 1 || g.ws, "gemm_tn: splits>1 needs ws");
  MHIMX_CHECK_ARG(cur_batch().n == 0 || (g.M > SKINNY_M && tn_dma_ok(g) && g.defer), "gemm_tn: a bag-batched launch takes the LDS-DMA product with its slab sum queued");
  if (g.M <= SKINNY_M) {
    hipLaunchKernelGGL(skinny_tn_kernel, dim3((unsigned)cdiv(g.K2, 256), (unsigned)g.K1), dim3(256), 0, st, g);
    MHIMX_LAUNCH_CHECK();
    return 0;
  }
  if (tn_dma_ok(g)) {
    int64_t avail = g.ws ? (g.ws_floats > 0 ? g.ws_floats : (int64_t)(g.splits > 1 ? g.splits : 0) * g.K1 * g.K2) : 0;
    const int used = gemm_tn_dma(st, g, avail, rider, rider_stage, rode);
    if (used == -2) goto generic;         // reduction too long for the LDS row table with this much workspace
    if (used < 0) return used;
    if (used > 1 && !defer_push(g.defer, reduce_job_slabs(g.ws, used, g.K1, g.K2, g.ldc, g.C, g.accumulate)))
      return reduce_slabs_now(st, g.ws, g.C, g.K1, g.K2, g.ldc, used, g.accumulate);
    return 0;
  }
generic:
  switch (g.prec) {
    case MHIMX_PREC_F32: return launch_tn<MHIMX_PREC_F32>(st, g);
    case MHIMX_PREC_F16S:      // fp16 has no headroom for gradients: use the bf16 3-term form
    case MHIMX_PREC_BF16X3: return launch_tn<MHIMX_PREC_BF16X3>(st, g);
    default: return fail(-1, "gemm_tn: unknown prec %d", g.prec);
  }
}

// =================================================================================================
// gemm_nn : C[M,N] = A[M,K] . B[K,N]   (B row-major [K,N]: the k-strided operand is staged like gemm_tn's)
// Used where both operands are activations (attention products, pseudo-inverse iterations, dX = dY . W).
// alpha scales the product; accumulate adds into C.  K % 4 == 0, lda % 4 == 0.
// =================================================================================================
template <int PREC>
__global__ __launch_bounds__(NTHREADS) void gemm_nn_kernel(mhimx_gemm_nt_args g, float alpha, int64_t kchunk, float* slabs, Batch bt) {
  using PP = Prec<PREC>;
  const int zb = blockIdx.z / bt.splits, zs = blockIdx.z % bt.splits;
  g.A += zb * bt.sA; g.B += zb * bt.sB; g.C += zb * bt.sC;
  using T = typename PP::T;
  constexpr int BK = PP::BK, PITCH = PP::PITCH, TSZ = 128 * PITCH;
  constexpr int F4R = BK / 4, ITEMS = BM * F4R / NTHREADS, HK = BK / 2;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* As = reinterpret_cast<T*>(smem_raw);
  T* Bs = As + PP::NA * TSZ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  const float* ap[ITEMS];
  int rr[ITEMS], kk[ITEMS];
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    const int idx = tid + j * NTHREADS;
    rr[j] = idx / F4R;
    kk[j] = (idx % F4R) * 4;
    const int64_t m = m0 + rr[j];
    ap[j] = m < g.M ? g.A + (g.rows ? g.rows[m] : m) * g.lda + kk[j] : nullptr;
  }
  const int c = tid & 127, half = tid >> 7;
  const bool bin = (n0 + c) < g.N;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float4 ra[ITEMS];
  float rb[HK];
  const int64_t kbeg = (int64_t)zs * kchunk;
  const int64_t kend = kbeg + kchunk < g.K ? kbeg + kchunk : g.K;
  auto fetch = [&](int64_t k0) {
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
      ra[j] = (ap[j] && (k0 + kk[j]) < kend) ? *reinterpret_cast<const float4*>(ap[j] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < HK; ++q) {
      const int64_t k = k0 + half * HK + q;
      rb[q] = (bin && k < kend) ? g.B[k * g.ldb + n0 + c] : 0.f;
    }
  };
  if (kbeg < kend) fetch(kbeg);
  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) lds_put4<PREC, PP::NA>(As, rr[j], kk[j], ra[j]);
#pragma unroll
    for (int q = 0; q < HK; q += 4)
      lds_put4<PREC, PP::NB>(Bs, c, half * HK + q, make_float4(rb[q], rb[q + 1], rb[q + 2], rb[q + 3]));
    __syncthreads();
    if (k0 + BK < kend) fetch(k0 + BK);
    mma_step<PREC>(As, Bs, wm, wn, lane, acc);
    __syncthreads();
  }
  float* outp = slabs ? slabs + (int64_t)blockIdx.z * g.M * g.N : g.C;
  const int64_t ldo = slabs ? g.N : g.ldc;
  const int cl = lane & 31, rh = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int64_t n = n0 + wn * 64 + nt * 32 + cl;
      if (n >= g.N) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t m = m0 + wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * rh;
        if (m >= g.M) continue;
        float v = acc[mt][nt][e] * alpha;
        float* cp = outp + m * ldo + n;
        if (g.accumulate && !slabs) v += *cp;
        *cp = v;
      }
    }
}

template <int PREC>
static int launch_nn(hipStream_t st, const mhimx_gemm_nt_args& g, float alpha, int splits, float* ws, int batch = 1,
                     Batch bt = Batch{0, 0, 0, 1}) {
  using PP = Prec<PREC>;
  const size_t smem = (size_t)(PP::NA + PP::NB) * 128 * PP::PITCH * sizeof(typename PP::T);
  if (splits < 1 || !ws) splits = 1;
  bt.splits = splits;
  const int64_t kchunk = align_up(cdiv(g.K, splits), PP::BK);
  dim3 grid((unsigned)cdiv(g.N, BN), (unsigned)cdiv(g.M, BM), (unsigned)(splits * batch));
  hipLaunchKernelGGL(gemm_nn_kernel<PREC>, grid, dim3(NTHREADS), smem, st, g, alpha, kchunk, splits > 1 ? ws : (float*)nullptr, bt);
  MHIMX_LAUNCH_CHECK();
  if (splits > 1) return reduce_slabs_now(st, ws, g.C, g.M, g.N, g.ldc, splits, g.accumulate, batch, bt.sC);
  return 0;
}

// unaligned / odd-shaped operands (K or lda not a multiple of 4): one thread per output element, fp32 FMA
__global__ void gemm_nn_scalar_kernel(mhimx_gemm_nt_args g, float alpha) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= g.N) return;
  for (int64_t m = blockIdx.y; m < g.M; m += gridDim.y) {
    float acc = 0.f;
    for (int64_t k = 0; k < g.K; ++k) acc = fmaf(g.A[m * g.lda + k], g.B[k * g.ldb + n], acc);
    float* c = g.C + m * g.ldc + n;
    *c = g.accumulate ? *c + alpha * acc : alpha * acc;
  }
}

int gemm_nn(hipStream_t st, const mhimx_gemm_nt_args& g, float alpha, int splits, float* ws) {
  MHIMX_CHECK_ARG(g.M >= 0 && g.N > 0 && g.K > 0 && g.A && g.B && g.C, "gemm_nn: bad args");
  if (g.M == 0) return 0;
  if (!(g.K % 4 == 0 && g.lda % 4 == 0 && aligned16(g.A))) {
    hipLaunchKernelGGL(gemm_nn_scalar_kernel, dim3((unsigned)cdiv(g.N, 256), (unsigned)(g.M < 65535 ? g.M : 65535)), dim3(256), 0, st, g, alpha);
    MHIMX_LAUNCH_CHECK();
    return 0;
  }
  switch (g.prec) {
    case MHIMX_PREC_F32: return launch_nn<MHIMX_PREC_F32>(st, g, alpha, splits, ws);
    case MHIMX_PREC_F16S:
    case MHIMX_PREC_BF16X3: return launch_nn<MHIMX_PREC_BF16X3>(st, g, alpha, splits, ws);
    default: return fail(-1, "gemm_nn: unknown prec %d", g.prec);
  }
}

// =================================================================================================
// batched form: `batch` GEMMs of one shape in ONE launch (attention heads, pseudo-inverse iterations).
// mode 0: C_b = A_b B_b^T (nt)   1: C_b = alpha A_b B_b (nn)   2: C_b = A_b^T B_b (tn; A_b is [K,M], B_b is [K,N])
// splits > 1 (nn / tn): split the reduction dimension, ws >= batch*splits*M*N floats.
// =================================================================================================
// C[i,j] = sum_m A[m,i] B[m,j] with a THIN left operand (K1 <= 16 columns, e.g. DSMIL's per-class attention A [M,C] against
// V [M,E]): a streaming pass over B - thread = column j, K1 accumulators, A's few values per row are wave-uniform loads - instead
// of a 128 x 128 matrix-core tile that is 98 % padding.  Row chunks -> slabs -> reduce_slabs_kernel (fixed order).
__global__ __launch_bounds__(256) void thin_tn_kernel(mhimx_gemm_tn_args t, int64_t chunk, float* __restrict__ out, int64_t out_ld,
                                                      int to_ws) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= t.K2) return;
  const int64_t m0 = (int64_t)blockIdx.y * chunk, m1 = m0 + chunk < t.M ? m0 + chunk : t.M;
  float acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = 0.f;
  const int K1 = (int)t.K1;
#pragma unroll 4
  for (int64_t m = m0; m < m1; ++m) {
    const float b = t.B[m * t.ldb + j];
    const float* ar = t.A + m * t.lda;
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c < K1) acc[c] += ar[c] * b;
  }
  float* o = to_ws ? out + (int64_t)blockIdx.y * t.K1 * t.K2 : out;
#pragma unroll
  for (int c = 0; c < 16; ++c)
    if (c < K1) {
      float* p = o + c * out_ld + j;
      *p = (!to_ws && t.accumulate) ? *p + acc[c] : acc[c];
    }
}

bool small_bmm_ok(int mode, const mhimx_gemm_nt_args& g, int batch, int64_t sA, int64_t sB, int64_t sC);                        // small_bmm.hip
int small_bmm(hipStream_t st, int mode, const mhimx_gemm_nt_args& a, int batch, int64_t sA, int64_t sB, int64_t sC, float alpha, float ident);

int gemm_batched(hipStream_t st, int mode, const mhimx_gemm_nt_args& g, int batch, int64_t sA, int64_t sB, int64_t sC, float alpha,
                 int splits, float* ws) {
  MHIMX_CHECK_ARG(batch >= 1 && batch <= 4096 && g.A && g.B && g.C && g.M > 0 && g.N > 0 && g.K > 0, "gemm_batched: bad args");
  MHIMX_CHECK_ARG(!g.rows && !g.bias && !g.rowv && !g.pre && g.act == 0 && g.drop_p == 0.f && !g.drop_mask,
                  "gemm_batched: no epilogue / gather in the batched form");
  if (splits < 1 || !ws) splits = 1;
  MHIMX_CHECK_ARG((int64_t)batch * splits <= 65535, "gemm_batched: batch*splits > 65535");
  const Batch bt{sA, sB, sC, splits};
  const bool f32 = g.prec == MHIMX_PREC_F32;
  if (splits == 1 && (mode != 0 || alpha == 1.f) && small_bmm_ok(mode, g, batch, sA, sB, sC))     // batches of small products: small_bmm.hip
    return small_bmm(st, mode, g, batch, sA, sB, sC, mode == 1 ? alpha : 1.f, 0.f);
  if (mode == 0) {
    MHIMX_CHECK_ARG(g.K % 4 == 0 && g.lda % 4 == 0 && g.ldb % 4 == 0 && sA % 4 == 0 && sB % 4 == 0 && aligned16(g.A) && aligned16(g.B),
                    "gemm_batched(nt): K, lda, ldb, strides must be multiples of 4 and A, B 16-byte aligned");
    return f32 ? launch_nt<MHIMX_PREC_F32>(st, g, batch, bt) : launch_nt<MHIMX_PREC_BF16X3>(st, g, batch, bt);
  }
  if (mode == 1) {
    MHIMX_CHECK_ARG(g.K % 4 == 0 && g.lda % 4 == 0 && sA % 4 == 0 && aligned16(g.A),
                    "gemm_batched(nn): K, lda, A stride multiples of 4 and A 16-byte aligned");
    return f32 ? launch_nn<MHIMX_PREC_F32>(st, g, alpha, splits, ws, batch, bt)
               : launch_nn<MHIMX_PREC_BF16X3>(st, g, alpha, splits, ws, batch, bt);
  }
  if (mode == 2) {
    mhimx_gemm_tn_args t{};
    t.A = g.A; t.lda = g.lda; t.B = g.B; t.ldb = g.ldb; t.rows = nullptr; t.C = g.C; t.ldc = g.ldc;
    t.M = g.K; t.K1 = g.M; t.K2 = g.N; t.splits = splits; t.ws = ws; t.accumulate = g.accumulate; t.prec = g.prec;
    if (batch == 1 && t.K1 <= 16 && t.M >= 1024) {             // thin left operand: streaming form (exact fp32 FMA)
      const int nchunks = (splits > 1 && ws) ? splits : 1;
      const int64_t chunk = cdiv(t.M, nchunks);
      hipLaunchKernelGGL(thin_tn_kernel, dim3((unsigned)cdiv(t.K2, 256), (unsigned)nchunks), dim3(256), 0, st, t, chunk,
                         nchunks > 1 ? ws : t.C, nchunks > 1 ? t.K2 : t.ldc, nchunks > 1 ? 1 : 0);
      MHIMX_LAUNCH_CHECK();
      if (nchunks > 1) {
        const int rc = reduce_slabs_now(st, ws, t.C, t.K1, t.K2, t.ldc, nchunks, t.accumulate);
        if (rc) return rc;
      }
      return 0;
    }
    return f32 ? launch_tn<MHIMX_PREC_F32>(st, t, batch, bt) : launch_tn<MHIMX_PREC_BF16X3>(st, t, batch, bt);
  }
  return fail(-1, "gemm_batched: unknown mode %d", mode);
}

// =================================================================================================
// transpose (32x32 LDS tile, +1 pad)
// =================================================================================================
__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t R, int64_t C) {
  __shared__ float tile[32][33];
  const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int64_t r = r0 + i, c = c0 + threadIdx.x;
    tile[i][threadIdx.x] = (r < R && c < C) ? in[r * C + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int64_t c = c0 + i, r = r0 + threadIdx.x;
    if (r < R && c < C) out[c * R + r] = tile[threadIdx.x][i];
  }
}

int transpose(hipStream_t st, const float* in, float* out, int64_t R, int64_t C) {
  MHIMX_CHECK_ARG(in && out && R > 0 && C > 0, "transpose: bad args");
  dim3 grid((unsigned)cdiv(C, 32), (unsigned)cdiv(R, 32));
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(32, 8), 0, st, in, out, R, C);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

}  // namespace mhimx

extern "C" int mhimx_gemm_nt(void* stream, const mhimx_gemm_nt_args* a) {
  if (!a) return mhimx::fail(-1, "gemm_nt: null args");
  return mhimx::gemm_nt((hipStream_t)stream, *a);
}
extern "C" int mhimx_prep_batch(void* stream, const mhimx_prep_job* jobs, int32_t n) {
  return mhimx::prep_batch((hipStream_t)stream, jobs, n);
}
extern "C" int mhimx_pair_planes(void* stream, const float* x, int64_t ldx, int64_t M, int64_t K, float* out) {
  return mhimx::pair_planes((hipStream_t)stream, x, ldx, M, K, out);
}
extern "C" int mhimx_gemm_batched(void* stream, int32_t mode, const mhimx_gemm_nt_args* a, int32_t batch, int64_t strideA,
                                  int64_t strideB, int64_t strideC, float alpha, int32_t splits, float* ws) {
  MHIMX_CHECK_ARG(a, "gemm_batched: null args");
  return mhimx::gemm_batched((hipStream_t)stream, mode, *a, batch, strideA, strideB, strideC, alpha, splits, ws);
}
extern "C" int mhimx_gemm_nn(void* stream, const mhimx_gemm_nt_args* a, float alpha, int32_t splits, float* ws) {
  if (!a) return mhimx::fail(-1, "gemm_nn: null args");
  return mhimx::gemm_nn((hipStream_t)stream, *a, alpha, splits, ws);
}
extern "C" int mhimx_gemm_tn(void* stream, const mhimx_gemm_tn_args* a) {
  if (!a) return mhimx::fail(-1, "gemm_tn: null args");
  return mhimx::gemm_tn((hipStream_t)stream, *a);
}
extern "C" int mhimx_transpose(void* stream, const float* in, float* out, int64_t R, int64_t C) {
  return mhimx::transpose((hipStream_t)stream, in, out, R, C);
}
