// mca2.hip — Merge (LayerNorm -> cross attention of k global queries over R rows -> to_out -> EMA; mhim_modules/merge.py:43-65,
// 127-144) WITHOUT the K/V projection of the rows.
//
// With k <= 6 queries the attention of Merge is a handful of softmax pools, and its linear maps commute with the pooling:
//
//   dots[h,i,r] = scale q_{i,h} . (Wk_h xn_r) = aq_{h,i} . xn_r          aq_{h,i} = scale Wk_h^T q_{i,h}    [J = 8k slots, E]  (parameters only)
//   O[i,h,:]    = sum_r Pd[h,i,r] (Wv_h xn_r) = Wv_h y_{h,i}              y_{h,i}  = sum_r Pd[h,i,r] xn_r    [J, E]  (one pool per slot)
//
// so the forward never forms K = xn Wk^T, V = xn Wv^T ([R, 1024], 1 GFLOP x 3 bf16 terms at R = 970): it scores the rows against
// J <= 48 fixed vectors and pools them J ways (~0.1 GFLOP).  The backward has the same shape:
//
//   dY[h,i]   = Wv_h^T dO[i,h]                                            (parameters x dz only)
//   dPd[r,j]  = dY_j . xn_r,   delta_j = dY_j . Y_j  (the softmax row dot),   ds = P (dP - delta)
//   dxn_r     = sum_j ds[r,j] aq_j + Pd[r,j] dY_j                         (a K = 2J product against two [J, E] matrices)
//   U_j       = sum_r ds[r,j] xn_r  -> dQ = scale Wk U,  dWk = scale Q (x) U,  dWv = dO (x) Y  (rank-k updates)
//
// i.e. no [R, 1024] gradient, no R-long weight-gradient GEMMs.  Exact algebra; only the fp32 summation order differs from the
// reference's.  Launches: parameters (1), rows forward (1), finalize + O (1), to_out (1, mca.hip's mca_out_kernel);
// backward: parameters x dz (1), rows backward incl. LayerNorm backward (1), two rank-k gradient launches.
// Built for E = 512, 8 heads x 64, k <= 6 (J <= 48), R <= 8192; other shapes take mca.hip's general path.
#include <math.h>

#include "mma_tile.hpp"

namespace mhimx {

constexpr int M2_E = 512, M2_H = 8, M2_DH = 64, M2_I = 512, M2_JP = 48, M2_JK = 64, M2_ROWS = 32, M2_THREADS = 256;
constexpr int M2_XLD = M2_E + 4;           // LDS pitch of a row tile (floats)
constexpr int M2_PLD = 36;                 // LDS pitch of the transposed [slot][row] tiles
constexpr int M2_CLD = 2 * M2_JK + 4;      // LDS pitch of the [row][2 x 64 slots] coefficient tile

typedef float m2_f4 __attribute__((ext_vector_type(4)));

struct Merge2Ws {
  float *gq, *gmean, *grstd, *Q, *aq, *aqf, *gtf_aq, *mean, *rstd, *S, *pm, *pl, *psd, *ypart, *stats, *Y, *O;
  float *dO, *dyf, *gtf_dy, *dpart, *upart, *lnpart, *dQ;
  int T;
};

int64_t merge2_ws_layout(Arena& ar, int64_t R, int64_t k, Merge2Ws* out) {
  Merge2Ws w;
  const int64_t T = cdiv(R, M2_ROWS);
  w.T = (int)T;
  w.gq = ar.take<float>(k * M2_E);
  w.gmean = ar.take<float>(k);
  w.grstd = ar.take<float>(k);
  w.Q = ar.take<float>(k * M2_I);
  w.aq = ar.take<float>(M2_JP * M2_E);
  w.aqf = ar.take<float>(3 * 16 * 64 * 8);
  w.gtf_aq = ar.take<float>(32 * 2 * 64 * 8);
  w.mean = ar.take<float>(R);
  w.rstd = ar.take<float>(R);
  w.S = ar.take<float>(R * M2_JP);
  w.pm = ar.take<float>(T * M2_JP);
  w.pl = ar.take<float>(T * M2_JP);
  w.psd = ar.take<float>(T * M2_JP);
  w.ypart = ar.take<float>(T * M2_JP * M2_E);
  w.stats = ar.take<float>(M2_JP * 2);
  w.Y = ar.take<float>(M2_JP * M2_E);
  w.O = ar.take<float>(k * M2_I);
  w.dO = ar.take<float>(k * M2_I);
  w.dyf = ar.take<float>(3 * 16 * 64 * 8);
  w.gtf_dy = ar.take<float>(32 * 2 * 64 * 8);
  w.dpart = ar.take<float>(M2_JP * 8);
  w.upart = ar.take<float>(T * M2_JP * M2_E);
  w.lnpart = ar.take<float>((T + 1) * 2 * M2_E);
  w.dQ = ar.take<float>(k * M2_I);
  if (out) *out = w;
  return ar.off;
}

int64_t merge2_ws_bytes(int64_t R, int64_t k) {
  Arena ar(nullptr, 0);
  return merge2_ws_layout(ar, R, k, nullptr);
}

bool merge2_ok(const mhimx_merge* m, int64_t R) {
  return m->E == M2_E && m->heads == M2_H && m->dim_head == M2_DH && m->k >= 1 && m->heads * m->k <= M2_JP && R >= 1 && R <= 8192 &&
         m->prec != MHIMX_PREC_F32 && aligned16(m->wq) && aligned16(m->wkv) && aligned16(m->wo) && aligned16(m->ln_w) && aligned16(m->ln_b) &&
         aligned16(m->q_param);
}

MHIMX_DEV void m2_split8(const float (&v)[8], bf8& hi, bf8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)v[i];
    hi[i] = h;
    lo[i] = (__bf16)(v[i] - (float)h);
  }
}
MHIMX_DEV f32x4 m2_mfma(const bf8& a, const bf8& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
// acc += A B^T in the 3-term bf16 form
MHIMX_DEV f32x4 m2_mfma3(const bf8& ah, const bf8& al, const bf8& bh, const bf8& bl, f32x4 c) {
  c = m2_mfma(al, bh, c);
  c = m2_mfma(ah, bl, c);
  return m2_mfma(ah, bh, c);
}
// 16 bytes of hi + 16 bytes of lo of a prep-time fragment image entry (32 bytes per lane)
MHIMX_DEV void m2_load_frag(const float* img, int entry, int lane, bf8& hi, bf8& lo) {
  const m2_f4* p = reinterpret_cast<const m2_f4*>(img + ((int64_t)entry * 64 + lane) * 8);
  hi = __builtin_bit_cast(bf8, p[0]);
  lo = __builtin_bit_cast(bf8, p[1]);
}
// store element (j, e) of a [slots, E] matrix into its two fragment images:
//   f   (B operand of  rows x slots  products, K = e):  entry (j / 16) * 16 + e / 32, lane ((e % 32) / 8) * 16 + j % 16, element e % 8
//   gtf (B operand of  rows x E  products, K = slot, padded to 64):  entry (e / 16) * 2 + j / 32, lane ((j % 32) / 8) * 16 + e % 16, element j % 8
MHIMX_DEV void m2_store_images(float* f, float* gtf, int j, int e, float v) {
  const __bf16 h = (__bf16)v, l = (__bf16)(v - (float)h);
  if (j < M2_JP) {
    __bf16* p = reinterpret_cast<__bf16*>(f) + (((int64_t)((j >> 4) * 16 + (e >> 5)) * 64 + ((e & 31) >> 3) * 16 + (j & 15)) * 16) + (e & 7);
    p[0] = h;
    p[8] = l;
  }
  __bf16* q = reinterpret_cast<__bf16*>(gtf) + (((int64_t)((e >> 4) * 2 + (j >> 5)) * 64 + ((j & 31) >> 3) * 16 + (e & 15)) * 16) + (j & 7);
  q[0] = h;
  q[8] = l;
}

// LayerNorm of one 512-wide row by one wave: lane holds e = 4 lane .. +3 and 256 + 4 lane .. +3
MHIMX_DEV void m2_ln_stats(const m2_f4& a, const m2_f4& b, float& mu, float& rs) {
  const float s = wave_sum((a[0] + a[1]) + (a[2] + a[3]) + (b[0] + b[1]) + (b[2] + b[3]));
  mu = s * (1.f / M2_E);
  float v = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) { const float d0 = a[q] - mu, d1 = b[q] - mu; v += d0 * d0 + d1 * d1; }
  rs = rsqrtf(wave_sum(v) * (1.f / M2_E) + 1e-5f);
}

// out[i][d] = rows[d][:] . vec[i][:] for NR consecutive weight rows (row pitch 512) and the 6 vectors vec[6][512] in LDS (rows >= k
// zero): one wave per row, NR / 4 rows per wave, all of them fetched before any arithmetic; the inner loop over the vectors is a
// compile-time 6 (a run-time k leaves every LDS read a dependent round trip).  gout (optional): the same values to global [i][512].
template <int NR>
MHIMX_DEV void m2_head_dots(const float* __restrict__ rows, const float* vec, int k, float* out, int out_ld, float* gout) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int PW = NR / 4;
  m2_f4 ra[PW], rb[PW];
#pragma unroll
  for (int q = 0; q < PW; ++q) {
    const float* row = rows + (int64_t)(wave * PW + q) * M2_E;
    ra[q] = *reinterpret_cast<const m2_f4*>(row + 4 * lane);
    rb[q] = *reinterpret_cast<const m2_f4*>(row + 256 + 4 * lane);
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const m2_f4 ga = *reinterpret_cast<const m2_f4*>(vec + i * M2_E + 4 * lane), gb = *reinterpret_cast<const m2_f4*>(vec + i * M2_E + 256 + 4 * lane);
#pragma unroll
    for (int q = 0; q < PW; ++q) {
      const m2_f4 a = ra[q], b = rb[q];
      float s = a[0] * ga[0] + a[1] * ga[1] + a[2] * ga[2] + a[3] * ga[3] + b[0] * gb[0] + b[1] * gb[1] + b[2] * gb[2] + b[3] * gb[3];
      s = wave_sum(s);
      if (lane == 0 && i < k) {
        out[i * out_ld + wave * PW + q] = s;
        if (gout) gout[i * M2_I + wave * PW + q] = s;
      }
    }
  }
}
// zero the rows k..5 of a [6][512] LDS block (so that loops over the queries can be a compile-time 6)
MHIMX_DEV void m2_zero_tail(float* v, int k) {
  for (int idx = k * M2_E + threadIdx.x; idx < 6 * M2_E; idx += M2_THREADS) v[idx] = 0.f;
}

// ----------------------------------------------------------------------------------------------------------------------
// 1. parameters: gq = LN(q), Q = gq Wq^T, aq[(h,i),:] = scale sum_d Q[i,h,d] Wk[h*64+d,:] and its two fragment images.
//    grid = 8 heads x 8 column blocks of 64.
// ----------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(M2_THREADS) void merge2_prep_kernel(const float* __restrict__ q_param, const float* __restrict__ ln_w,
                                                                const float* __restrict__ ln_b, const float* __restrict__ wq,
                                                                const float* __restrict__ wkv, int k, float scale, Merge2Ws w) {
  __shared__ __attribute__((aligned(16))) float gqs[6 * M2_E];
  __shared__ float qh[6 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x >> 3, eb = blockIdx.x & 7;
  const int J = M2_H * k;
  const int c = tid & 63, e = eb * 64 + c;
  float wv[64];                                               // this thread's column of the head's Wk block: in flight from the start
#pragma unroll
  for (int d = 0; d < 64; ++d) wv[d] = wkv[(int64_t)(h * 64 + d) * M2_E + e];
  m2_zero_tail(gqs, k);
  for (int idx = k * 64 + tid; idx < 6 * 64; idx += M2_THREADS) qh[idx] = 0.f;
  for (int i = wave; i < k; i += 4) {
    const float* row = q_param + (int64_t)i * M2_E;
    const m2_f4 a = *reinterpret_cast<const m2_f4*>(row + 4 * lane), b = *reinterpret_cast<const m2_f4*>(row + 256 + 4 * lane);
    float mu, rs;
    m2_ln_stats(a, b, mu, rs);
    const m2_f4 wa = *reinterpret_cast<const m2_f4*>(ln_w + 4 * lane), wb = *reinterpret_cast<const m2_f4*>(ln_w + 256 + 4 * lane);
    const m2_f4 ba = *reinterpret_cast<const m2_f4*>(ln_b + 4 * lane), bb = *reinterpret_cast<const m2_f4*>(ln_b + 256 + 4 * lane);
    const m2_f4 ya = (a - mu) * rs * wa + ba, yb = (b - mu) * rs * wb + bb;
    *reinterpret_cast<m2_f4*>(gqs + i * M2_E + 4 * lane) = ya;
    *reinterpret_cast<m2_f4*>(gqs + i * M2_E + 256 + 4 * lane) = yb;
    if (blockIdx.x == 0) {
      *reinterpret_cast<m2_f4*>(w.gq + i * M2_E + 4 * lane) = ya;
      *reinterpret_cast<m2_f4*>(w.gq + i * M2_E + 256 + 4 * lane) = yb;
      if (lane == 0) { w.gmean[i] = mu; w.grstd[i] = rs; }
    }
  }
  __syncthreads();
  m2_head_dots<64>(wq + (int64_t)h * 64 * M2_E, gqs, k, qh, 64, eb == 0 ? w.Q + h * 64 : nullptr);     // Q of this head
  __syncthreads();
  // aq for the 64 columns of this block, then the images
  for (int i = tid >> 6; i < k; i += 4) {
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < 64; ++d) acc += qh[i * 64 + d] * wv[d];
    acc *= scale;
    const int j = h * k + i;
    m2_store_images(w.aqf, w.gtf_aq, j, e, acc);
  }
  for (int j = J + h; j < M2_JK; j += M2_H)                 // zero padding slots (this head's share), 4 threads per column
    if ((tid >> 6) == ((j - J) >> 3) % 4) m2_store_images(w.aqf, w.gtf_aq, j, e, 0.f);
}

// ----------------------------------------------------------------------------------------------------------------------
// shared pieces of the two row kernels
// ----------------------------------------------------------------------------------------------------------------------
// rows of the tile -> xhat = (x - mean) rstd in LDS [32][516]; HAVE_STATS: mean / rstd are read instead of computed.
// Also stages the LayerNorm weight and bias in LDS (lnw[512], lnb[512]).
template <bool HAVE_STATS>
MHIMX_DEV void m2_load_rows(const float* __restrict__ X, const int64_t* __restrict__ xrows, int64_t R, int64_t row0, float* xh, float* mean,
                            float* rstd, const float* __restrict__ ln_w, const float* __restrict__ ln_b, float* lnw, float* lnb, float* rs_tile) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  m2_f4 a[8], b[8];
  float mu8[8], rs8[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {                               // all 16 row loads of this wave in flight
    const int64_t n = row0 + wave + 4 * q;
    const int64_t nc = n < R ? n : R - 1;
    const float* src = X + (xrows ? xrows[nc] : nc) * M2_E;
    a[q] = *reinterpret_cast<const m2_f4*>(src + 4 * lane);
    b[q] = *reinterpret_cast<const m2_f4*>(src + 256 + 4 * lane);
    if (HAVE_STATS) { mu8[q] = mean[nc]; rs8[q] = rstd[nc]; }
  }
  if (tid < 128) *reinterpret_cast<m2_f4*>(lnw + 4 * tid) = *reinterpret_cast<const m2_f4*>(ln_w + 4 * tid);
  else *reinterpret_cast<m2_f4*>(lnb + 4 * (tid - 128)) = *reinterpret_cast<const m2_f4*>(ln_b + 4 * (tid - 128));
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int rr = wave + 4 * q;
    const int64_t n = row0 + rr;
    float mu, rs;
    if (HAVE_STATS) {
      mu = mu8[q];
      rs = rs8[q];
      if (lane == 0) rs_tile[rr] = rs;
    } else {
      m2_ln_stats(a[q], b[q], mu, rs);
      if (lane == 0 && n < R) { mean[n] = mu; rstd[n] = rs; }
    }
    m2_f4 ya = (a[q] - mu) * rs, yb = (b[q] - mu) * rs;
    if (n >= R) { ya = m2_f4{0.f, 0.f, 0.f, 0.f}; yb = ya; }
    *reinterpret_cast<m2_f4*>(xh + rr * M2_XLD + 4 * lane) = ya;
    *reinterpret_cast<m2_f4*>(xh + rr * M2_XLD + 256 + 4 * lane) = yb;
  }
}

// the 12 B fragments (3 slot blocks x this wave's 4 k-steps) of a rows x slots product, fetched before the rows are even loaded
struct M2Frags { m2_f4 h[3][4], l[3][4]; };
MHIMX_DEV void m2_fetch_frags(const float* __restrict__ img, M2Frags& f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int nb = 0; nb < 3; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const m2_f4* p = reinterpret_cast<const m2_f4*>(img + ((int64_t)(nb * 16 + 4 * wave + q) * 64 + lane) * 8);
      f.h[nb][q] = p[0];
      f.l[nb][q] = p[1];
    }
}

// red[wave][32][48] = (xhat w + b)[32 x 512] . img^T over this wave's quarter of the 512-deep reduction (3-term bf16)
MHIMX_DEV void m2_rows_times_slots(const float* xh, const float* lnw, const float* lnb, const M2Frags& f, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kg = lane >> 4;
  f32x4 acc[2][3];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) acc[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e0 = (4 * wave + q) * 32 + kg * 8;
    const m2_f4 w0 = *reinterpret_cast<const m2_f4*>(lnw + e0), w1 = *reinterpret_cast<const m2_f4*>(lnw + e0 + 4);
    const m2_f4 b0 = *reinterpret_cast<const m2_f4*>(lnb + e0), b1 = *reinterpret_cast<const m2_f4*>(lnb + e0 + 4);
    bf8 ah[2], al[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const float* p = xh + (rb * 16 + r16) * M2_XLD + e0;
      const m2_f4 x0 = *reinterpret_cast<const m2_f4*>(p) * w0 + b0, x1 = *reinterpret_cast<const m2_f4*>(p + 4) * w1 + b1;
      const float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
      m2_split8(v, ah[rb], al[rb]);
    }
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
      const bf8 bh = __builtin_bit_cast(bf8, f.h[nb][q]), bl = __builtin_bit_cast(bf8, f.l[nb][q]);
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) acc[rb][nb] = m2_mfma3(ah[rb], al[rb], bh, bl, acc[rb][nb]);
    }
  }
  float* out = red + wave * (M2_ROWS * M2_JP);
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
      for (int i = 0; i < 4; ++i) out[(rb * 16 + 4 * kg + i) * M2_JP + nb * 16 + r16] = acc[rb][nb][i];
}

// part[slot][:] = sum_r coefT[slot][r] xhat[r][:]   ([48 x 32] . [32 x 512], 3-term bf16): one 32-deep MFMA step per 16 x 16 block
MHIMX_DEV void m2_pool_rows(const float* coefT /* LDS [48][36] */, const float* xh, float* __restrict__ part /* global [48][512] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kg = lane >> 4;
  bf8 ah[3], al[3];
#pragma unroll
  for (int jb = 0; jb < 3; ++jb) {
    const float* p = coefT + (jb * 16 + r16) * M2_PLD + kg * 8;
    const m2_f4 c0 = *reinterpret_cast<const m2_f4*>(p), c1 = *reinterpret_cast<const m2_f4*>(p + 4);
    const float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
    m2_split8(v, ah[jb], al[jb]);
  }
#pragma unroll 2
  for (int eb = 8 * wave; eb < 8 * wave + 8; ++eb) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = xh[(kg * 8 + q) * M2_XLD + eb * 16 + r16];
    bf8 bh, bl;
    m2_split8(v, bh, bl);
#pragma unroll
    for (int jb = 0; jb < 3; ++jb) {
      const f32x4 acc = m2_mfma3(ah[jb], al[jb], bh, bl, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int i = 0; i < 4; ++i) part[(int64_t)(jb * 16 + 4 * kg + i) * M2_E + eb * 16 + r16] = acc[i];
    }
  }
}

MHIMX_DEV bool m2_keep(uint64_t seed, int j, int64_t r, float p) { return drop_keep(seed, (uint64_t)j, (uint32_t)r, p); }

// ----------------------------------------------------------------------------------------------------------------------
// 2. rows forward: LayerNorm, scores against the J slots, per-tile softmax partials, pooled rows.   grid = ceil(R / 32)
// ----------------------------------------------------------------------------------------------------------------------
constexpr size_t M2_FWD_SMEM = (size_t)(M2_ROWS * M2_XLD + 4 * M2_ROWS * M2_JP + M2_JP * M2_PLD + 2 * M2_E) * sizeof(float);

__global__ __launch_bounds__(M2_THREADS) void merge2_rows_fwd_kernel(const float* __restrict__ X, const int64_t* __restrict__ xrows, int64_t R,
                                                                    const float* __restrict__ ln_w, const float* __restrict__ ln_b, int J,
                                                                    float drop_p, uint64_t seed0, const uint64_t* __restrict__ tick, Merge2Ws w) {
  extern __shared__ __attribute__((aligned(16))) float m2sm[];
  float* xh = m2sm;                                  // [32][516]
  float* red = xh + M2_ROWS * M2_XLD;                // [4][32][48]
  float* pdT = red + 4 * M2_ROWS * M2_JP;            // [48][36]
  float* lnw = pdT + M2_JP * M2_PLD;                 // [512]
  float* lnb = lnw + M2_E;                           // [512]
  const int tid = threadIdx.x;
  const int t = blockIdx.x;
  const int64_t row0 = (int64_t)t * M2_ROWS;
  M2Frags fr;
  m2_fetch_frags(w.aqf, fr);
  m2_load_rows<false>(X, xrows, R, row0, xh, w.mean, w.rstd, ln_w, ln_b, lnw, lnb, nullptr);
  __syncthreads();
  m2_rows_times_slots(xh, lnw, lnb, fr, red);
  __syncthreads();
  for (int idx = tid; idx < M2_ROWS * M2_JP; idx += M2_THREADS) {
    const float s = (red[idx] + red[M2_ROWS * M2_JP + idx]) + (red[2 * M2_ROWS * M2_JP + idx] + red[3 * M2_ROWS * M2_JP + idx]);
    red[idx] = s;
    const int r = idx / M2_JP;
    if (row0 + r < R) w.S[(row0 + r) * M2_JP + (idx - r * M2_JP)] = s;
  }
  __syncthreads();
  // per-slot softmax partials of the tile: 4 threads per slot (8 rows each), combined through LDS
  float* sc = red + M2_ROWS * M2_JP;                 // [3][4][48] scratch (the partial-product slabs 1..3 are free)
  const int j = tid % M2_JP, rq = tid / M2_JP;        // rq < 4 for the first 192 threads
  const int nv = (int)((R - row0) < M2_ROWS ? (R - row0) : M2_ROWS);
  float sreg[8], m = -INFINITY;
  if (rq < 4) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = rq * 8 + q;
      sreg[q] = red[r * M2_JP + j];
      if (j < J && r < nv) m = fmaxf(m, sreg[q]);
    }
    sc[rq * M2_JP + j] = m;
  }
  __syncthreads();
  if (rq < 4) {
    m = fmaxf(fmaxf(sc[j], sc[M2_JP + j]), fmaxf(sc[2 * M2_JP + j], sc[3 * M2_JP + j]));
    const uint64_t seed = drop_p > 0.f ? eff_seed(seed0, tick) : 0;
    const float ks = 1.f / (1.f - drop_p);
    float l = 0.f, sd = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = rq * 8 + q;
      float p = 0.f, pd = 0.f;
      if (j < J && r < nv) {
        p = __expf(sreg[q] - m);
        pd = (drop_p > 0.f && !m2_keep(seed, j, row0 + r, drop_p)) ? 0.f : p * ks;
      }
      l += p;
      sd += pd;
      pdT[j * M2_PLD + r] = pd;
    }
    sc[(4 + rq) * M2_JP + j] = l;
    sc[(8 + rq) * M2_JP + j] = sd;
  }
  __syncthreads();
  if (tid < M2_JP) {
    w.pm[t * M2_JP + tid] = fmaxf(fmaxf(sc[tid], sc[M2_JP + tid]), fmaxf(sc[2 * M2_JP + tid], sc[3 * M2_JP + tid]));
    w.pl[t * M2_JP + tid] = (sc[4 * M2_JP + tid] + sc[5 * M2_JP + tid]) + (sc[6 * M2_JP + tid] + sc[7 * M2_JP + tid]);
    w.psd[t * M2_JP + tid] = (sc[8 * M2_JP + tid] + sc[9 * M2_JP + tid]) + (sc[10 * M2_JP + tid] + sc[11 * M2_JP + tid]);
  }
  m2_pool_rows(pdT, xh, w.ypart + (int64_t)t * M2_JP * M2_E);
}

// ----------------------------------------------------------------------------------------------------------------------
// 3a. merge the T tile partials of every slot: out[j][e] = (sum_t part[t][j][e] wgt_t) scaled.   grid = J slots x 4 column blocks
//     of 128; 256 threads = 128 columns x 2 halves of the tiles, 16 loads in flight per thread.
//     SOFTMAX: wgt_t = e^{pm_t - M} / L (online softmax merge, fixed order), out = y ln_w + (sum_t psd_t wgt_t) ln_b, stats = (M, L);
//     else wgt_t = 1 and out = u ln_w.
// ----------------------------------------------------------------------------------------------------------------------
template <bool SOFTMAX>
__global__ __launch_bounds__(M2_THREADS) void merge2_partials_kernel(const float* __restrict__ part, const float* __restrict__ ln_w,
                                                                    const float* __restrict__ ln_b, float* __restrict__ out, Merge2Ws w) {
  __shared__ float wt[256];
  __shared__ float red[8];
  __shared__ float half1[128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = blockIdx.x >> 2, e = (blockIdx.x & 3) * 128 + (tid & 127), half = tid >> 7;
  const int T = w.T;
  float sdl = 0.f;
  if (SOFTMAX) {
    const float pm = tid < T ? w.pm[tid * M2_JP + j] : -INFINITY;
    const float pl = tid < T ? w.pl[tid * M2_JP + j] : 0.f;
    const float ps = tid < T ? w.psd[tid * M2_JP + j] : 0.f;
    float m = wave_max(pm);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    const float M = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float wgt = tid < T ? __expf(pm - M) : 0.f;
    const float l = wave_sum(pl * wgt), sd = wave_sum(ps * wgt);
    if (lane == 0) { red[4 + wave] = l; wt[252 + wave] = sd; }      // (wt[252..255] are beyond any tile: T <= 256 uses wt[0..T-1])
    __syncthreads();
    const float L = (red[4] + red[5]) + (red[6] + red[7]);
    const float SD = (wt[252] + wt[253]) + (wt[254] + wt[255]);
    __syncthreads();
    wt[tid] = wgt / L;
    sdl = SD / L;
    if ((blockIdx.x & 3) == 0 && tid == 0) { w.stats[2 * j] = M; w.stats[2 * j + 1] = L; }
  } else {
    wt[tid] = tid < T ? 1.f : 0.f;
  }
  __syncthreads();
  float acc = 0.f;
  const float* pj = part + (int64_t)j * M2_E + e;
#pragma unroll 1
  for (int t0 = 0; t0 < T; t0 += 32) {
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int t = t0 + half * 16 + q;
      v[q] = t < T ? pj[(int64_t)t * M2_JP * M2_E] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) acc += v[q] * wt[(t0 + half * 16 + q) & 255];
  }
  if (half == 1) half1[tid & 127] = acc;
  __syncthreads();
  if (half == 0) {
    acc += half1[tid];
    out[j * M2_E + e] = SOFTMAX ? acc * ln_w[e] + sdl * ln_b[e] : acc * ln_w[e];
  }
}

// ----------------------------------------------------------------------------------------------------------------------
// 3b. O[i, h*64+d] = Wv[h*64+d, :] . Y[(h,i), :].   grid = 8 heads x 4 quarters of the head's 64 columns
// ----------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(M2_THREADS) void merge2_o_kernel(const float* __restrict__ wkv, int k, Merge2Ws w) {
  __shared__ __attribute__((aligned(16))) float ys[6 * M2_E];
  __shared__ float oh[6 * 16];
  const int tid = threadIdx.x;
  const int h = blockIdx.x >> 2, qd = blockIdx.x & 3;
  m2_zero_tail(ys, k);
  for (int idx = tid; idx < k * (M2_E / 4); idx += M2_THREADS)
    reinterpret_cast<m2_f4*>(ys)[idx] = reinterpret_cast<const m2_f4*>(w.Y + (int64_t)h * k * M2_E)[idx];
  __syncthreads();
  m2_head_dots<16>(wkv + (int64_t)(M2_I + h * 64 + qd * 16) * M2_E, ys, k, oh, 16, w.O + h * 64 + qd * 16);       // the V half of to_kv
}

// ----------------------------------------------------------------------------------------------------------------------
// 4. backward, parameters x dz: dz0 = dz keep/(1-p), d_bo, dO = dz0 Wo, dY[(h,i),:] = sum_d dO[i,h,d] Wv[h*64+d,:] (as the two
//    fragment images), delta partials dY.Y.   grid = 8 heads x 8 column blocks of 64.
// ----------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(M2_THREADS) void merge2_bwd_pre_kernel(const float* __restrict__ dz, const float* __restrict__ wo_t,
                                                                   const float* __restrict__ wkv, int k, float drop_p, uint64_t seed0,
                                                                   const uint64_t* __restrict__ tick, float* __restrict__ d_bo, int accumulate,
                                                                   Merge2Ws w) {
  __shared__ __attribute__((aligned(16))) float dzs[6 * M2_E];
  __shared__ float doh[6 * 64];
  __shared__ float dys[6 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x >> 3, eb = blockIdx.x & 7;
  const int J = M2_H * k;
  const int c = tid & 63, e = eb * 64 + c;
  float wv[64];                                               // this thread's column of the head's Wv block: in flight from the start
#pragma unroll
  for (int d = 0; d < 64; ++d) wv[d] = wkv[(int64_t)(M2_I + h * 64 + d) * M2_E + e];
  float yv[2] = {0.f, 0.f};
  for (int i = tid >> 6, q = 0; i < k; i += 4, ++q) yv[q] = w.Y[(h * k + i) * M2_E + e];
  const uint64_t seed = drop_p > 0.f ? eff_seed(seed0, tick) : 0;
  const float ks = 1.f / (1.f - drop_p);
  for (int idx = tid; idx < k * M2_E; idx += M2_THREADS) {
    const int i = idx >> 9, ee = idx & 511;
    float v = dz[idx];
    if (drop_p > 0.f) v = drop_keep(seed, (uint64_t)i, (uint32_t)ee, drop_p) ? v * ks : 0.f;
    dzs[idx] = v;
  }
  m2_zero_tail(dzs, k);
  __syncthreads();
  if (h == 0 && tid < 64) {
    float s = 0.f;
    for (int i = 0; i < k; ++i) s += dzs[i * M2_E + e];
    d_bo[e] = accumulate ? d_bo[e] + s : s;
  }
  m2_head_dots<64>(wo_t + (int64_t)h * 64 * M2_E, dzs, k, doh, 64, eb == 0 ? w.dO + h * 64 : nullptr);
  __syncthreads();
  for (int i = tid >> 6, q = 0; i < k; i += 4, ++q) {
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < 64; ++d) acc += doh[i * 64 + d] * wv[d];
    const int j = h * k + i;
    m2_store_images(w.dyf, w.gtf_dy, j, e, acc);
    const float s = wave_sum(acc * yv[q]);                    // (a wave = one query i x the 64 columns of this block)
    if (lane == 0) w.dpart[j * 8 + eb] = s;
  }
  for (int j = J + h; j < M2_JK; j += M2_H)
    if ((tid >> 6) == ((j - J) >> 3) % 4) m2_store_images(w.dyf, w.gtf_dy, j, e, 0.f);
  (void)dys; (void)wave;
}

// ----------------------------------------------------------------------------------------------------------------------
// 5. rows backward: dPd = xn dY^T, softmax backward, dxn = ds aq + Pd dY, LayerNorm backward (dX scattered to the rows' places,
//    per-tile d_ln_w / d_ln_b partials), pooled U partials.   grid = ceil(R / 32)
// ----------------------------------------------------------------------------------------------------------------------
constexpr size_t M2_BWD_SMEM = (size_t)(2 * M2_ROWS * M2_XLD + M2_ROWS * M2_CLD + M2_JP * M2_PLD + 2 * M2_E + 3 * M2_JK + M2_ROWS) * sizeof(float);
static_assert(M2_ROWS * M2_CLD >= 8 * M2_E, "the LayerNorm partials reuse the coefficient tile");

__global__ __launch_bounds__(M2_THREADS) void merge2_rows_bwd_kernel(const float* __restrict__ X, const int64_t* __restrict__ xrows, int64_t R,
                                                                    const float* __restrict__ ln_w, const float* __restrict__ ln_b, int J,
                                                                    float drop_p, uint64_t seed0, const uint64_t* __restrict__ tick,
                                                                    float* __restrict__ dX, Merge2Ws w) {
  extern __shared__ __attribute__((aligned(16))) float m2sm[];
  float* xh = m2sm;                                   // [32][516]
  float* dxs = xh + M2_ROWS * M2_XLD;                 // [32][516]; first the [4][32][48] reduction buffer of dPd
  float* cf = dxs + M2_ROWS * M2_XLD;                 // [32][132]: ds (slots 0..63) | Pd (64..127)
  float* dsT = cf + M2_ROWS * M2_CLD;                 // [48][36]
  float* lnw = dsT + M2_JP * M2_PLD;                  // [512]
  float* lnb = lnw + M2_E;                            // [512]
  float* sst = lnb + M2_E;                            // [64][3]: softmax max, 1 / sum, delta of every slot
  float* rst = sst + 3 * M2_JK;                       // [32] rstd of the tile's rows
  float* lnred = cf;                                  // [4][2][512]: the coefficient tile is in registers by then
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kg = lane >> 4;
  const int t = blockIdx.x;
  const int64_t row0 = (int64_t)t * M2_ROWS;
  M2Frags fr;
  m2_fetch_frags(w.dyf, fr);
  if (tid < M2_JK) {
    const int j = tid;
    float mx = 0.f, il = 0.f, de = 0.f;
    if (j < J) {
      mx = w.stats[2 * j];
      il = 1.f / w.stats[2 * j + 1];
      const m2_f4 d0 = *reinterpret_cast<const m2_f4*>(w.dpart + j * 8), d1 = *reinterpret_cast<const m2_f4*>(w.dpart + j * 8 + 4);
      de = ((d0[0] + d0[1]) + (d0[2] + d0[3])) + ((d1[0] + d1[1]) + (d1[2] + d1[3]));
    }
    sst[3 * j] = mx; sst[3 * j + 1] = il; sst[3 * j + 2] = de;
  }
  // the scores of the tile's (row, slot) pairs this thread will turn into probabilities: in flight under the row loads
  float sv[M2_ROWS * M2_JK / M2_THREADS];
#pragma unroll
  for (int q = 0; q < M2_ROWS * M2_JK / M2_THREADS; ++q) {
    const int idx = tid + q * M2_THREADS, r = idx >> 6, j = idx & 63;
    sv[q] = (j < J && row0 + r < R) ? w.S[(row0 + r) * M2_JP + j] : 0.f;
  }
  m2_load_rows<true>(X, xrows, R, row0, xh, w.mean, w.rstd, ln_w, ln_b, lnw, lnb, rst);
  __syncthreads();
  m2_rows_times_slots(xh, lnw, lnb, fr, dxs);
  __syncthreads();
  // ---- softmax backward per (row, slot)
  {
    const uint64_t seed = drop_p > 0.f ? eff_seed(seed0, tick) : 0;
    const float ks = 1.f / (1.f - drop_p);
#pragma unroll
    for (int q = 0; q < M2_ROWS * M2_JK / M2_THREADS; ++q) {
      const int idx = tid + q * M2_THREADS, r = idx >> 6, j = idx & 63;
      float ds = 0.f, pd = 0.f;
      if (j < J && row0 + r < R) {
        const int qq = r * M2_JP + j;
        const float dpd = (dxs[qq] + dxs[M2_ROWS * M2_JP + qq]) + (dxs[2 * M2_ROWS * M2_JP + qq] + dxs[3 * M2_ROWS * M2_JP + qq]);
        const float p = __expf(sv[q] - sst[3 * j]) * sst[3 * j + 1];
        const float kf = (drop_p > 0.f && !m2_keep(seed, j, row0 + r, drop_p)) ? 0.f : ks;
        pd = p * kf;
        ds = p * (dpd * kf - sst[3 * j + 2]);
      }
      cf[r * M2_CLD + j] = ds;
      cf[r * M2_CLD + M2_JK + j] = pd;
      if (j < M2_JP) dsT[j * M2_PLD + r] = ds;
    }
  }
  __syncthreads();
  // ---- dxn[32 x 512] = cf[32 x 128] . [aq ; dY]  (K = 128 = 4 steps of 32; B from the transposed fragment images, 4 column blocks
  //      = 32 fragment loads in flight at a time)
  {
    bf8 ah[2][4], al[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const float* p = cf + (rb * 16 + r16) * M2_CLD + ks * 32 + kg * 8;
        const m2_f4 c0 = *reinterpret_cast<const m2_f4*>(p), c1 = *reinterpret_cast<const m2_f4*>(p + 4);
        const float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
        m2_split8(v, ah[rb][ks], al[rb][ks]);
      }
    __syncthreads();                                  // (every wave has read its dPd partials and the coefficient tile: both free)
#pragma unroll 1
    for (int eb0 = 8 * wave; eb0 < 8 * wave + 8; eb0 += 4) {
      m2_f4 bh[4][4], bl[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const m2_f4* p = reinterpret_cast<const m2_f4*>((ks < 2 ? w.gtf_aq : w.gtf_dy) + ((int64_t)((eb0 + q) * 2 + (ks & 1)) * 64 + lane) * 8);
          bh[q][ks] = p[0];
          bl[q][ks] = p[1];
        }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
            acc[rb] = m2_mfma3(ah[rb][ks], al[rb][ks], __builtin_bit_cast(bf8, bh[q][ks]), __builtin_bit_cast(bf8, bl[q][ks]), acc[rb]);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int i = 0; i < 4; ++i) dxs[(rb * 16 + 4 * kg + i) * M2_XLD + (eb0 + q) * 16 + r16] = acc[rb][i];
      }
    }
  }
  __syncthreads();
  // ---- LayerNorm backward, 8 rows per wave: dxhat = dxn w;  dx = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat))
  {
    const m2_f4 wa = *reinterpret_cast<const m2_f4*>(lnw + 4 * lane), wb = *reinterpret_cast<const m2_f4*>(lnw + 256 + 4 * lane);
    m2_f4 dwa = m2_f4{0.f, 0.f, 0.f, 0.f}, dwb = dwa, dba = dwa, dbb = dwa;
    int64_t dst_row[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int64_t n = row0 + wave + 4 * q;
      dst_row[q] = n < R ? (xrows ? xrows[n] : n) : -1;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int rr = wave + 4 * q;
      if (dst_row[q] < 0) continue;
      const m2_f4 ga = *reinterpret_cast<const m2_f4*>(dxs + rr * M2_XLD + 4 * lane), gb = *reinterpret_cast<const m2_f4*>(dxs + rr * M2_XLD + 256 + 4 * lane);
      const m2_f4 xa = *reinterpret_cast<const m2_f4*>(xh + rr * M2_XLD + 4 * lane), xb = *reinterpret_cast<const m2_f4*>(xh + rr * M2_XLD + 256 + 4 * lane);
      dwa += ga * xa; dwb += gb * xb; dba += ga; dbb += gb;
      const m2_f4 ha = ga * wa, hb = gb * wb;
      float s1 = (ha[0] + ha[1]) + (ha[2] + ha[3]) + (hb[0] + hb[1]) + (hb[2] + hb[3]);
      const m2_f4 pa = ha * xa, pb = hb * xb;
      float s2 = (pa[0] + pa[1]) + (pa[2] + pa[3]) + (pb[0] + pb[1]) + (pb[2] + pb[3]);
      s1 = wave_sum(s1) * (1.f / M2_E);
      s2 = wave_sum(s2) * (1.f / M2_E);
      const float rs = rst[rr];
      float* dst = dX + dst_row[q] * M2_E;
      *reinterpret_cast<m2_f4*>(dst + 4 * lane) = (ha - s1 - xa * s2) * rs;
      *reinterpret_cast<m2_f4*>(dst + 256 + 4 * lane) = (hb - s1 - xb * s2) * rs;
    }
    __syncthreads();                                  // (lnred aliases cf; nobody reads cf any more, but keep the waves together)
    *reinterpret_cast<m2_f4*>(lnred + (wave * 2) * M2_E + 4 * lane) = dwa;
    *reinterpret_cast<m2_f4*>(lnred + (wave * 2) * M2_E + 256 + 4 * lane) = dwb;
    *reinterpret_cast<m2_f4*>(lnred + (wave * 2 + 1) * M2_E + 4 * lane) = dba;
    *reinterpret_cast<m2_f4*>(lnred + (wave * 2 + 1) * M2_E + 256 + 4 * lane) = dbb;
  }
  __syncthreads();
  for (int idx = tid; idx < 2 * M2_E; idx += M2_THREADS)
    w.lnpart[(int64_t)t * 2 * M2_E + idx] = (lnred[idx] + lnred[2 * M2_E + idx]) + (lnred[4 * M2_E + idx] + lnred[6 * M2_E + idx]);
  // ---- pooled U partial (of xhat; the LayerNorm weight is applied when the partials are merged)
  m2_pool_rows(dsT, xh, w.upart + (int64_t)t * M2_JP * M2_E);
}

// ----------------------------------------------------------------------------------------------------------------------
// 6. rank-k gradients, first launch (U merged by merge2_partials_kernel<false> before).   blocks 0..31: (head, quarter): dQ = scale Wk U and
//    16 + 16 rows of d_wkv (K part: scale Q (x) U, V part: dO (x) Y);   blocks 32..47: 32 rows of d_wo = dz0^T O.
// ----------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(M2_THREADS) void merge2_grads1_kernel(const float* __restrict__ dz, const float* __restrict__ U,
                                                                  const float* __restrict__ wkv, int k, float scale, float drop_p, uint64_t seed0,
                                                                  const uint64_t* __restrict__ tick, float* __restrict__ d_wkv,
                                                                  float* __restrict__ d_wo, int accumulate, Merge2Ws w) {
  __shared__ __attribute__((aligned(16))) float us[6 * M2_E];
  __shared__ __attribute__((aligned(16))) float ysh[6 * M2_E];
  __shared__ __attribute__((aligned(16))) float qd[32 * 12];        // [row][6 x scale Q | 6 x dO]  (d_wo blocks: [32][8] dz0)
  __shared__ float dqh[6 * 16];
  const int tid = threadIdx.x;
  if (blockIdx.x >= 32) {
    // 32 rows of d_wo[e, c] = sum_i dz0[i, e] O[i, c]: thread = two columns c, the k values of O in registers
    const int e0 = (blockIdx.x - 32) * 32;
    const uint64_t seed = drop_p > 0.f ? eff_seed(seed0, tick) : 0;
    const float ks = 1.f / (1.f - drop_p);
    float o0[6], o1[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      o0[i] = i < k ? w.O[i * M2_I + tid] : 0.f;
      o1[i] = i < k ? w.O[i * M2_I + tid + 256] : 0.f;
    }
    for (int idx = tid; idx < 32 * 8; idx += M2_THREADS) {
      const int r = idx >> 3, i = idx & 7, e = e0 + r;
      float v = 0.f;
      if (i < k) {
        v = dz[i * M2_E + e];
        if (drop_p > 0.f) v = drop_keep(seed, (uint64_t)i, (uint32_t)e, drop_p) ? v * ks : 0.f;
      }
      qd[idx] = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < 32; ++r) {
      const m2_f4 z0 = *reinterpret_cast<const m2_f4*>(qd + r * 8), z1 = *reinterpret_cast<const m2_f4*>(qd + r * 8 + 4);
      const float s0 = z0[0] * o0[0] + z0[1] * o0[1] + z0[2] * o0[2] + z0[3] * o0[3] + z1[0] * o0[4] + z1[1] * o0[5];
      const float s1 = z0[0] * o1[0] + z0[1] * o1[1] + z0[2] * o1[2] + z0[3] * o1[3] + z1[0] * o1[4] + z1[1] * o1[5];
      float* o = d_wo + (int64_t)(e0 + r) * M2_I + tid;
      o[0] = accumulate ? o[0] + s0 : s0;
      o[256] = accumulate ? o[256] + s1 : s1;
    }
    return;
  }
  const int h = blockIdx.x >> 2, qr = blockIdx.x & 3;             // 16 of the head's 64 rows
  m2_zero_tail(us, k);
  for (int idx = tid; idx < k * (M2_E / 4); idx += M2_THREADS) reinterpret_cast<m2_f4*>(us)[idx] = reinterpret_cast<const m2_f4*>(U + (int64_t)h * k * M2_E)[idx];
  float y0[6], y1[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    y0[i] = i < k ? w.Y[(h * k + i) * M2_E + tid] : 0.f;
    y1[i] = i < k ? w.Y[(h * k + i) * M2_E + tid + 256] : 0.f;
  }
  for (int idx = tid; idx < 16 * 12; idx += M2_THREADS) {
    const int dl = idx / 12, c = idx - dl * 12, i = c % 6, d = h * 64 + qr * 16 + dl;
    qd[idx] = i < k ? (c < 6 ? scale * w.Q[i * M2_I + d] : w.dO[i * M2_I + d]) : 0.f;
  }
  __syncthreads();
  m2_head_dots<16>(wkv + (int64_t)(h * 64 + qr * 16) * M2_E, us, k, dqh, 16, nullptr);             // the K half of to_kv
  float u0[6], u1[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) { u0[i] = us[i * M2_E + tid]; u1[i] = us[i * M2_E + tid + 256]; }
  __syncthreads();
  if (tid < 16 * 6 && (tid / 16) < k) w.dQ[(tid / 16) * M2_I + h * 64 + qr * 16 + (tid & 15)] = scale * dqh[tid];
#pragma unroll 4
  for (int dl = 0; dl < 16; ++dl) {
    const m2_f4 c0 = *reinterpret_cast<const m2_f4*>(qd + dl * 12), c1 = *reinterpret_cast<const m2_f4*>(qd + dl * 12 + 4),
                c2 = *reinterpret_cast<const m2_f4*>(qd + dl * 12 + 8);
    const float q6[6] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1]}, o6[6] = {c1[2], c1[3], c2[0], c2[1], c2[2], c2[3]};
    float sk0 = 0.f, sk1 = 0.f, sv0 = 0.f, sv1 = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) { sk0 += q6[i] * u0[i]; sk1 += q6[i] * u1[i]; sv0 += o6[i] * y0[i]; sv1 += o6[i] * y1[i]; }
    float* ok = d_wkv + (int64_t)(h * 64 + qr * 16 + dl) * M2_E + tid;
    float* ov = d_wkv + (int64_t)(M2_I + h * 64 + qr * 16 + dl) * M2_E + tid;
    ok[0] = accumulate ? ok[0] + sk0 : sk0;
    ok[256] = accumulate ? ok[256] + sk1 : sk1;
    ov[0] = accumulate ? ov[0] + sv0 : sv0;
    ov[256] = accumulate ? ov[256] + sv1 : sv1;
  }
}

// ----------------------------------------------------------------------------------------------------------------------
// 7. rank-k gradients, second launch (needs all of dQ).   blocks 0..15: 32 rows of d_wq = dQ^T gq;   blocks 16..23: 64 columns of
//    dgq = dQ Wq and their LayerNorm-parameter gradients (the queries themselves are not trained) -> partial row T of lnpart.
// ----------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(M2_THREADS) void merge2_grads2_kernel(const float* __restrict__ q_param, const float* __restrict__ wq, int k,
                                                                  float* __restrict__ d_wq, int accumulate, Merge2Ws w) {
  __shared__ float dqs[6 * M2_I];
  __shared__ __attribute__((aligned(16))) float part[4 * 6 * 64];
  const int tid = threadIdx.x;
  if (blockIdx.x < 16) {
    // 32 rows of d_wq[c, e] = sum_i dQ[i, c] gq[i, e]: thread = two columns e, the k values of gq in registers
    const int c0 = blockIdx.x * 32;
    float g0[6], g1[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      g0[i] = i < k ? w.gq[i * M2_E + tid] : 0.f;
      g1[i] = i < k ? w.gq[i * M2_E + tid + 256] : 0.f;
    }
    for (int idx = tid; idx < 32 * 8; idx += M2_THREADS) {
      const int r = idx >> 3, i = idx & 7;
      part[idx] = i < k ? w.dQ[i * M2_I + c0 + r] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < 32; ++r) {
      const m2_f4 z0 = *reinterpret_cast<const m2_f4*>(part + r * 8), z1 = *reinterpret_cast<const m2_f4*>(part + r * 8 + 4);
      const float s0 = z0[0] * g0[0] + z0[1] * g0[1] + z0[2] * g0[2] + z0[3] * g0[3] + z1[0] * g0[4] + z1[1] * g0[5];
      const float s1 = z0[0] * g1[0] + z0[1] * g1[1] + z0[2] * g1[2] + z0[3] * g1[3] + z1[0] * g1[4] + z1[1] * g1[5];
      float* o = d_wq + (int64_t)(c0 + r) * M2_E + tid;
      o[0] = accumulate ? o[0] + s0 : s0;
      o[256] = accumulate ? o[256] + s1 : s1;
    }
    return;
  }
  const int eb = blockIdx.x - 16, c = tid & 63, e = eb * 64 + c, cq = tid >> 6;
  // this thread's 128 values of column e (rows cq, cq + 4, ...): fetched 32 at a time, before the query gradients are needed
  float acc[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) acc[i] = 0.f;
  for (int idx = tid; idx < 6 * M2_I; idx += M2_THREADS) dqs[idx] = idx < k * M2_I ? w.dQ[idx] : 0.f;
  __syncthreads();
#pragma unroll 1
  for (int c0 = 0; c0 < M2_I; c0 += 128) {
    float wv[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) wv[q] = wq[(int64_t)(c0 + cq + 4 * q) * M2_E + e];
#pragma unroll
    for (int q = 0; q < 32; ++q)
#pragma unroll
      for (int i = 0; i < 6; ++i) acc[i] += dqs[i * M2_I + c0 + cq + 4 * q] * wv[q];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) part[(cq * 6 + i) * 64 + c] = acc[i];
  __syncthreads();
  if (tid < 64) {
    float dw = 0.f, db = 0.f;
    for (int i = 0; i < k; ++i) {
      const float g = (part[(0 * 6 + i) * 64 + c] + part[(1 * 6 + i) * 64 + c]) + (part[(2 * 6 + i) * 64 + c] + part[(3 * 6 + i) * 64 + c]);
      const float xhat = (q_param[(int64_t)i * M2_E + e] - w.gmean[i]) * w.grstd[i];
      dw += g * xhat;
      db += g;
    }
    w.lnpart[(int64_t)w.T * 2 * M2_E + e] = dw;
    w.lnpart[(int64_t)w.T * 2 * M2_E + M2_E + e] = db;
  }
}

// ----------------------------------------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------------------------------------
int mca_out(hipStream_t st, const float* O, const float* wo, const float* bo, int k, int E, int I, float p, uint64_t seed0, const uint64_t* tick,
            float* z, const float* q, float* q_new, float mm);                                      // mca.hip
int reduce_parts2(hipStream_t st, const float* part0, const float* part1, int G, int W, int ld, float* out0, float* out1, int accumulate);   // rows.hip

int merge2_fwd(hipStream_t st, const mhimx_merge* m, const float* X, int64_t R, float* z, float* q_new, int update_q, void* ws, int64_t ws_bytes) {
  Arena ar(ws, ws_bytes);
  Merge2Ws w;
  merge2_ws_layout(ar, R, m->k, &w);
  MHIMX_CHECK_ARG(ar.ok(), "merge_fwd: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)ar.off);
  MHIMX_CHECK_ARG(!update_q || q_new, "merge_fwd: update_q needs q_new");
  const int k = (int)m->k, J = M2_H * k;
  const float scale = 1.0f / sqrtf((float)M2_DH);
  hipLaunchKernelGGL(merge2_prep_kernel, dim3(64), dim3(M2_THREADS), 0, st, m->q_param, m->ln_w, m->ln_b, m->wq, m->wkv, k, scale, w);
  MHIMX_LAUNCH_CHECK();
  MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)merge2_rows_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)M2_FWD_SMEM)));
  hipLaunchKernelGGL(merge2_rows_fwd_kernel, dim3((unsigned)w.T), dim3(M2_THREADS), M2_FWD_SMEM, st, X, m->x_rows, R, m->ln_w, m->ln_b, J, m->drop_p,
                     m->drop_seed, m->drop_tick, w);
  MHIMX_LAUNCH_CHECK();
  hipLaunchKernelGGL(merge2_partials_kernel<true>, dim3((unsigned)(J * 4)), dim3(M2_THREADS), 0, st, w.ypart, m->ln_w, m->ln_b, w.Y, w);
  MHIMX_LAUNCH_CHECK();
  hipLaunchKernelGGL(merge2_o_kernel, dim3(M2_H * 4), dim3(M2_THREADS), 0, st, m->wkv, k, w);
  MHIMX_LAUNCH_CHECK();
  return mca_out(st, w.O, m->wo, m->bo, k, M2_E, M2_I, m->drop_p, m->drop_seed + 0x9E3779B97F4A7C15ull, m->drop_tick, z, m->q_param,
                 update_q ? q_new : (float*)nullptr, m->mm);
}

int merge2_bwd(hipStream_t st, const mhimx_merge* m, const float* X, int64_t R, const float* dz, float* dX, const mhimx_merge_grad* gr, void* ws,
               int64_t ws_bytes) {
  Arena ar(ws, ws_bytes);
  Merge2Ws w;
  merge2_ws_layout(ar, R, m->k, &w);
  MHIMX_CHECK_ARG(ar.ok(), "merge_bwd: workspace too small");
  MHIMX_CHECK_ARG(m->wo_t && aligned16(m->wo_t), "merge_bwd: transposed to_out weight missing");
  const int k = (int)m->k, J = M2_H * k;
  const float scale = 1.0f / sqrtf((float)M2_DH);
  const int acc = gr->accumulate;
  const uint64_t oseed = m->drop_seed + 0x9E3779B97F4A7C15ull;
  hipLaunchKernelGGL(merge2_bwd_pre_kernel, dim3(64), dim3(M2_THREADS), 0, st, dz, m->wo_t, m->wkv, k, m->drop_p, oseed, m->drop_tick, gr->d_bo, acc, w);
  MHIMX_LAUNCH_CHECK();
  MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)merge2_rows_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)M2_BWD_SMEM)));
  hipLaunchKernelGGL(merge2_rows_bwd_kernel, dim3((unsigned)w.T), dim3(M2_THREADS), M2_BWD_SMEM, st, X, m->x_rows, R, m->ln_w, m->ln_b, J, m->drop_p,
                     m->drop_seed, m->drop_tick, dX, w);
  MHIMX_LAUNCH_CHECK();
  float* U = w.aq;                                         // (the fp32 copy of aq is not needed any more: its place takes U [J, E])
  hipLaunchKernelGGL(merge2_partials_kernel<false>, dim3((unsigned)(J * 4)), dim3(M2_THREADS), 0, st, w.upart, m->ln_w, m->ln_b, U, w);
  MHIMX_LAUNCH_CHECK();
  hipLaunchKernelGGL(merge2_grads1_kernel, dim3(48), dim3(M2_THREADS), 0, st, dz, U, m->wkv, k, scale, m->drop_p, oseed, m->drop_tick, gr->d_wkv,
                     gr->d_wo, acc, w);
  MHIMX_LAUNCH_CHECK();
  hipLaunchKernelGGL(merge2_grads2_kernel, dim3(24), dim3(M2_THREADS), 0, st, m->q_param, m->wq, k, gr->d_wq, acc, w);
  MHIMX_LAUNCH_CHECK();
  // d_ln_w / d_ln_b: T row-tile partials + one row from the queries
  if (gr->defer && gr->defer->n + 2 <= MHIMX_REDUCE_MAX) {
    defer_push(gr->defer, reduce_job_parts(w.lnpart, w.T + 1, M2_E, 2 * M2_E, gr->d_ln_w, acc));
    defer_push(gr->defer, reduce_job_parts(w.lnpart + M2_E, w.T + 1, M2_E, 2 * M2_E, gr->d_ln_b, acc));
    return 0;
  }
  return reduce_parts2(st, w.lnpart, w.lnpart + M2_E, w.T + 1, M2_E, 2 * M2_E, gr->d_ln_w, gr->d_ln_b, acc);
}

}  // namespace mhimx
