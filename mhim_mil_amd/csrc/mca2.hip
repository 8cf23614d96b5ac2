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
  // Q of this head: one wave per output column d, all k queries against the same weight row
  for (int d = wave; d < 64; d += 4) {
    const float* row = wq + (int64_t)(h * 64 + d) * M2_E;
    const m2_f4 a = *reinterpret_cast<const m2_f4*>(row + 4 * lane), b = *reinterpret_cast<const m2_f4*>(row + 256 + 4 * lane);
    for (int i = 0; i < k; ++i) {
      const m2_f4 ga = *reinterpret_cast<const m2_f4*>(gqs + i * M2_E + 4 * lane), gb = *reinterpret_cast<const m2_f4*>(gqs + i * M2_E + 256 + 4 * lane);
      float s = a[0] * ga[0] + a[1] * ga[1] + a[2] * ga[2] + a[3] * ga[3] + b[0] * gb[0] + b[1] * gb[1] + b[2] * gb[2] + b[3] * gb[3];
      s = wave_sum(s);
      if (lane == 0) {
        qh[i * 64 + d] = s;
        if (eb == 0) w.Q[i * M2_I + h * 64 + d] = s;
      }
    }
  }
  __syncthreads();
  // aq for the 64 columns of this block (rows of Wk are read as 256-byte segments), then the images
  const int c = tid & 63, e = eb * 64 + c;
  for (int i = tid >> 6; i < k; i += 4) {
    float acc = 0.f;
#pragma unroll 8
    for (int d = 0; d < 64; ++d) acc += qh[i * 64 + d] * wkv[(int64_t)(h * 64 + d) * M2_E + e];
    acc *= scale;
    const int j = h * k + i;
    w.aq[j * M2_E + e] = acc;
    m2_store_images(w.aqf, w.gtf_aq, j, e, acc);
  }
  for (int j = J + h; j < M2_JK; j += M2_H)                 // zero padding slots (this head's share), 4 threads per column
    if ((tid >> 6) == ((j - J) >> 3) % 4) {
      if (j < M2_JP) w.aq[j * M2_E + e] = 0.f;
      m2_store_images(w.aqf, w.gtf_aq, j, e, 0.f);
    }
}

// ----------------------------------------------------------------------------------------------------------------------
// shared pieces of the two row kernels
// ----------------------------------------------------------------------------------------------------------------------
// rows of the tile -> xhat = (x - mean) rstd in LDS [32][516]; `have_stats`: mean / rstd are read instead of computed
template <bool HAVE_STATS>
MHIMX_DEV void m2_load_rows(const float* __restrict__ X, const int64_t* __restrict__ xrows, int64_t R, int64_t row0, float* xh, float* mean,
                            float* rstd) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  m2_f4 a[8], b[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {                               // all 16 loads of this wave's rows in flight
    const int64_t n = row0 + wave + 4 * q;
    const int64_t nc = n < R ? n : R - 1;
    const float* src = X + (xrows ? xrows[nc] : nc) * M2_E;
    a[q] = *reinterpret_cast<const m2_f4*>(src + 4 * lane);
    b[q] = *reinterpret_cast<const m2_f4*>(src + 256 + 4 * lane);
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int rr = wave + 4 * q;
    const int64_t n = row0 + rr;
    float mu, rs;
    if (HAVE_STATS) {
      mu = n < R ? mean[n] : 0.f;
      rs = n < R ? rstd[n] : 0.f;
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

// red[wave][32][48] += (xhat w + b)[32 x 512] . img^T over this wave's quarter of the 512-deep reduction (3-term bf16)
MHIMX_DEV void m2_rows_times_slots(const float* xh, const float* __restrict__ ln_w, const float* __restrict__ ln_b, const float* __restrict__ img,
                                   float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kg = lane >> 4;
  f32x4 acc[2][3];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) acc[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int ks = 4 * wave; ks < 4 * wave + 4; ++ks) {
    const int e0 = ks * 32 + kg * 8;
    const m2_f4 w0 = *reinterpret_cast<const m2_f4*>(ln_w + e0), w1 = *reinterpret_cast<const m2_f4*>(ln_w + e0 + 4);
    const m2_f4 b0 = *reinterpret_cast<const m2_f4*>(ln_b + e0), b1 = *reinterpret_cast<const m2_f4*>(ln_b + e0 + 4);
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
      bf8 bh, bl;
      m2_load_frag(img, nb * 16 + ks, lane, bh, bl);
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
constexpr size_t M2_FWD_SMEM = (size_t)(M2_ROWS * M2_XLD + 4 * M2_ROWS * M2_JP + M2_JP * M2_PLD) * sizeof(float);

__global__ __launch_bounds__(M2_THREADS) void merge2_rows_fwd_kernel(const float* __restrict__ X, const int64_t* __restrict__ xrows, int64_t R,
                                                                    const float* __restrict__ ln_w, const float* __restrict__ ln_b, int J,
                                                                    float drop_p, uint64_t seed0, const uint64_t* __restrict__ tick, Merge2Ws w) {
  extern __shared__ __attribute__((aligned(16))) float m2sm[];
  float* xh = m2sm;                                  // [32][516]
  float* red = xh + M2_ROWS * M2_XLD;                // [4][32][48]
  float* pdT = red + 4 * M2_ROWS * M2_JP;            // [48][36]
  const int tid = threadIdx.x;
  const int t = blockIdx.x;
  const int64_t row0 = (int64_t)t * M2_ROWS;
  m2_load_rows<false>(X, xrows, R, row0, xh, w.mean, w.rstd);
  __syncthreads();
  m2_rows_times_slots(xh, ln_w, ln_b, w.aqf, red);
  __syncthreads();
  for (int idx = tid; idx < M2_ROWS * M2_JP; idx += M2_THREADS) {
    const float s = (red[idx] + red[M2_ROWS * M2_JP + idx]) + (red[2 * M2_ROWS * M2_JP + idx] + red[3 * M2_ROWS * M2_JP + idx]);
    red[idx] = s;
    const int r = idx / M2_JP;
    if (row0 + r < R) w.S[(row0 + r) * M2_JP + (idx - r * M2_JP)] = s;
  }
  __syncthreads();
  if (tid < M2_JP) {
    const int j = tid;
    const int nv = (int)((R - row0) < M2_ROWS ? (R - row0) : M2_ROWS);
    float m = -INFINITY;
    if (j < J)
      for (int r = 0; r < nv; ++r) m = fmaxf(m, red[r * M2_JP + j]);
    const uint64_t seed = drop_p > 0.f ? eff_seed(seed0, tick) : 0;
    const float ks = 1.f / (1.f - drop_p);
    float l = 0.f, sd = 0.f;
    for (int r = 0; r < M2_ROWS; ++r) {
      float p = 0.f, pd = 0.f;
      if (j < J && r < nv) {
        p = __expf(red[r * M2_JP + j] - m);
        pd = (drop_p > 0.f && !m2_keep(seed, j, row0 + r, drop_p)) ? 0.f : p * ks;
      }
      l += p;
      sd += pd;
      pdT[j * M2_PLD + r] = pd;
    }
    w.pm[t * M2_JP + j] = m;
    w.pl[t * M2_JP + j] = l;
    w.psd[t * M2_JP + j] = sd;
  }
  __syncthreads();
  m2_pool_rows(pdT, xh, w.ypart + (int64_t)t * M2_JP * M2_E);
}

// ----------------------------------------------------------------------------------------------------------------------
// 3. merge the tile partials of one head's slots (online softmax, fixed order), Y = Yh w + (sum Pd) b, O = Wv_h Y.
//    grid = 8 heads x 2 halves of the head's 64 output columns.
// ----------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(M2_THREADS) void merge2_fin_kernel(const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                               const float* __restrict__ wkv, int k, Merge2Ws w) {
  __shared__ __attribute__((aligned(16))) float ys[6 * M2_E];
  __shared__ float wt[1024];
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x >> 1, half = blockIdx.x & 1;
  const int T = w.T;
  for (int i = 0; i < k; ++i) {
    const int j = h * k + i;
    float m = -INFINITY;
    for (int t = tid; t < T; t += M2_THREADS) m = fmaxf(m, w.pm[t * M2_JP + j]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    const float M = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float L = 0.f, SD = 0.f, y0 = 0.f, y1 = 0.f;
    for (int t0 = 0; t0 < T; t0 += 1024) {
      const int nt = T - t0 < 1024 ? T - t0 : 1024;
      for (int t = tid; t < nt; t += M2_THREADS) wt[t] = __expf(w.pm[(t0 + t) * M2_JP + j] - M);
      __syncthreads();
      if (tid == 0) {                                   // fixed order: deterministic
        float l = 0.f, s = 0.f;
        for (int t = 0; t < nt; ++t) { l += w.pl[(t0 + t) * M2_JP + j] * wt[t]; s += w.psd[(t0 + t) * M2_JP + j] * wt[t]; }
        red[4] = l;
        red[5] = s;
      }
      const float* yp = w.ypart + ((int64_t)t0 * M2_JP + j) * M2_E;
#pragma unroll 4
      for (int t = 0; t < nt; ++t) {
        y0 += yp[(int64_t)t * M2_JP * M2_E + tid] * wt[t];
        y1 += yp[(int64_t)t * M2_JP * M2_E + tid + 256] * wt[t];
      }
      __syncthreads();
      L += red[4];
      SD += red[5];
      __syncthreads();
    }
    const float inv = 1.f / L;
    const float v0 = y0 * inv * ln_w[tid] + SD * inv * ln_b[tid], v1 = y1 * inv * ln_w[tid + 256] + SD * inv * ln_b[tid + 256];
    ys[i * M2_E + tid] = v0;
    ys[i * M2_E + tid + 256] = v1;
    if (half == 0) {
      w.Y[j * M2_E + tid] = v0;
      w.Y[j * M2_E + tid + 256] = v1;
      if (tid == 0) { w.stats[2 * j] = M; w.stats[2 * j + 1] = L; }
    }
  }
  __syncthreads();
  for (int d = half * 32 + wave; d < half * 32 + 32; d += 4) {
    const float* row = wkv + (int64_t)(M2_I + h * 64 + d) * M2_E;          // the V half of to_kv
    const m2_f4 a = *reinterpret_cast<const m2_f4*>(row + 4 * lane), b = *reinterpret_cast<const m2_f4*>(row + 256 + 4 * lane);
    for (int i = 0; i < k; ++i) {
      const m2_f4 ga = *reinterpret_cast<const m2_f4*>(ys + i * M2_E + 4 * lane), gb = *reinterpret_cast<const m2_f4*>(ys + i * M2_E + 256 + 4 * lane);
      float s = a[0] * ga[0] + a[1] * ga[1] + a[2] * ga[2] + a[3] * ga[3] + b[0] * gb[0] + b[1] * gb[1] + b[2] * gb[2] + b[3] * gb[3];
      s = wave_sum(s);
      if (lane == 0) w.O[i * M2_I + h * 64 + d] = s;
    }
  }
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
  const uint64_t seed = drop_p > 0.f ? eff_seed(seed0, tick) : 0;
  const float ks = 1.f / (1.f - drop_p);
  for (int idx = tid; idx < k * M2_E; idx += M2_THREADS) {
    const int i = idx >> 9, e = idx & 511;
    float v = dz[idx];
    if (drop_p > 0.f) v = drop_keep(seed, (uint64_t)i, (uint32_t)e, drop_p) ? v * ks : 0.f;
    dzs[idx] = v;
  }
  __syncthreads();
  if (h == 0 && tid < 64) {
    const int e = eb * 64 + tid;
    float s = 0.f;
    for (int i = 0; i < k; ++i) s += dzs[i * M2_E + e];
    d_bo[e] = accumulate ? d_bo[e] + s : s;
  }
  for (int d = wave; d < 64; d += 4) {
    const float* row = wo_t + (int64_t)(h * 64 + d) * M2_E;
    const m2_f4 a = *reinterpret_cast<const m2_f4*>(row + 4 * lane), b = *reinterpret_cast<const m2_f4*>(row + 256 + 4 * lane);
    for (int i = 0; i < k; ++i) {
      const m2_f4 ga = *reinterpret_cast<const m2_f4*>(dzs + i * M2_E + 4 * lane), gb = *reinterpret_cast<const m2_f4*>(dzs + i * M2_E + 256 + 4 * lane);
      float s = a[0] * ga[0] + a[1] * ga[1] + a[2] * ga[2] + a[3] * ga[3] + b[0] * gb[0] + b[1] * gb[1] + b[2] * gb[2] + b[3] * gb[3];
      s = wave_sum(s);
      if (lane == 0) {
        doh[i * 64 + d] = s;
        if (eb == 0) w.dO[i * M2_I + h * 64 + d] = s;
      }
    }
  }
  __syncthreads();
  const int c = tid & 63, e = eb * 64 + c;
  for (int i = tid >> 6; i < k; i += 4) {
    float acc = 0.f;
#pragma unroll 8
    for (int d = 0; d < 64; ++d) acc += doh[i * 64 + d] * wkv[(int64_t)(M2_I + h * 64 + d) * M2_E + e];
    const int j = h * k + i;
    dys[i * 64 + c] = acc;
    m2_store_images(w.dyf, w.gtf_dy, j, e, acc);
  }
  for (int j = J + h; j < M2_JK; j += M2_H)
    if ((tid >> 6) == ((j - J) >> 3) % 4) m2_store_images(w.dyf, w.gtf_dy, j, e, 0.f);
  __syncthreads();
  for (int i = wave; i < k; i += 4) {
    const int j = h * k + i;
    const float s = wave_sum(dys[i * 64 + lane] * w.Y[j * M2_E + eb * 64 + lane]);
    if (lane == 0) w.dpart[j * 8 + eb] = s;
  }
  if (h == 0 && eb == 0)
    for (int j = J + tid; j < M2_JP; j += M2_THREADS)
      for (int q = 0; q < 8; ++q) w.dpart[j * 8 + q] = 0.f;
}

// ----------------------------------------------------------------------------------------------------------------------
// 5. rows backward: dPd = xn dY^T, softmax backward, dxn = ds aq + Pd dY, LayerNorm backward (dX scattered to the rows' places,
//    per-tile d_ln_w / d_ln_b partials), pooled U partials.   grid = ceil(R / 32)
// ----------------------------------------------------------------------------------------------------------------------
constexpr size_t M2_BWD_SMEM = (size_t)(2 * M2_ROWS * M2_XLD + M2_ROWS * M2_CLD + M2_JP * M2_PLD) * sizeof(float);
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
  float* lnred = cf;                                  // [4][2][512]: the coefficient tile is in registers by then
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kg = lane >> 4;
  const int t = blockIdx.x;
  const int64_t row0 = (int64_t)t * M2_ROWS;
  m2_load_rows<true>(X, xrows, R, row0, xh, w.mean, w.rstd);
  __syncthreads();
  m2_rows_times_slots(xh, ln_w, ln_b, w.dyf, dxs);
  __syncthreads();
  // ---- softmax backward per (row, slot)
  {
    const uint64_t seed = drop_p > 0.f ? eff_seed(seed0, tick) : 0;
    const float ks = 1.f / (1.f - drop_p);
    for (int idx = tid; idx < M2_ROWS * M2_JK; idx += M2_THREADS) {
      const int r = idx >> 6, j = idx & 63;
      float ds = 0.f, pd = 0.f;
      if (j < J && row0 + r < R) {
        const int q = r * M2_JP + j;
        const float dpd = (dxs[q] + dxs[M2_ROWS * M2_JP + q]) + (dxs[2 * M2_ROWS * M2_JP + q] + dxs[3 * M2_ROWS * M2_JP + q]);
        const float p = __expf(w.S[(row0 + r) * M2_JP + j] - w.stats[2 * j]) / w.stats[2 * j + 1];
        const float* dp8 = w.dpart + j * 8;
        const float delta = ((dp8[0] + dp8[1]) + (dp8[2] + dp8[3])) + ((dp8[4] + dp8[5]) + (dp8[6] + dp8[7]));
        const float kf = (drop_p > 0.f && !m2_keep(seed, j, row0 + r, drop_p)) ? 0.f : ks;
        pd = p * kf;
        ds = p * (dpd * kf - delta);
      }
      cf[r * M2_CLD + j] = ds;
      cf[r * M2_CLD + M2_JK + j] = pd;
      if (j < M2_JP) dsT[j * M2_PLD + r] = ds;
    }
  }
  __syncthreads();
  // ---- dxn[32 x 512] = cf[32 x 128] . [aq ; dY]  (K = 128 = 4 steps of 32; B from the transposed fragment images)
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
    __syncthreads();                                  // (every wave has read its dPd partials: dxs is free)
#pragma unroll 1
    for (int eb = 8 * wave; eb < 8 * wave + 8; ++eb) {
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf8 bh, bl;
        m2_load_frag(ks < 2 ? w.gtf_aq : w.gtf_dy, eb * 2 + (ks & 1), lane, bh, bl);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) acc[rb] = m2_mfma3(ah[rb][ks], al[rb][ks], bh, bl, acc[rb]);
      }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int i = 0; i < 4; ++i) dxs[(rb * 16 + 4 * kg + i) * M2_XLD + eb * 16 + r16] = acc[rb][i];
    }
  }
  __syncthreads();
  // ---- LayerNorm backward, 8 rows per wave: dxhat = dxn w;  dx = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat))
  {
    const m2_f4 wa = *reinterpret_cast<const m2_f4*>(ln_w + 4 * lane), wb = *reinterpret_cast<const m2_f4*>(ln_w + 256 + 4 * lane);
    m2_f4 dwa = m2_f4{0.f, 0.f, 0.f, 0.f}, dwb = dwa, dba = dwa, dbb = dwa;
#pragma unroll 2
    for (int q = 0; q < 8; ++q) {
      const int rr = wave + 4 * q;
      const int64_t n = row0 + rr;
      if (n >= R) continue;
      const m2_f4 ga = *reinterpret_cast<const m2_f4*>(dxs + rr * M2_XLD + 4 * lane), gb = *reinterpret_cast<const m2_f4*>(dxs + rr * M2_XLD + 256 + 4 * lane);
      const m2_f4 xa = *reinterpret_cast<const m2_f4*>(xh + rr * M2_XLD + 4 * lane), xb = *reinterpret_cast<const m2_f4*>(xh + rr * M2_XLD + 256 + 4 * lane);
      dwa += ga * xa; dwb += gb * xb; dba += ga; dbb += gb;
      const m2_f4 ha = ga * wa, hb = gb * wb;
      float s1 = (ha[0] + ha[1]) + (ha[2] + ha[3]) + (hb[0] + hb[1]) + (hb[2] + hb[3]);
      const m2_f4 pa = ha * xa, pb = hb * xb;
      float s2 = (pa[0] + pa[1]) + (pa[2] + pa[3]) + (pb[0] + pb[1]) + (pb[2] + pb[3]);
      s1 = wave_sum(s1) * (1.f / M2_E);
      s2 = wave_sum(s2) * (1.f / M2_E);
      const float rs = w.rstd[n];
      float* dst = dX + (xrows ? xrows[n] : n) * M2_E;
      *reinterpret_cast<m2_f4*>(dst + 4 * lane) = (ha - s1 - xa * s2) * rs;
      *reinterpret_cast<m2_f4*>(dst + 256 + 4 * lane) = (hb - s1 - xb * s2) * rs;
    }
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
// 6. rank-k gradients, first launch.   blocks 0..15: (head, half): U = (sum_t upart) w, dQ = scale Wk U, d_wkv rows of this half
//    (K part: scale Q (x) U, V part: dO (x) Y);   blocks 16..31: 32 rows of d_wo = dz0^T O.
// ----------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(M2_THREADS) void merge2_grads1_kernel(const float* __restrict__ dz, const float* __restrict__ ln_w,
                                                                  const float* __restrict__ wkv, int k, float scale, float drop_p, uint64_t seed0,
                                                                  const uint64_t* __restrict__ tick, float* __restrict__ d_wkv,
                                                                  float* __restrict__ d_wo, int accumulate, Merge2Ws w) {
  __shared__ __attribute__((aligned(16))) float us[6 * M2_E];
  __shared__ __attribute__((aligned(16))) float ysh[6 * M2_E];
  __shared__ float qd[2 * 6 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (blockIdx.x >= 16) {
    const int e0 = (blockIdx.x - 16) * 32;
    const uint64_t seed = drop_p > 0.f ? eff_seed(seed0, tick) : 0;
    const float ks = 1.f / (1.f - drop_p);
    float* dzr = us;                                     // [k][32] dz0 of these rows
    for (int idx = tid; idx < k * 32; idx += M2_THREADS) {
      const int i = idx >> 5, e = e0 + (idx & 31);
      float v = dz[i * M2_E + e];
      if (drop_p > 0.f) v = drop_keep(seed, (uint64_t)i, (uint32_t)e, drop_p) ? v * ks : 0.f;
      dzr[idx] = v;
    }
    for (int idx = tid; idx < k * M2_I; idx += M2_THREADS) ysh[idx] = w.O[idx];
    __syncthreads();
    for (int idx = tid; idx < 32 * M2_I; idx += M2_THREADS) {
      const int r = idx >> 9, c = idx & 511;
      float s = 0.f;
      for (int i = 0; i < k; ++i) s += dzr[i * 32 + r] * ysh[i * M2_I + c];
      float* o = d_wo + (int64_t)(e0 + r) * M2_I + c;
      *o = accumulate ? *o + s : s;
    }
    return;
  }
  const int h = blockIdx.x >> 1, half = blockIdx.x & 1;
  const int T = w.T;
  for (int i = 0; i < k; ++i) {
    const int j = h * k + i;
    const float* up = w.upart + (int64_t)j * M2_E;
    float u0 = 0.f, u1 = 0.f;
#pragma unroll 4
    for (int t = 0; t < T; ++t) {                        // fixed order: deterministic
      u0 += up[(int64_t)t * M2_JP * M2_E + tid];
      u1 += up[(int64_t)t * M2_JP * M2_E + tid + 256];
    }
    us[i * M2_E + tid] = u0 * ln_w[tid];
    us[i * M2_E + tid + 256] = u1 * ln_w[tid + 256];
    ysh[i * M2_E + tid] = w.Y[j * M2_E + tid];
    ysh[i * M2_E + tid + 256] = w.Y[j * M2_E + tid + 256];
  }
  for (int idx = tid; idx < k * 32; idx += M2_THREADS) {
    const int i = idx >> 5, d = half * 32 + (idx & 31);
    qd[idx] = scale * w.Q[i * M2_I + h * 64 + d];
    qd[6 * 32 + idx] = w.dO[i * M2_I + h * 64 + d];
  }
  __syncthreads();
  for (int d = half * 32 + wave; d < half * 32 + 32; d += 4) {
    const float* row = wkv + (int64_t)(h * 64 + d) * M2_E;            // the K half of to_kv
    const m2_f4 a = *reinterpret_cast<const m2_f4*>(row + 4 * lane), b = *reinterpret_cast<const m2_f4*>(row + 256 + 4 * lane);
    for (int i = 0; i < k; ++i) {
      const m2_f4 ga = *reinterpret_cast<const m2_f4*>(us + i * M2_E + 4 * lane), gb = *reinterpret_cast<const m2_f4*>(us + i * M2_E + 256 + 4 * lane);
      float s = a[0] * ga[0] + a[1] * ga[1] + a[2] * ga[2] + a[3] * ga[3] + b[0] * gb[0] + b[1] * gb[1] + b[2] * gb[2] + b[3] * gb[3];
      s = wave_sum(s);
      if (lane == 0) w.dQ[i * M2_I + h * 64 + d] = scale * s;
    }
  }
  for (int idx = tid; idx < 32 * M2_E; idx += M2_THREADS) {
    const int dl = idx >> 9, e = idx & 511;
    float sk = 0.f, sv = 0.f;
    for (int i = 0; i < k; ++i) {
      sk += qd[i * 32 + dl] * us[i * M2_E + e];
      sv += qd[6 * 32 + i * 32 + dl] * ysh[i * M2_E + e];
    }
    float* ok = d_wkv + (int64_t)(h * 64 + half * 32 + dl) * M2_E + e;
    float* ov = d_wkv + (int64_t)(M2_I + h * 64 + half * 32 + dl) * M2_E + e;
    *ok = accumulate ? *ok + sk : sk;
    *ov = accumulate ? *ov + sv : sv;
  }
}

// ----------------------------------------------------------------------------------------------------------------------
// 7. rank-k gradients, second launch (needs all of dQ).   blocks 0..15: 32 rows of d_wq = dQ^T gq;   blocks 16..23: 64 columns of
//    dgq = dQ Wq and their LayerNorm-parameter gradients (the queries themselves are not trained) -> partial row T of lnpart.
// ----------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(M2_THREADS) void merge2_grads2_kernel(const float* __restrict__ q_param, const float* __restrict__ wq, int k,
                                                                  float* __restrict__ d_wq, int accumulate, Merge2Ws w) {
  __shared__ __attribute__((aligned(16))) float gqs[6 * M2_E];
  __shared__ float dqs[6 * M2_I];
  __shared__ float part[4 * 6 * 64];
  const int tid = threadIdx.x;
  for (int idx = tid; idx < k * M2_I; idx += M2_THREADS) dqs[idx] = w.dQ[idx];
  if (blockIdx.x < 16) {
    for (int idx = tid; idx < k * M2_E; idx += M2_THREADS) gqs[idx] = w.gq[idx];
    __syncthreads();
    const int c0 = blockIdx.x * 32;
    for (int idx = tid; idx < 32 * M2_E; idx += M2_THREADS) {
      const int r = idx >> 9, e = idx & 511;
      float s = 0.f;
      for (int i = 0; i < k; ++i) s += dqs[i * M2_I + c0 + r] * gqs[i * M2_E + e];
      float* o = d_wq + (int64_t)(c0 + r) * M2_E + e;
      *o = accumulate ? *o + s : s;
    }
    return;
  }
  __syncthreads();
  const int eb = blockIdx.x - 16, c = tid & 63, e = eb * 64 + c, cq = tid >> 6;
  float acc[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) acc[i] = 0.f;
  for (int cc = cq; cc < M2_I; cc += 4) {                  // rows of Wq as 256-byte segments, 4 row groups
    const float wv = wq[(int64_t)cc * M2_E + e];
#pragma unroll
    for (int i = 0; i < 6; ++i)
      if (i < k) acc[i] += dqs[i * M2_I + cc] * wv;
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
  hipLaunchKernelGGL(merge2_fin_kernel, dim3(16), dim3(M2_THREADS), 0, st, m->ln_w, m->ln_b, m->wkv, k, w);
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
  hipLaunchKernelGGL(merge2_grads1_kernel, dim3(32), dim3(M2_THREADS), 0, st, dz, m->ln_w, m->wkv, k, scale, m->drop_p, oseed, m->drop_tick, gr->d_wkv,
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
