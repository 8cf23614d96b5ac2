// mca2_prep.hpp — the pieces of the projection-free Merge (mca2.hip) that the step's preparation launch (gemm_dma.hip) shares: workspace
// layout, fragment-image stores, and the parameter-only kernel body (gq = LN(q), Q, aq and its images).
#pragma once
#include <stdlib.h>

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
  unsigned* gate;            // arrivals of the backward tail's stage 2 (zeroed by the rows backward): stage 3 may share its launch
  // An instance-sharded bag (sharded.py, BASELINE c5): the row list x_rows holds BAG row ids, this shard owns [own_lo, own_lo + own_n) and
  // its X / dX hold only those rows (row id - own_lo).  Rows of the list outside the range take no part here (score -inf, zero gradient):
  // they are another shard's.  own_n == 0: every row is this process's (one GPU).
  int64_t own_lo, own_n;
  int T;                     // row tiles
  int rt;                    // rows per tile: 16 (R <= 4096: twice the workgroups on a pass that is a latency chain per tile) or 32
};

// a workgroup of a bag-batched launch (common.hpp: BagBatch) moves the workspace pointers to its own bag's copy
MHIMX_DEV void bag_move(Merge2Ws& w, const BagBatch& bb) {
  if (blockIdx.z == 0) return;
#define M2_MV(f) w.f = bag_ptr(w.f, bb)
  M2_MV(gq); M2_MV(gmean); M2_MV(grstd); M2_MV(Q); M2_MV(aq); M2_MV(aqf); M2_MV(gtf_aq); M2_MV(mean); M2_MV(rstd); M2_MV(S); M2_MV(pm); M2_MV(pl);
  M2_MV(psd); M2_MV(ypart); M2_MV(stats); M2_MV(Y); M2_MV(O); M2_MV(dO); M2_MV(dyf); M2_MV(gtf_dy); M2_MV(dpart); M2_MV(upart); M2_MV(lnpart);
  M2_MV(dQ); M2_MV(gate);
#undef M2_MV
}

// rows per tile of a Merge over R rows.  The row passes are per-tile latency chains on ceil(R / rt) CUs; half-size tiles halve the row loads,
// LayerNorm reductions and matrix-core steps of every wave and double the CUs at work (c2: 970 rows, 31 -> 61 workgroups).  Long row
// lists keep 32 rows (the per-tile pooled partials are [48, 512] floats each: twice the tiles = twice that traffic).
inline int m2_tile_rows(int64_t R) {
  static const int forced = getenv("MHIMX_MERGE_TILE") ? atoi(getenv("MHIMX_MERGE_TILE")) : 0;
  if (forced == 16 || forced == 32) return forced;
  return R <= 4096 ? 16 : 32;
}

inline int64_t merge2_ws_layout(Arena& ar, int64_t R, int64_t k, Merge2Ws* out) {
  Merge2Ws w;
  w.rt = m2_tile_rows(R);
  const int64_t T = cdiv(R, w.rt);
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
  w.lnpart = ar.take<float>(T * 2 * M2_E);
  w.dQ = ar.take<float>(k * M2_I);
  w.gate = reinterpret_cast<unsigned*>(ar.take<float>(64));
  w.own_lo = 0; w.own_n = 0;
  if (out) *out = w;
  return ar.off;
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

// out[i][d] = rows[d][:] . vec[i][:] for NR (16 or 64) consecutive weight rows (row pitch 512) and the 6 vectors vec[6][512] in LDS (rows >= k
// zero), on the matrix cores (3-term bf16): [6 -> 16 x 512] . [512 x 16 rows per wave] = 16 MFMA steps per wave and NO cross-lane
// reductions (the wave-per-row form spent most of its time in 96 DPP reductions per wave).  Every load of the wave's 16 rows is in flight
// before the first MFMA.  gout (optional): the same values to global [i][512].
// the weight rows of m2_head_dots as a register block: requested by m2_head_rows_load (as early as the caller can - ahead of the barrier
// that publishes `vec`), consumed by m2_head_dots_use
struct M2HeadRows { m2_f4 b0[16], b1[16]; };
template <int NR>
MHIMX_DEV void m2_head_rows_load(const float* __restrict__ rows, M2HeadRows& r) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (NR == 16 && wave != 0) return;
  const int n = lane & 15, kg = lane >> 4, wrow0 = NR == 16 ? 0 : wave * 16;
  const float* rp = rows + (int64_t)(wrow0 + n) * M2_E + kg * 8;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    r.b0[ks] = *reinterpret_cast<const m2_f4*>(rp + ks * 32);
    r.b1[ks] = *reinterpret_cast<const m2_f4*>(rp + ks * 32 + 4);
  }
}
template <int NR>
MHIMX_DEV void m2_head_dots_use(const M2HeadRows& r, const float* vec, int k, float* out, int out_ld, float* gout) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (NR == 16 && wave != 0) return;
  const int n = lane & 15, kg = lane >> 4, wrow0 = NR == 16 ? 0 : wave * 16;
  const bool am = n < 6;                                     // A row m = lane & 15: vector m (zero beyond the 6th)
  const float* ap = vec + (am ? n : 0) * M2_E + kg * 8;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    m2_f4 a0 = *reinterpret_cast<const m2_f4*>(ap + ks * 32), a1 = *reinterpret_cast<const m2_f4*>(ap + ks * 32 + 4);
    if (!am) { a0 = m2_f4{0.f, 0.f, 0.f, 0.f}; a1 = a0; }
    const float av[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    const float bv[8] = {r.b0[ks][0], r.b0[ks][1], r.b0[ks][2], r.b0[ks][3], r.b1[ks][0], r.b1[ks][1], r.b1[ks][2], r.b1[ks][3]};
    bf8 ah, al, bh, bl;
    m2_split8(av, ah, al);
    m2_split8(bv, bh, bl);
    acc = m2_mfma3(ah, al, bh, bl, acc);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = 4 * kg + i;                                // C: row m = vector, column n = weight row
    if (m < k) {
      if (out) out[m * out_ld + wrow0 + n] = acc[i];
      if (gout) gout[m * M2_I + wrow0 + n] = acc[i];
    }
  }
}
template <int NR>
MHIMX_DEV void m2_head_dots(const float* __restrict__ rows, const float* vec, int k, float* out, int out_ld, float* gout) {
  M2HeadRows r;
  m2_head_rows_load<NR>(rows, r);
  m2_head_dots_use<NR>(r, vec, k, out, out_ld, gout);
}
// zero the rows k..5 of a [6][512] LDS block (so that loops over the queries can be a compile-time 6)
MHIMX_DEV void m2_zero_tail(float* v, int k) {
  for (int idx = k * M2_E + threadIdx.x; idx < 6 * M2_E; idx += M2_THREADS) v[idx] = 0.f;
}

// ----------------------------------------------------------------------------------------------------------------------
// 1. parameters: gq = LN(q), Q = gq Wq^T, aq[(h,i),:] = scale sum_d Q[i,h,d] Wk[h*64+d,:] and its two fragment images.
//    grid = 8 heads x 8 column blocks of 64.
// ----------------------------------------------------------------------------------------------------------------------
// (a device function: it is the body of merge2_prep_kernel (mca2.hip) and of job kind 6 of the step's ONE preparation launch
// (prep_batch_kernel, gemm_dma.hip), where it runs beside the other parameter-only jobs instead of on the student's chain)
// lds: 6 * 512 + 6 * 64 floats, 16-byte aligned (the caller's: a rider must not add static LDS to its host kernel)
MHIMX_DEV void merge2_prep_body(int block, float* lds, const float* __restrict__ q_param, const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                const float* __restrict__ wq, const float* __restrict__ wkv, int k, float scale, const Merge2Ws& w) {
  float* gqs = lds;                 // [6][512]
  float* qh = gqs + 6 * M2_E;       // [6][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = block >> 3, eb = block & 7;
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
    if (block == 0) {
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


struct Merge2PrepArgs { const float *q_param, *ln_w, *ln_b, *wq, *wkv; int k; float scale; Merge2Ws w; };

}  // namespace mhimx
