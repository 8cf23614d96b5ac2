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
// Built for E = 512, 8 heads x 64, k <= 6 (J <= 48), R <= 32768 (the merge of the row-tile partials walks chunks of 256 tiles); other
// shapes take mca.hip's general path.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "mca2_side.hpp"

namespace mhimx {

int64_t merge2_part_floats() { return M2_PART_FLOATS; }
int64_t merge2_ws_bytes(int64_t R, int64_t k) {
  Arena ar(nullptr, 0);
  return merge2_ws_layout(ar, R, k, nullptr);
}

bool merge2_ok(const mhimx_merge* m, int64_t R) {
  return m->E == M2_E && m->heads == M2_H && m->dim_head == M2_DH && m->k >= 1 && m->heads * m->k <= M2_JP && R >= 1 && R <= 32768 &&
         m->prec != MHIMX_PREC_F32 && aligned16(m->wq) && aligned16(m->wkv) && aligned16(m->wo) && aligned16(m->ln_w) && aligned16(m->ln_b) &&
         aligned16(m->q_param);
}

// ----------------------------------------------------------------------------------------------------------------------
// 1. parameters (mca2_prep.hpp): standalone launch; a trainer runs the same body as a job of its preparation launch instead
// ----------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(M2_THREADS) void merge2_prep_kernel(Merge2PrepArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[6 * M2_E + 6 * 64];
  merge2_prep_body((int)blockIdx.x, lds, a.q_param, a.ln_w, a.ln_b, a.wq, a.wkv, a.k, a.scale, a.w);
}

// ----------------------------------------------------------------------------------------------------------------------
// shared pieces of the two row kernels
// ----------------------------------------------------------------------------------------------------------------------
// rows of the tile -> xhat = (x - mean) rstd in LDS [32][516]; HAVE_STATS: mean / rstd are read instead of computed.
// Also stages the LayerNorm weight and bias in LDS (lnw[512], lnb[512]).
// ok[32] (LDS): 1.f for the rows of the tile that take part - inside the list AND, for an instance-sharded bag (Merge2Ws.own_*), owned by
// this shard; the others are loaded as zeros (their source is clamped to a valid row).  Returns nothing; a tile without any such row is
// detected by the caller (m2_tile_dead) before this is called.
MHIMX_DEV bool m2_row_ok(const int64_t* __restrict__ xrows, int64_t R, int64_t n, const Merge2Ws& w, int64_t& src_row) {
  if (n >= R) { src_row = 0; return false; }
  const int64_t id = xrows ? xrows[n] : n;
  if (w.own_n > 0) {
    const bool own = id >= w.own_lo && id < w.own_lo + w.own_n;
    src_row = own ? id - w.own_lo : 0;
    return own;
  }
  src_row = id;
  return true;
}
// sharded bags only: true when no row of the tile is this shard's (every thread gets the same answer; flags: 4 ints of LDS)
MHIMX_DEV bool m2_tile_dead(const int64_t* __restrict__ xrows, int64_t R, int64_t row0, const Merge2Ws& w, int* flags) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int64_t dummy;
  const bool mine = lane < 8 && m2_row_ok(xrows, R, row0 + wave + 4 * lane, w, dummy);
  const bool any = __builtin_amdgcn_ballot_w64(mine) != 0;
  if (lane == 0) flags[wave] = any ? 1 : 0;
  __syncthreads();
  const bool dead = (flags[0] | flags[1] | flags[2] | flags[3]) == 0;
  __syncthreads();
  return dead;
}
template <bool HAVE_STATS>
MHIMX_DEV void m2_load_rows(const float* __restrict__ X, const int64_t* __restrict__ xrows, int64_t R, int64_t row0, float* xh, float* mean,
                            float* rstd, const float* __restrict__ ln_w, const float* __restrict__ ln_b, float* lnw, float* lnb, float* rs_tile,
                            const Merge2Ws& w, float* ok) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  m2_f4 a[8], b[8];
  float mu8[8], rs8[8];
  bool okq[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {                               // all 16 row loads of this wave in flight
    const int64_t n = row0 + wave + 4 * q;
    const int64_t nc = n < R ? n : R - 1;
    int64_t srow;
    okq[q] = m2_row_ok(xrows, R, n, w, srow);
    const float* src = X + srow * M2_E;
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
      if (lane == 0 && okq[q]) { mean[n] = mu; rstd[n] = rs; }
    }
    if (lane == 0) ok[rr] = okq[q] ? 1.f : 0.f;
    m2_f4 ya = (a[q] - mu) * rs, yb = (b[q] - mu) * rs;
    if (!okq[q]) { ya = m2_f4{0.f, 0.f, 0.f, 0.f}; yb = ya; }
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
constexpr size_t M2_FWD_SMEM = (size_t)(M2_ROWS * M2_XLD + 4 * M2_ROWS * M2_JP + M2_JP * M2_PLD + 2 * M2_E + M2_ROWS + 4) * sizeof(float);

__global__ __launch_bounds__(M2_THREADS) void merge2_rows_fwd_kernel(const float* __restrict__ X, const int64_t* __restrict__ xrows, int64_t R,
                                                                    const float* __restrict__ ln_w, const float* __restrict__ ln_b, int J,
                                                                    float drop_p, uint64_t seed0, const uint64_t* __restrict__ tick, Merge2Ws w) {
  extern __shared__ __attribute__((aligned(16))) float m2sm[];
  float* xh = m2sm;                                  // [32][516]
  float* red = xh + M2_ROWS * M2_XLD;                // [4][32][48]
  float* pdT = red + 4 * M2_ROWS * M2_JP;            // [48][36]
  float* lnw = pdT + M2_JP * M2_PLD;                 // [512]
  float* lnb = lnw + M2_E;                           // [512]
  float* ok = lnb + M2_E;                            // [32] 1 = the row takes part
  int* flags = reinterpret_cast<int*>(ok + M2_ROWS); // [4]
  const int tid = threadIdx.x;
  const int t = blockIdx.x;
  const int64_t row0 = (int64_t)t * M2_ROWS;
  if (w.own_n > 0 && m2_tile_dead(xrows, R, row0, w, flags)) {
    // an instance-sharded bag: no row of this tile is this shard's - an empty partial (weight 0 in every merge; its pooled rows are never read)
    if (tid < M2_JP) { w.pm[t * M2_JP + tid] = -INFINITY; w.pl[t * M2_JP + tid] = 0.f; w.psd[t * M2_JP + tid] = 0.f; }
    return;
  }
  M2Frags fr;
  m2_fetch_frags(w.aqf, fr);
  m2_load_rows<false>(X, xrows, R, row0, xh, w.mean, w.rstd, ln_w, ln_b, lnw, lnb, nullptr, w, ok);
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
  float sreg[8], m = -INFINITY;
  bool rv[8];
  if (rq < 4) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = rq * 8 + q;
      sreg[q] = red[r * M2_JP + j];
      rv[q] = ok[r] != 0.f;
      if (j < J && rv[q]) m = fmaxf(m, sreg[q]);
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
      if (j < J && rv[q]) {
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
template <int MODE>
__global__ __launch_bounds__(M2_THREADS) void merge2_partials_kernel(M2Parts in, int live_only, const float* __restrict__ ln_w,
                                                                    const float* __restrict__ ln_b, float* __restrict__ out,
                                                                    float* __restrict__ stats, float* __restrict__ raw_stats) {
  __shared__ float lds[M2_PARTIALS_LDS];
  merge2_partials_body<MODE>((int)blockIdx.x, lds, in, live_only != 0, ln_w, ln_b, out, stats, raw_stats);
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

// (Round 4, measured and NOT kept: 3a + 3b + mca_out as ONE launch - 8 head workgroups of 1024 threads: partial merge -> O_h -> the head's
// share of to_out as a [k, 512] partial, a ticket, the last head sums the 8 partials, applies bias / dropout / the EMA.  27 us in its first
// form (ten dependent memory round trips), 33 us with every load of a phase in flight (128 VGPRs, 12 spilled) against 4.7 + 7.8 + 9.2 us
// for the three launches: each of these kernels is a chain of 3-6 dependent round trips of ~1.5 us, a launch boundary costs ~3 us, and
// eight workgroups cannot hide what 160 + 32 + 128 do.  The step kept the three launches.)
// ----------------------------------------------------------------------------------------------------------------------
// 4. backward, parameters x dz: dz0 = dz keep/(1-p), d_bo, dO = dz0 Wo, dY[(h,i),:] = sum_d dO[i,h,d] Wv[h*64+d,:] (as the two
//    fragment images), delta partials dY.Y.   grid = 8 heads x 8 column blocks of 64.
// ----------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(M2_THREADS) void merge2_bwd_pre_kernel(const float* __restrict__ dz, const float* __restrict__ wo_t,
                                                                   const float* __restrict__ wkv, int k, float drop_p, uint64_t seed0,
                                                                   const uint64_t* __restrict__ tick, float* __restrict__ d_bo, int accumulate,
                                                                   Merge2Ws w, float rep) {
  __shared__ __attribute__((aligned(16))) float lds[M2_BWD_PRE_LDS];
  merge2_bwd_pre_body((int)blockIdx.x, lds, dz, wo_t, wkv, k, drop_p, seed0, tick, d_bo, accumulate, w, rep);      // (mca2_side.hpp)
}

// ----------------------------------------------------------------------------------------------------------------------
// 5. rows backward: dPd = xn dY^T, softmax backward, dxn = ds aq + Pd dY, LayerNorm backward (dX scattered to the rows' places,
//    per-tile d_ln_w / d_ln_b partials), pooled U partials.   grid = ceil(R / 32)
// ----------------------------------------------------------------------------------------------------------------------
constexpr size_t M2_BWD_SMEM = (size_t)(2 * M2_ROWS * M2_XLD + M2_ROWS * M2_CLD + M2_JP * M2_PLD + 2 * M2_E + 3 * M2_JK + 2 * M2_ROWS + 4) * sizeof(float);
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
  float* ok = rst + M2_ROWS;                          // [32] 1 = the row takes part
  int* flags = reinterpret_cast<int*>(ok + M2_ROWS);  // [4]
  float* lnred = cf;                                  // [4][2][512]: the coefficient tile is in registers by then
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kg = lane >> 4;
  const int t = blockIdx.x;
  const int64_t row0 = (int64_t)t * M2_ROWS;
  if (t == 0 && tid == 0) *w.gate = 0u;               // (the tail's stage-2 arrivals are counted from here: bag_wgrad_ws_kernel)
  // an instance-sharded bag: a tile without a row of this shard leaves no gradient and no partial (the merges of the pooled / LayerNorm
  // partials skip the tiles whose forward partial is empty: w.pl == 0)
  if (w.own_n > 0 && m2_tile_dead(xrows, R, row0, w, flags)) return;
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
  m2_load_rows<true>(X, xrows, R, row0, xh, w.mean, w.rstd, ln_w, ln_b, lnw, lnb, rst, w, ok);
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
      if (j < J && ok[r] != 0.f) {
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
      int64_t srow;
      dst_row[q] = m2_row_ok(xrows, R, row0 + wave + 4 * q, w, srow) ? srow : -1;      // (a shard's dX holds its own rows: id - own_lo)
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
__global__ __launch_bounds__(M2_THREADS) void merge2_grads1_kernel(Merge2Side a) {
  __shared__ __attribute__((aligned(16))) float lds[M2_GRADS1_LDS];
  merge2_grads1_body((int)blockIdx.x, lds, a);
}

// ----------------------------------------------------------------------------------------------------------------------
// 7. rank-k gradients, second launch (needs all of dQ).   blocks 0..15: 32 rows of d_wq = dQ^T gq;   blocks 16..23: 64 columns of
//    dgq = dQ Wq and their LayerNorm-parameter gradients (the queries themselves are not trained) -> partial row T of lnpart.
// ----------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(M2_THREADS) void merge2_grads2_kernel(Merge2Side a) {
  __shared__ __attribute__((aligned(16))) float lds[M2_GRADS2_LDS];
  merge2_grads2_body((int)blockIdx.x, lds, a);
}

// ----------------------------------------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------------------------------------
int mca_out(hipStream_t st, const float* O, const float* wo, const float* bo, int k, int E, int I, float p, uint64_t seed0, const uint64_t* tick,
            float* z, const float* q, float* q_new, float mm);                                      // mca.hip
int reduce_parts2(hipStream_t st, const float* part0, const float* part1, int G, int W, int ld, float* out0, float* out1, int accumulate);   // rows.hip

int merge2_side_launch(hipStream_t st, int stage, const Merge2Side& sd) {
  if (stage == 1) hipLaunchKernelGGL(merge2_partials_kernel<0>, dim3((unsigned)(sd.J * 4)), dim3(M2_THREADS), 0, st, m2_parts_tiles(sd.w, sd.w.upart),
                                     sd.w.own_n > 0 ? 1 : 0, sd.ln_w, sd.ln_b, const_cast<float*>(sd.U), (float*)nullptr, (float*)nullptr);
  else if (stage == 2) hipLaunchKernelGGL(merge2_grads1_kernel, dim3(M2_GRADS1_BLOCKS), dim3(M2_THREADS), 0, st, sd);
  else hipLaunchKernelGGL(merge2_grads2_kernel, dim3(M2_GRADS2_BLOCKS), dim3(M2_THREADS), 0, st, sd);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
// run every stage of a deferred tail that has not had its ride yet, as launches of its own (mhimx_reduce_flush, rows.hip)
int merge2_side_finish(hipStream_t st, mhimx_side_work* side, int upto_stage) {
  Merge2Side sd;
  while (side->pending != 0 && side->pending <= upto_stage) {
    memcpy(&sd, side->blob, sizeof(sd));
    if (int r = merge2_side_launch(st, side->pending, sd)) return r;
    side->pending = side->pending == 3 ? 0 : side->pending + 1;
  }
  return 0;
}

// the forward up to the row tiles' partials: parameters (unless prepared), rows pass.  An instance-sharded bag (m->own_n > 0): only the rows
// this shard owns take part.
static int merge2_fwd_rows(hipStream_t st, const mhimx_merge* m, const float* X, int64_t R, void* ws, int64_t ws_bytes, Merge2Ws* wout) {
  Arena ar(ws, ws_bytes);
  Merge2Ws w;
  merge2_ws_layout(ar, R, m->k, &w);
  MHIMX_CHECK_ARG(ar.ok(), "merge_fwd: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)ar.off);
  MHIMX_CHECK_ARG(m->own_n >= 0 && (m->own_n == 0 || m->x_rows), "merge_fwd: a shard's row range needs the bag-level row list (x_rows)");
  w.own_lo = m->own_n > 0 ? m->own_lo : 0;
  w.own_n = m->own_n;
  const int k = (int)m->k, J = M2_H * k;
  const float scale = 1.0f / sqrtf((float)M2_DH);
  if (!m->prepared) {
    hipLaunchKernelGGL(merge2_prep_kernel, dim3(64), dim3(M2_THREADS), 0, st, Merge2PrepArgs{m->q_param, m->ln_w, m->ln_b, m->wq, m->wkv, k, scale, w});
    MHIMX_LAUNCH_CHECK();
  }
  MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)merge2_rows_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)M2_FWD_SMEM)));
  hipLaunchKernelGGL(merge2_rows_fwd_kernel, dim3((unsigned)w.T), dim3(M2_THREADS), M2_FWD_SMEM, st, X, m->x_rows, R, m->ln_w, m->ln_b, J, m->drop_p,
                     m->drop_seed, m->drop_tick, w);
  MHIMX_LAUNCH_CHECK();
  *wout = w;
  return 0;
}
// from the merged pooled rows Y (and the softmax statistics) to the tokens: O = Wv Y, to_out, dropout, the queries' EMA
static int merge2_fwd_tail(hipStream_t st, const mhimx_merge* m, float* z, float* q_new, int update_q, const Merge2Ws& w) {
  const int k = (int)m->k;
  hipLaunchKernelGGL(merge2_o_kernel, dim3(M2_H * 4), dim3(M2_THREADS), 0, st, m->wkv, k, w);
  MHIMX_LAUNCH_CHECK();
  return mca_out(st, w.O, m->wo, m->bo, k, M2_E, M2_I, m->drop_p, m->drop_seed + 0x9E3779B97F4A7C15ull, m->drop_tick, z, m->q_param,
                 update_q ? q_new : (float*)nullptr, m->mm);
}

int merge2_fwd(hipStream_t st, const mhimx_merge* m, const float* X, int64_t R, float* z, float* q_new, int update_q, void* ws, int64_t ws_bytes) {
  MHIMX_CHECK_ARG(!update_q || q_new, "merge_fwd: update_q needs q_new");
  MHIMX_CHECK_ARG(m->own_n == 0, "merge_fwd: a shard of an instance-sharded bag runs mhimx_merge_fwd_part + mhimx_merge_fwd_finish");
  Merge2Ws w;
  if (int r = merge2_fwd_rows(st, m, X, R, ws, ws_bytes, &w)) return r;
  const int J = M2_H * (int)m->k;
  hipLaunchKernelGGL(merge2_partials_kernel<1>, dim3((unsigned)(J * 4)), dim3(M2_THREADS), 0, st, m2_parts_tiles(w, w.ypart), 0, m->ln_w, m->ln_b, w.Y,
                     w.stats, (float*)nullptr);
  MHIMX_LAUNCH_CHECK();
  return merge2_fwd_tail(st, m, z, q_new, update_q, w);
}

// One shard's half of the forward of an instance-sharded bag (sharded.py, BASELINE c5): the rows pass over the rows this shard owns and the
// merge of ITS tile partials, left raw: part [M2_PART_FLOATS] = {max[48] | sum[48] | dropped sum[48] | sum_t e^{pm_t - max} ypart_t [48][512]}.
// The shards all-gather their blocks (99 KB each instead of an all-reduce of the [R, 512] rows) and every shard finishes alike.
int merge2_fwd_part(hipStream_t st, const mhimx_merge* m, const float* X, int64_t R, float* part, void* ws, int64_t ws_bytes) {
  MHIMX_CHECK_ARG(part && aligned16(part), "merge_fwd_part: null / unaligned output block");
  Merge2Ws w;
  if (int r = merge2_fwd_rows(st, m, X, R, ws, ws_bytes, &w)) return r;
  const int J = M2_H * (int)m->k;
  // (slots >= J keep whatever the block held: nobody reads them)
  hipLaunchKernelGGL(merge2_partials_kernel<2>, dim3((unsigned)(J * 4)), dim3(M2_THREADS), 0, st, m2_parts_tiles(w, w.ypart), 0, m->ln_w, m->ln_b,
                     part + 3 * M2_JP, (float*)nullptr, part);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
// ... and the other half, the same on every shard: the W blocks merged in shard order (a second level of the same online-softmax merge:
// fixed order, bit-identical replicas) -> Y and the statistics in THIS shard's workspace (its backward reads them), then the tokens.
int merge2_fwd_finish(hipStream_t st, const mhimx_merge* m, const float* parts, int W, int64_t R, float* z, float* q_new, int update_q, void* ws,
                      int64_t ws_bytes) {
  MHIMX_CHECK_ARG(parts && W >= 1 && W <= 256 && z, "merge_fwd_finish: 1..256 shard blocks");
  MHIMX_CHECK_ARG(!update_q || q_new, "merge_fwd_finish: update_q needs q_new");
  Arena ar(ws, ws_bytes);
  Merge2Ws w;
  merge2_ws_layout(ar, R, m->k, &w);
  MHIMX_CHECK_ARG(ar.ok(), "merge_fwd_finish: workspace too small");
  const int J = M2_H * (int)m->k;
  hipLaunchKernelGGL(merge2_partials_kernel<1>, dim3((unsigned)(J * 4)), dim3(M2_THREADS), 0, st, m2_parts_shards(parts, W), 0, m->ln_w, m->ln_b, w.Y,
                     w.stats, (float*)nullptr);
  MHIMX_LAUNCH_CHECK();
  return merge2_fwd_tail(st, m, z, q_new, update_q, w);
}

int gemm_tn_rider(hipStream_t st, const mhimx_gemm_tn_args& g, const Merge2Side* rider, int stage);      // gemm.hip

int merge2_bwd(hipStream_t st, const mhimx_merge* m, const float* X, int64_t R, const float* dz, float* dX, const mhimx_merge_grad* gr, void* ws,
               int64_t ws_bytes) {
  Arena ar(ws, ws_bytes);
  Merge2Ws w;
  merge2_ws_layout(ar, R, m->k, &w);
  MHIMX_CHECK_ARG(ar.ok(), "merge_bwd: workspace too small");
  MHIMX_CHECK_ARG(m->own_n >= 0 && (m->own_n == 0 || m->x_rows), "merge_bwd: a shard's row range needs the bag-level row list (x_rows)");
  w.own_lo = m->own_n > 0 ? m->own_lo : 0;           // (an instance-sharded bag: this shard's rows only; dX holds them at row id - own_lo)
  w.own_n = m->own_n;
  MHIMX_CHECK_ARG(m->wo_t && aligned16(m->wo_t), "merge_bwd: transposed to_out weight missing");
  const int k = (int)m->k, J = M2_H * k;
  const float scale = 1.0f / sqrtf((float)M2_DH);
  const int acc = gr->accumulate;
  const uint64_t oseed = m->drop_seed + 0x9E3779B97F4A7C15ull;
  Merge2Side sd;
  sd.w = w; sd.dz = dz; sd.U = w.aq; sd.ln_w = m->ln_w; sd.ln_b = m->ln_b; sd.wkv = m->wkv; sd.wq = m->wq; sd.q_param = m->q_param; sd.wo_t = m->wo_t;
  sd.d_wkv = gr->d_wkv; sd.d_wo = gr->d_wo; sd.d_wq = gr->d_wq; sd.d_ln_w = gr->d_ln_w; sd.d_ln_b = gr->d_ln_b; sd.d_bo = gr->d_bo; sd.tick = m->drop_tick;
  sd.oseed = oseed; sd.scale = scale; sd.drop_p = m->drop_p; sd.k = k; sd.accumulate = acc; sd.J = J;
  sd.rep = m->own_n > 0 ? m->rep : 1.f;
  bool pre_done = false;
  if (gr->defer && gr->defer->parked.pending) {
    // the pool backward's scorer-weight-gradient GEMM waits in the list: launch it now, with this backward's parameter-only first stage
    // riding along as its first 64 workgroups (both depend only on the pool backward's outputs)
    mhimx_gemm_tn_args pg;
    memcpy(&pg, gr->defer->parked.blob, sizeof(pg));
    gr->defer->parked.pending = 0;
    const int rc = gemm_tn_rider(st, pg, &sd, 0);
    if (rc < 0) return rc;
    pre_done = rc == 1;
  }
  if (!pre_done) {
    hipLaunchKernelGGL(merge2_bwd_pre_kernel, dim3(M2_BWD_PRE_BLOCKS), dim3(M2_THREADS), 0, st, dz, m->wo_t, m->wkv, k, m->drop_p, oseed, m->drop_tick, gr->d_bo, acc, w, sd.rep);
    MHIMX_LAUNCH_CHECK();
  }
  MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)merge2_rows_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)M2_BWD_SMEM)));
  hipLaunchKernelGGL(merge2_rows_bwd_kernel, dim3((unsigned)w.T), dim3(M2_THREADS), M2_BWD_SMEM, st, X, m->x_rows, R, m->ln_w, m->ln_b, J, m->drop_p,
                     m->drop_seed, m->drop_tick, dX, w);
  MHIMX_LAUNCH_CHECK();
  // the parameter-gradient tail: U [J, E] takes the place of the fp32 copy of aq (not needed any more)
  static const bool no_ride = getenv("MHIMX_MERGE_NO_RIDE") != nullptr;      // (experiments: the tail as three launches of its own)
  if (!no_ride && gr->defer && gr->defer->side.pending == 0) {
    // deferred: the three stages ride in later launches of this backward (mhimx_rows_dpre, the weight-gradient mhimx_gemm_tn,
    // mhimx_reduce_flush); whatever did not get a ride is launched by mhimx_reduce_flush before the reductions
    memcpy(gr->defer->side.blob, &sd, sizeof(sd));
    gr->defer->side.pending = 1;
  } else {
    for (int stage = 1; stage <= 3; ++stage)
      if (int r = merge2_side_launch(st, stage, sd)) return r;
  }
  return 0;
}

}  // namespace mhimx

// host side of prep job kind 6 (gemm_dma.hip): the kernel arguments of the parameter-only part for this Merge and workspace
namespace mhimx {
int merge2_prep_args(const mhimx_merge* m, int64_t R, void* ws, int64_t ws_bytes, Merge2PrepArgs* out) {
  MHIMX_CHECK_ARG(m && merge2_ok(m, R), "prep_batch: the Merge preparation job needs the projection-free form (E = 512, 8 x 64, k <= 6, R <= 32768)");
  Arena ar(ws, ws_bytes);
  Merge2Ws w;
  merge2_ws_layout(ar, R, m->k, &w);
  MHIMX_CHECK_ARG(ar.ok(), "prep_batch: Merge workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)ar.off);
  *out = Merge2PrepArgs{m->q_param, m->ln_w, m->ln_b, m->wq, m->wkv, (int)m->k, 1.0f / sqrtf((float)M2_DH), w};
  return 0;
}
}  // namespace mhimx
