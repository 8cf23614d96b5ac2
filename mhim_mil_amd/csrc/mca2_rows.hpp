// mca2_rows.hpp - the row-tile passes of the projection-free Merge (mca2.hip) as device functions: the pieces the two row kernels share and
// the rows backward's body, which also runs as riding workgroups of another launch (gemm_dma.hip).
#pragma once
#include <math.h>

#include "mca2_side.hpp"

namespace mhimx {

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
template <int RT>
MHIMX_DEV bool m2_tile_dead(const int64_t* __restrict__ xrows, int64_t R, int64_t row0, const Merge2Ws& w, int* flags) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int64_t dummy;
  const bool mine = lane < RT / 4 && m2_row_ok(xrows, R, row0 + wave + 4 * lane, w, dummy);
  const bool any = __builtin_amdgcn_ballot_w64(mine) != 0;
  if (lane == 0) flags[wave] = any ? 1 : 0;
  __syncthreads();
  const bool dead = (flags[0] | flags[1] | flags[2] | flags[3]) == 0;
  __syncthreads();
  return dead;
}
// RT = rows of a tile (16 or 32: M2_ROWS is the larger; the LDS pitches are the same, the row counts are RT)
template <bool HAVE_STATS, int RT>
MHIMX_DEV void m2_load_rows(const float* __restrict__ X, const int64_t* __restrict__ xrows, int64_t R, int64_t row0, float* xh, float* mean,
                            float* rstd, const float* __restrict__ ln_w, const float* __restrict__ ln_b, float* lnw, float* lnb, float* rs_tile,
                            const Merge2Ws& w, float* ok) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int QN = RT / 4;                                  // rows per wave
  m2_f4 a[QN], b[QN];
  float mu8[QN], rs8[QN];
  bool okq[QN];
#pragma unroll
  for (int q = 0; q < QN; ++q) {                              // all the row loads of this wave in flight
    const int64_t n = row0 + wave + 4 * q;
    const int64_t nc = n < R ? n : R - 1;
    int64_t srow;
    okq[q] = m2_row_ok(xrows, R, n, w, srow);
    const float* src = X + srow * M2_E;
    a[q] = *reinterpret_cast<const m2_f4*>(src + 4 * lane);
    b[q] = *reinterpret_cast<const m2_f4*>(src + 256 + 4 * lane);
    if (HAVE_STATS) { mu8[q] = mean[nc]; rs8[q] = rstd[nc]; }
  }
  if (lnw) {                                                  // (the backward reads the 4 KB of LayerNorm parameters where they lie: no LDS copy)
    if (tid < 128) *reinterpret_cast<m2_f4*>(lnw + 4 * tid) = *reinterpret_cast<const m2_f4*>(ln_w + 4 * tid);
    else *reinterpret_cast<m2_f4*>(lnb + 4 * (tid - 128)) = *reinterpret_cast<const m2_f4*>(ln_b + 4 * (tid - 128));
  }
#pragma unroll
  for (int q = 0; q < QN; ++q) {
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
template <int RT>
MHIMX_DEV void m2_rows_times_slots(const float* xh, const float* lnw, const float* lnb, const M2Frags& f, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kg = lane >> 4;
  constexpr int RB = RT / 16;
  f32x4 acc[RB][3];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) acc[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e0 = (4 * wave + q) * 32 + kg * 8;
    const m2_f4 w0 = *reinterpret_cast<const m2_f4*>(lnw + e0), w1 = *reinterpret_cast<const m2_f4*>(lnw + e0 + 4);
    const m2_f4 b0 = *reinterpret_cast<const m2_f4*>(lnb + e0), b1 = *reinterpret_cast<const m2_f4*>(lnb + e0 + 4);
    bf8 ah[RB], al[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const float* p = xh + (rb * 16 + r16) * M2_XLD + e0;
      const m2_f4 x0 = *reinterpret_cast<const m2_f4*>(p) * w0 + b0, x1 = *reinterpret_cast<const m2_f4*>(p + 4) * w1 + b1;
      const float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
      m2_split8(v, ah[rb], al[rb]);
    }
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
      const bf8 bh = __builtin_bit_cast(bf8, f.h[nb][q]), bl = __builtin_bit_cast(bf8, f.l[nb][q]);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[rb][nb] = m2_mfma3(ah[rb], al[rb], bh, bl, acc[rb][nb]);
    }
  }
  float* out = red + wave * (RT * M2_JP);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
      for (int i = 0; i < 4; ++i) out[(rb * 16 + 4 * kg + i) * M2_JP + nb * 16 + r16] = acc[rb][nb][i];
}

// part[slot][:] = sum_r coefT[slot][r] xhat[r][:]   ([48 x 32] . [32 x 512], 3-term bf16): one 32-deep MFMA step per 16 x 16 block
// (RT = 16: the upper half of the 32-deep step is zero - rows 16..31 do not exist)
template <int RT>
MHIMX_DEV void m2_pool_rows(const float* coefT /* LDS [48][RT + 4] */, const float* xh, float* __restrict__ part /* global [48][512] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kg = lane >> 4;
  constexpr int PLD = RT + 4;
  const bool kin = kg * 8 < RT;
  bf8 ah[3], al[3];
#pragma unroll
  for (int jb = 0; jb < 3; ++jb) {
    const float* p = coefT + (jb * 16 + r16) * PLD + (kin ? kg * 8 : 0);
    m2_f4 c0 = *reinterpret_cast<const m2_f4*>(p), c1 = *reinterpret_cast<const m2_f4*>(p + 4);
    if (!kin) { c0 = m2_f4{0.f, 0.f, 0.f, 0.f}; c1 = c0; }
    const float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
    m2_split8(v, ah[jb], al[jb]);
  }
#pragma unroll 2
  for (int eb = 8 * wave; eb < 8 * wave + 8; ++eb) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = kin ? xh[(kg * 8 + q) * M2_XLD + eb * 16 + r16] : 0.f;
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

// rows per tile of a Merge over R rows (mca2_prep.hpp: merge2_ws_layout lays the partial buffers out for the same choice)
MHIMX_DEV bool m2_keep(uint64_t seed, int j, int64_t r, float p) { return drop_keep(seed, (uint64_t)j, (uint32_t)r, p); }

// ----------------------------------------------------------------------------------------------------------------------
// 2. rows forward: LayerNorm, scores against the J slots, per-tile softmax partials, pooled rows.   grid = ceil(R / RT)
// ----------------------------------------------------------------------------------------------------------------------
struct M2RowsFwd { const float* X; const int64_t* xrows; int64_t R; const float *ln_w, *ln_b; int J; float drop_p; uint64_t seed0; const uint64_t* tick; Merge2Ws w; };
MHIMX_DEV void bag_move(M2RowsFwd& a, const BagBatch& bb) {
  if (blockIdx.z == 0) return;
  bag_move(a.w, bb);
  a.X = bag_ptr(a.X, bb);
  a.xrows = bag_ptr(a.xrows, bb);
  a.seed0 = bag_mca_seed(a.seed0, bb);
}
constexpr size_t m2_fwd_smem(int rt) { return (size_t)(rt * M2_XLD + 4 * rt * M2_JP + M2_JP * (rt + 4) + 2 * M2_E + rt + 4) * sizeof(float); }

// (a device function: the body of merge2_rows_fwd_kernel (mca2.hip) and of the row tiles that ride at the front of the student's one-pass
// scorer launch, scorer_fused.hip)   t: the row tile; m2sm: m2_fwd_smem(RT) bytes of LDS, 16-byte aligned
template <int RT>
MHIMX_DEV void merge2_rows_fwd_body(const int t, float* m2sm, const float* __restrict__ X, const int64_t* __restrict__ xrows, int64_t R,
                                    const float* __restrict__ ln_w, const float* __restrict__ ln_b, int J, float drop_p, uint64_t seed0,
                                    const uint64_t* __restrict__ tick, const Merge2Ws& w) {
  constexpr int PLD = RT + 4, RQ = RT / 4;            // RQ: rows per thread of the four that share a slot
  float* xh = m2sm;                                  // [RT][516]
  float* red = xh + RT * M2_XLD;                     // [4][RT][48]
  float* pdT = red + 4 * RT * M2_JP;                 // [48][RT + 4]
  float* lnw = pdT + M2_JP * PLD;                    // [512]
  float* lnb = lnw + M2_E;                           // [512]
  float* ok = lnb + M2_E;                            // [RT] 1 = the row takes part
  int* flags = reinterpret_cast<int*>(ok + RT);      // [4]
  const int tid = threadIdx.x;
  const int64_t row0 = (int64_t)t * RT;
  if (t == 0 && tid == 0) w.gate[1] = 0u;           // (the backward's first stage may ride behind this gate: scorer_fused_bwd_kernel)
  if (w.own_n > 0 && m2_tile_dead<RT>(xrows, R, row0, w, flags)) {
    // an instance-sharded bag: no row of this tile is this shard's - an empty partial (weight 0 in every merge; its pooled rows are never read)
    if (tid < M2_JP) { w.pm[t * M2_JP + tid] = -INFINITY; w.pl[t * M2_JP + tid] = 0.f; w.psd[t * M2_JP + tid] = 0.f; }
    return;
  }
  M2Frags fr;
  m2_fetch_frags(w.aqf, fr);
  m2_load_rows<false, RT>(X, xrows, R, row0, xh, w.mean, w.rstd, ln_w, ln_b, lnw, lnb, nullptr, w, ok);
  __syncthreads();
  m2_rows_times_slots<RT>(xh, lnw, lnb, fr, red);
  __syncthreads();
  for (int idx = tid; idx < RT * M2_JP; idx += M2_THREADS) {
    const float s = (red[idx] + red[RT * M2_JP + idx]) + (red[2 * RT * M2_JP + idx] + red[3 * RT * M2_JP + idx]);
    red[idx] = s;
    const int r = idx / M2_JP;
    if (row0 + r < R) w.S[(row0 + r) * M2_JP + (idx - r * M2_JP)] = s;
  }
  __syncthreads();
  // per-slot softmax partials of the tile: 4 threads per slot (RT / 4 rows each), combined through LDS
  float* sc = red + RT * M2_JP;                      // [3][4][48] scratch (the partial-product slabs 1..3 are free)
  const int j = tid % M2_JP, rq = tid / M2_JP;        // rq < 4 for the first 192 threads
  float sreg[RQ], m = -INFINITY;
  bool rv[RQ];
  if (rq < 4) {
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
      const int r = rq * RQ + q;
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
    for (int q = 0; q < RQ; ++q) {
      const int r = rq * RQ + q;
      float p = 0.f, pd = 0.f;
      if (j < J && rv[q]) {
        p = __expf(sreg[q] - m);
        pd = (drop_p > 0.f && !m2_keep(seed, j, row0 + r, drop_p)) ? 0.f : p * ks;
      }
      l += p;
      sd += pd;
      pdT[j * PLD + r] = pd;
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
  m2_pool_rows<RT>(pdT, xh, w.ypart + (int64_t)t * M2_JP * M2_E);
}


// ----------------------------------------------------------------------------------------------------------------------
// 5. rows backward: dPd = xn dY^T, softmax backward, dxn = ds aq + Pd dY, LayerNorm backward (dX scattered to the rows' places,
//    per-tile d_ln_w / d_ln_b partials), pooled U partials.   grid = ceil(R / 32)
// ----------------------------------------------------------------------------------------------------------------------
// (round 6: no LDS copy of the LayerNorm weight / bias - 4 KB that put a 16-row tile's 83.3 KB above half a CU's LDS: two tiles per CU now)
constexpr size_t m2_bwd_smem(int rt) { return (size_t)(2 * rt * M2_XLD + rt * M2_CLD + M2_JP * (rt + 4) + 3 * M2_JK + 2 * rt + 4) * sizeof(float); }
static_assert(2 * m2_bwd_smem(16) <= 160 * 1024, "two 16-row tiles of the Merge rows backward share a CU");
constexpr size_t M2_BWD_SMEM = m2_bwd_smem(M2_ROWS);
static_assert(16 * M2_XLD >= 8 * M2_E, "the LayerNorm partials reuse the gradient tile");

// (a device function: the body of merge2_rows_bwd_kernel (mca2.hip) and of the rows pass that rides in the scorer-weight-gradient product's
// launch, gemm_dma.hip: gemm_tn_dma_kernel<.., true>)   t: the row tile; m2sm: M2_BWD_SMEM bytes of LDS, 16-byte aligned
template <int RT>
MHIMX_DEV void merge2_rows_bwd_body(const int t, float* m2sm, const float* __restrict__ X, const int64_t* __restrict__ xrows, int64_t R,
                                    const float* __restrict__ ln_w, const float* __restrict__ ln_b, int J, float drop_p, uint64_t seed0,
                                    const uint64_t* __restrict__ tick, float* __restrict__ dX, const Merge2Ws& w) {
  constexpr int PLD = RT + 4, RB = RT / 16, QN = RT / 4;
  float* xh = m2sm;                                   // [RT][516]
  float* dxs = xh + RT * M2_XLD;                      // [RT][516]; first the [4][RT][48] reduction buffer of dPd
  float* cf = dxs + RT * M2_XLD;                      // [RT][132]: ds (slots 0..63) | Pd (64..127)
  float* dsT = cf + RT * M2_CLD;                      // [48][RT + 4]
  const float* lnw = ln_w;                            // (read where they lie: global memory, 2 KB each, cache resident)
  const float* lnb = ln_b;
  float* sst = dsT + M2_JP * PLD;                     // [64][3]: softmax max, 1 / sum, delta of every slot
  float* rst = sst + 3 * M2_JK;                       // [RT] rstd of the tile's rows
  float* ok = rst + RT;                               // [RT] 1 = the row takes part
  int* flags = reinterpret_cast<int*>(ok + RT);       // [4]
  float* lnred = dxs;                                 // [4][2][512]: the gradient tile has gone out to dX by then
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kg = lane >> 4;
  const int64_t row0 = (int64_t)t * RT;
  if (t == 0 && tid == 0) *w.gate = 0u;               // (the tail's stage-2 arrivals are counted from here: bag_wgrad_ws_kernel)
  // an instance-sharded bag: a tile without a row of this shard leaves no gradient and no partial (the merges of the pooled / LayerNorm
  // partials skip the tiles whose forward partial is empty: w.pl == 0)
  if (w.own_n > 0 && m2_tile_dead<RT>(xrows, R, row0, w, flags)) return;
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
  float sv[RT * M2_JK / M2_THREADS];
#pragma unroll
  for (int q = 0; q < RT * M2_JK / M2_THREADS; ++q) {
    const int idx = tid + q * M2_THREADS, r = idx >> 6, j = idx & 63;
    sv[q] = (j < J && row0 + r < R) ? w.S[(row0 + r) * M2_JP + j] : 0.f;
  }
  m2_load_rows<true, RT>(X, xrows, R, row0, xh, w.mean, w.rstd, ln_w, ln_b, nullptr, nullptr, rst, w, ok);
  __syncthreads();
  m2_rows_times_slots<RT>(xh, lnw, lnb, fr, dxs);
  __syncthreads();
  // ---- softmax backward per (row, slot)
  {
    const uint64_t seed = drop_p > 0.f ? eff_seed(seed0, tick) : 0;
    const float ks = 1.f / (1.f - drop_p);
#pragma unroll
    for (int q = 0; q < RT * M2_JK / M2_THREADS; ++q) {
      const int idx = tid + q * M2_THREADS, r = idx >> 6, j = idx & 63;
      float ds = 0.f, pd = 0.f;
      if (j < J && ok[r] != 0.f) {
        const int qq = r * M2_JP + j;
        const float dpd = (dxs[qq] + dxs[RT * M2_JP + qq]) + (dxs[2 * RT * M2_JP + qq] + dxs[3 * RT * M2_JP + qq]);
        const float p = __expf(sv[q] - sst[3 * j]) * sst[3 * j + 1];
        const float kf = (drop_p > 0.f && !m2_keep(seed, j, row0 + r, drop_p)) ? 0.f : ks;
        pd = p * kf;
        ds = p * (dpd * kf - sst[3 * j + 2]);
      }
      cf[r * M2_CLD + j] = ds;
      cf[r * M2_CLD + M2_JK + j] = pd;
      if (j < M2_JP) dsT[j * PLD + r] = ds;
    }
  }
  __syncthreads();
  // ---- dxn[32 x 512] = cf[32 x 128] . [aq ; dY]  (K = 128 = 4 steps of 32; B from the transposed fragment images, 4 column blocks
  //      = 32 fragment loads in flight at a time)
  {
    bf8 ah[RB][4], al[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
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
        f32x4 acc[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int rb = 0; rb < RB; ++rb)
            acc[rb] = m2_mfma3(ah[rb][ks], al[rb][ks], __builtin_bit_cast(bf8, bh[q][ks]), __builtin_bit_cast(bf8, bl[q][ks]), acc[rb]);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
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
    int64_t dst_row[QN];
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      int64_t srow;
      dst_row[q] = m2_row_ok(xrows, R, row0 + wave + 4 * q, w, srow) ? srow : -1;      // (a shard's dX holds its own rows: id - own_lo)
    }
#pragma unroll
    for (int q = 0; q < QN; ++q) {
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
    __syncthreads();                                  // (lnred aliases dxs: every wave has read its rows of the gradient tile)
    *reinterpret_cast<m2_f4*>(lnred + (wave * 2) * M2_E + 4 * lane) = dwa;
    *reinterpret_cast<m2_f4*>(lnred + (wave * 2) * M2_E + 256 + 4 * lane) = dwb;
    *reinterpret_cast<m2_f4*>(lnred + (wave * 2 + 1) * M2_E + 4 * lane) = dba;
    *reinterpret_cast<m2_f4*>(lnred + (wave * 2 + 1) * M2_E + 256 + 4 * lane) = dbb;
  }
  __syncthreads();
  for (int idx = tid; idx < 2 * M2_E; idx += M2_THREADS)
    w.lnpart[(int64_t)t * 2 * M2_E + idx] = (lnred[idx] + lnred[2 * M2_E + idx]) + (lnred[4 * M2_E + idx] + lnred[6 * M2_E + idx]);
  // ---- pooled U partial (of xhat; the LayerNorm weight is applied when the partials are merged)
  m2_pool_rows<RT>(dsT, xh, w.upart + (int64_t)t * M2_JP * M2_E);
}


}  // namespace mhimx
