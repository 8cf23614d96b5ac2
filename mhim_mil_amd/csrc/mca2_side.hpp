// mca2_side.hpp — the parameter-gradient tail of the projection-free Merge backward (mca2.hip) as device functions: merging the pooled-row
// partials U, the rank-k weight gradients, the query-side LayerNorm gradients.  Nothing on the data path waits for them (only the optimiser
// reads their outputs), so a trainer lets them RIDE in later launches of the backward instead of paying three launches on the step's serial
// chain: stage 1 in the activation-backward launch (rows.hip), stage 2 in the projection's weight-gradient GEMM (gemm_dma.hip), stage 3 in the
// deferred-reduction launch (rows.hip) - see mhimx_reduce_list.side.  The standalone kernels of mca2.hip run the same bodies.
#pragma once
#include "mca2_prep.hpp"

namespace mhimx {

struct Merge2Side {
  Merge2Ws w;
  const float *dz, *U, *ln_w, *ln_b, *wkv, *wq, *q_param, *wo_t;
  float *d_wkv, *d_wo, *d_wq, *d_ln_w, *d_ln_b, *d_bo;
  const uint64_t* tick;
  uint64_t oseed;
  float scale, drop_p;
  int k, accumulate, J;
  // the rows backward as a stage of its own (stage 4: it rides in the scorer-weight-gradient product's launch, mca2_rows.hpp): its operands
  const float* X; const int64_t* xrows; int64_t R; float* dX; uint64_t seed0;
  float rep;       // weight of the terms every shard of an instance-sharded bag computes identically (d_bo, d_wo, the V half of d_wkv): 1 on one
                   // shard, 0 on the others, so that the SUM over the shards counts them once; 1 for one process
};
static_assert(sizeof(Merge2Side) <= MHIMX_SIDE_BYTES, "Merge2Side must fit the opaque side-work block of mhimx_reduce_list");
// a workgroup of a bag-batched launch: its own bag's workspace, dz / dX / gradient slab and Merge seeds (parameters are shared: they stay)
MHIMX_DEV void bag_move(Merge2Side& a, const BagBatch& bb) {
  if (blockIdx.z == 0) return;
  bag_move(a.w, bb);
#define M2_MV(f) a.f = bag_ptr(a.f, bb)
  M2_MV(dz); M2_MV(U); M2_MV(q_param); M2_MV(wo_t); M2_MV(d_wkv); M2_MV(d_wo); M2_MV(d_wq); M2_MV(d_ln_w); M2_MV(d_ln_b); M2_MV(d_bo); M2_MV(X); M2_MV(xrows);
  M2_MV(dX);
#undef M2_MV
  a.oseed = bag_mca_seed(a.oseed, bb);
  a.seed0 = bag_mca_seed(a.seed0, bb);
}

constexpr int M2_PARTIALS_LDS = 256 + 8 + 128;
// Where the partials of a merge live.  Tiles of ONE process: the workspace arrays (stats pitch 48, pooled rows [T][48][512]).  The shards
// of an instance-sharded bag: the all-gathered per-shard blocks [W][M2_PART_FLOATS] = {max[48] | sum[48] | dropped sum[48] | rows[48][512]}.
constexpr int M2_PART_FLOATS = 3 * M2_JP + M2_JP * M2_E;
struct M2Parts { const float *pm, *pl, *psd, *y; int T; int64_t ld_s, ld_y; };
inline __host__ __device__ M2Parts m2_parts_tiles(const Merge2Ws& w, const float* y) { return M2Parts{w.pm, w.pl, w.psd, y, w.T, M2_JP, (int64_t)M2_JP * M2_E}; }
inline __host__ __device__ M2Parts m2_parts_shards(const float* parts, int W) {
  return M2Parts{parts, parts + M2_JP, parts + 2 * M2_JP, parts + 3 * M2_JP, W, M2_PART_FLOATS, M2_PART_FLOATS};
}
// MODE 0: out[j] = (sum_t y_t[j]) ln_w            (the backward's pooled U; LIVE: only the tiles whose forward partial is not empty count -
//                                                   an instance-sharded bag leaves the others unwritten)
// MODE 1: the online-softmax merge, out = y ln_w + (sum_t psd_t wgt_t) ln_b with wgt_t = e^{pm_t - M} / L, stats = (M, L)
// MODE 2: the same merge left RAW for a second level: raw[j] = sum_t y_t[j] e^{pm_t - M} and (M, L, SD) - one shard's block of the exchange
template <int MODE>
MHIMX_DEV void merge2_partials_body(int block, float* lds, const M2Parts& in, bool live_only, const float* __restrict__ ln_w,
                                    const float* __restrict__ ln_b, float* __restrict__ out, float* __restrict__ stats /* MODE 1: [48][2] */,
                                    float* __restrict__ raw_stats /* MODE 2: {M[48] | L[48] | SD[48]} */) {
  constexpr bool SOFTMAX = MODE != 0;
  float* wt = lds;              // [256]: the weights of one chunk of 256 row tiles
  float* red = wt + 256;        // [8]
  float* half1 = red + 8;       // [128]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = block >> 2, e = (block & 3) * 128 + (tid & 127), half = tid >> 7;
  const int T = in.T;
  float sdl = 0.f, M = 0.f, Lsum = 1.f;
  if (SOFTMAX) {
    // statistics over ALL T tiles (T <= 256: one element per thread, as before; more tiles: a strided loop - R up to 32 768 rows)
    float pm = -INFINITY;
    for (int tt = tid; tt < T; tt += M2_THREADS) pm = fmaxf(pm, in.pm[tt * in.ld_s + j]);
    float m = wave_max(pm);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    M = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float pls = 0.f, pss = 0.f;
    for (int tt = tid; tt < T; tt += M2_THREADS) {
      const float pmt = in.pm[tt * in.ld_s + j];
      const float wgt = pmt == -INFINITY ? 0.f : __expf(pmt - M);         // (an empty partial: weight 0, also when every partial is empty)
      pls += in.pl[tt * in.ld_s + j] * wgt;
      pss += in.psd[tt * in.ld_s + j] * wgt;
    }
    const float l = wave_sum(pls), sd = wave_sum(pss);
    if (lane == 0) { red[4 + wave] = l; half1[wave] = sd; }
    __syncthreads();
    const float L = (red[4] + red[5]) + (red[6] + red[7]);
    const float SD = (half1[0] + half1[1]) + (half1[2] + half1[3]);
    Lsum = L;
    sdl = SD / L;
    if (MODE == 1 && (block & 3) == 0 && tid == 0) { stats[2 * j] = M; stats[2 * j + 1] = L; }
    if (MODE == 2 && (block & 3) == 0 && tid == 0) { raw_stats[j] = M; raw_stats[M2_JP + j] = L; raw_stats[2 * M2_JP + j] = SD; }
  }
  float acc = 0.f;
  const float* pj = in.y + (int64_t)j * M2_E + e;
#pragma unroll 1
  for (int c0 = 0; c0 < T; c0 += 256) {                     // chunks of 256 tiles: their weights in LDS
    __syncthreads();                                        // (the statistics scratch / the previous chunk's weights have been read)
    const int tt = c0 + tid;
    float wv = 0.f;
    if (tt < T) {
      if (SOFTMAX) {
        const float pmt = in.pm[tt * in.ld_s + j];
        wv = pmt == -INFINITY ? 0.f : (MODE == 1 ? __expf(pmt - M) / Lsum : __expf(pmt - M));
      } else {
        wv = (!live_only || in.pl[tt * in.ld_s] != 0.f) ? 1.f : 0.f;
      }
    }
    wt[tid] = wv;
    __syncthreads();
    const int Tc = T - c0 < 256 ? T - c0 : 256;
#pragma unroll 1
    for (int t0 = 0; t0 < Tc; t0 += 32) {
      float v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int t = t0 + half * 16 + q;
        v[q] = (t < Tc && wt[t & 255] != 0.f) ? pj[(int64_t)(c0 + t) * in.ld_y] : 0.f;      // (a tile of weight 0 is never read: it may be unwritten)
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) acc += v[q] * wt[(t0 + half * 16 + q) & 255];
    }
  }
  __syncthreads();
  if (half == 1) half1[tid & 127] = acc;
  __syncthreads();
  if (half == 0) {
    acc += half1[tid];
    out[j * M2_E + e] = MODE == 1 ? acc * ln_w[e] + sdl * ln_b[e] : (MODE == 2 ? acc : acc * ln_w[e]);
  }
}


constexpr int M2_GRADS1_LDS = 12 * M2_E + 32 * 12 + 6 * 16;
constexpr int M2_GRADS1_BLOCKS = 48;
MHIMX_DEV void merge2_grads1_body(int block, float* lds, const Merge2Side& a) {
  const float* __restrict__ dz = a.dz;
  const float* __restrict__ U = a.U;
  const float* __restrict__ wkv = a.wkv;
  const int k = a.k, accumulate = a.accumulate;
  const float scale = a.scale, drop_p = a.drop_p;
  const uint64_t seed0 = a.oseed;
  const uint64_t* __restrict__ tick = a.tick;
  float* __restrict__ d_wkv = a.d_wkv;
  float* __restrict__ d_wo = a.d_wo;
  const Merge2Ws& w = a.w;
  float* us = lds;                                    // [6][512]
  float* ysh = us + 6 * M2_E;                         // [6][512]
  float* qd = ysh + 6 * M2_E;                         // [32][12]: [row][6 x scale Q | 6 x dO]  (d_wo blocks: [32][8] dz0)
  float* dqh = qd + 32 * 12;                          // [6][16]
  const int tid = threadIdx.x;
  if (block >= 32) {
    // 32 rows of d_wo[e, c] = sum_i dz0[i, e] O[i, c]: thread = two columns c, the k values of O in registers
    const int e0 = (block - 32) * 32;
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
      o[0] = accumulate ? o[0] + a.rep * s0 : a.rep * s0;
      o[256] = accumulate ? o[256] + a.rep * s1 : a.rep * s1;
    }
    return;
  }
  const int h = block >> 2, qr = block & 3;             // 16 of the head's 64 rows
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
  // (dQ is what stage 3 reads from this stage.  When both share a launch (bag_wgrad_ws_kernel's trailing workgroups) their workgroups sit on
  // different XCDs: agent-scope relaxed atomics = write-through stores / cache-bypassing loads of just these 10 KB, instead of an L2
  // write-back and invalidation per workgroup - the invalidations took the weight-gradient tiles' operands out of the L2s: +6 us)
  if (tid < 16 * 6 && (tid / 16) < k)
    __hip_atomic_store(w.dQ + (tid / 16) * M2_I + h * 64 + qr * 16 + (tid & 15), scale * dqh[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll 4
  for (int dl = 0; dl < 16; ++dl) {
    const m2_f4 c0 = *reinterpret_cast<const m2_f4*>(qd + dl * 12), c1 = *reinterpret_cast<const m2_f4*>(qd + dl * 12 + 4),
                c2 = *reinterpret_cast<const m2_f4*>(qd + dl * 12 + 8);
    const float q6[6] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1]}, o6[6] = {c1[2], c1[3], c2[0], c2[1], c2[2], c2[3]};
    float sk0 = 0.f, sk1 = 0.f, sv0 = 0.f, sv1 = 0.f;
#pragma unroll
    // (scalar FMAs: this file is built with -fno-slp-vectorize - build.py; the packed form the SLP vectorizer made of the (sv0, sv1) pair
    // lost the i = 3 term in lanes 48..63 about once per 500 launches when a second process shared the GPU)
    for (int i = 0; i < 6; ++i) { sk0 += q6[i] * u0[i]; sk1 += q6[i] * u1[i]; sv0 += o6[i] * y0[i]; sv1 += o6[i] * y1[i]; }
    float* ok = d_wkv + (int64_t)(h * 64 + qr * 16 + dl) * M2_E + tid;
    float* ov = d_wkv + (int64_t)(M2_I + h * 64 + qr * 16 + dl) * M2_E + tid;
    ok[0] = accumulate ? ok[0] + sk0 : sk0;
    ok[256] = accumulate ? ok[256] + sk1 : sk1;
    ov[0] = accumulate ? ov[0] + a.rep * sv0 : a.rep * sv0;          // (dO (x) Y: the same on every shard of a sharded bag - Merge2Side.rep)
    ov[256] = accumulate ? ov[256] + a.rep * sv1 : a.rep * sv1;
  }
}


constexpr int M2_GRADS2_LDS = 6 * M2_I + 4 * 6 * 64;
constexpr int M2_GRADS2_BLOCKS = 48;
MHIMX_DEV void merge2_grads2_body(int block, float* lds, const Merge2Side& a) {
  const float* __restrict__ q_param = a.q_param;
  const float* __restrict__ wq = a.wq;
  const int k = a.k, accumulate = a.accumulate;
  float* __restrict__ d_wq = a.d_wq;
  const Merge2Ws& w = a.w;
  float* part = lds;                                  // [4][6][64]  (d_wq blocks: [32][8])
  float* dqs = part + 4 * 6 * 64;                     // [6][512]
  const int tid = threadIdx.x;
  if (block < 16) {
    // 32 rows of d_wq[c, e] = sum_i dQ[i, c] gq[i, e]: thread = two columns e, the k values of gq in registers
    const int c0 = block * 32;
    float g0[6], g1[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      g0[i] = i < k ? w.gq[i * M2_E + tid] : 0.f;
      g1[i] = i < k ? w.gq[i * M2_E + tid + 256] : 0.f;
    }
    for (int idx = tid; idx < 32 * 8; idx += M2_THREADS) {
      const int r = idx >> 3, i = idx & 7;
      part[idx] = i < k ? __hip_atomic_load(w.dQ + i * M2_I + c0 + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
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
  // 16 columns of dgq per block: thread = (column, one of 16 row groups of Wq), its 32 weights in flight at once
  const int eb = block - 16, c = tid & 15, e = eb * 16 + c, cq = tid >> 4;
  float acc[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) acc[i] = 0.f;
  float wv[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) wv[q] = wq[(int64_t)(cq + 16 * q) * M2_E + e];
  for (int idx = tid; idx < 6 * M2_I; idx += M2_THREADS) dqs[idx] = idx < k * M2_I ? __hip_atomic_load(w.dQ + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 32; ++q)
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[i] += dqs[i * M2_I + cq + 16 * q] * wv[q];
#pragma unroll
  for (int i = 0; i < 6; ++i) part[(cq * 6 + i) * 16 + c] = acc[i];
  __syncthreads();
  if (tid < 16) {
    float dw = 0.f, db = 0.f;
    for (int i = 0; i < k; ++i) {
      float g = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) g += part[(q * 6 + i) * 16 + c];
      const float xhat = (q_param[(int64_t)i * M2_E + e] - w.gmean[i]) * w.grstd[i];
      dw += g * xhat;
      db += g;
    }
    // + the T row-tile partials of the rows' LayerNorm backward (fixed order): the final d_ln_w / d_ln_b, no reduction launch after this
    const float* lp = w.lnpart + e;
    for (int t0 = 0; t0 < w.T; t0 += 16) {
      float pw[16], pb[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        // (an instance-sharded bag: the tiles without a row of this shard wrote no partial - their forward partial is empty)
        const bool ok = t0 + q < w.T && (w.own_n == 0 || w.pl[(int64_t)(t0 + q) * M2_JP] != 0.f);
        pw[q] = ok ? lp[(int64_t)(t0 + q) * 2 * M2_E] : 0.f;
        pb[q] = ok ? lp[(int64_t)(t0 + q) * 2 * M2_E + M2_E] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) { dw += pw[q]; db += pb[q]; }
    }
    a.d_ln_w[e] = accumulate ? a.d_ln_w[e] + dw : dw;
    a.d_ln_b[e] = accumulate ? a.d_ln_b[e] + db : db;
  }
}


constexpr int M2_SIDE_LDS = M2_GRADS1_LDS;                   // floats: the largest of the three stages

// block `b` of stage `stage` (1: U partials, 2: rank-k gradients I, 3: rank-k gradients II); lds: >= M2_SIDE_LDS floats, 16-byte aligned

// Stage 0 (the first stage of the backward, parameters x dz): dz0 = dz keep/(1-p), d_bo, dO = dz0 Wo, dY[(h,i),:] = sum_d dO[i,h,d] Wv[h*64+d,:]
// (as the two fragment images), delta partials dY.Y.   64 blocks = 8 heads x 8 column blocks of 64.  lds: 6 * 512 + 6 * 64 floats.
constexpr int M2_BWD_PRE_BLOCKS = 64, M2_BWD_PRE_LDS = 6 * M2_E + 6 * 64;
// gate (optional): the stage rides in the launch that PRODUCES dz (scorer_fused_bwd_kernel): it requests everything that does not depend on
// dz, then waits until `gate_target` workgroups have announced their rows of dz (write-through stores + one relaxed agent-scope add each)
// and reads dz past the caches.  A bounded spin: the riders are the launch's LAST workgroups, so the producers are resident or done.
constexpr unsigned M2_GATE_SPINS = 1u << 22;
MHIMX_DEV void merge2_bwd_pre_body(int block, float* lds, const float* __restrict__ dz, const float* __restrict__ wo_t,
                                   const float* __restrict__ wkv, int k, float drop_p, uint64_t seed0, const uint64_t* __restrict__ tick,
                                   float* __restrict__ d_bo, int accumulate, const Merge2Ws& w, float rep = 1.f, const unsigned* gate = nullptr,
                                   unsigned gate_target = 0) {
  float* dzs = lds;                 // [6][512]
  float* doh = dzs + 6 * M2_E;      // [6][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int h = block >> 3, eb = block & 7;
  const int J = M2_H * k;
  const int c = tid & 63, e = eb * 64 + c;
  float wv[64];                                               // this thread's column of the head's Wv block: in flight from the start
#pragma unroll
  for (int d = 0; d < 64; ++d) wv[d] = wkv[(int64_t)(M2_I + h * 64 + d) * M2_E + e];
  float yv[2] = {0.f, 0.f};
  for (int i = tid >> 6, q = 0; i < k; i += 4, ++q) yv[q] = w.Y[(h * k + i) * M2_E + e];
  M2HeadRows wr;                                              // ... and so are the head's 64 rows of Wo^T
  m2_head_rows_load<64>(wo_t + (int64_t)h * 64 * M2_E, wr);
  const uint64_t seed = drop_p > 0.f ? eff_seed(seed0, tick) : 0;
  const float ks = 1.f / (1.f - drop_p);
  const float dbo_old = (h == 0 && tid < 64 && accumulate) ? d_bo[e] : 0.f;
  bool gave_up = false;
  if (gate) {
    if (tid == 0) {
      unsigned spins = 0;
      while (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gate_target && ++spins < M2_GATE_SPINS) __builtin_amdgcn_s_sleep(1);
      doh[0] = spins >= M2_GATE_SPINS ? 1.f : 0.f;            // (doh is written for real only after the loop below)
    }
    __syncthreads();
    // the backstop expired (it cannot in a launch whose producers are dispatched first; a preempted or CU-masked queue could): the gradients
    // of this Merge are POISONED with NaN instead of being computed from stale rows of dz - a loud failure, not a silent one
    gave_up = doh[0] != 0.f;
    __syncthreads();
  }
  for (int idx = tid; idx < k * M2_E; idx += M2_THREADS) {
    const int i = idx >> 9, ee = idx & 511;
    float v = gate ? __hip_atomic_load(dz + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : dz[idx];
    if (gave_up) v = __builtin_nanf("");
    if (drop_p > 0.f) v = drop_keep(seed, (uint64_t)i, (uint32_t)ee, drop_p) ? v * ks : 0.f;
    dzs[idx] = v;
  }
  m2_zero_tail(dzs, k);
  __syncthreads();
  if (h == 0 && tid < 64) {
    float s = 0.f;
    for (int i = 0; i < k; ++i) s += dzs[i * M2_E + e];
    d_bo[e] = accumulate ? dbo_old + rep * s : rep * s;
  }
  m2_head_dots_use<64>(wr, dzs, k, doh, 64, eb == 0 ? w.dO + h * 64 : nullptr);
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
}

MHIMX_DEV void merge2_side_stage(int stage, int b, float* lds, const Merge2Side& a) {
  if (stage == 0) merge2_bwd_pre_body(b, lds, a.dz, a.wo_t, a.wkv, a.k, a.drop_p, a.oseed, a.tick, a.d_bo, a.accumulate, a.w, a.rep);
  else if (stage == 1) merge2_partials_body<0>(b, lds, m2_parts_tiles(a.w, a.w.upart), a.w.own_n > 0, a.ln_w, a.ln_b, const_cast<float*>(a.U), nullptr, nullptr);
  else if (stage == 2) merge2_grads1_body(b, lds, a);
  else merge2_grads2_body(b, lds, a);
}
inline int merge2_side_blocks(int stage, const Merge2Side& a) {
  return stage == 0 ? M2_BWD_PRE_BLOCKS : (stage == 1 ? a.J * 4 : (stage == 2 ? M2_GRADS1_BLOCKS : M2_GRADS2_BLOCKS));
}
static_assert(M2_BWD_PRE_LDS <= M2_SIDE_LDS, "stage 0 must fit the riders' LDS");

}  // namespace mhimx
