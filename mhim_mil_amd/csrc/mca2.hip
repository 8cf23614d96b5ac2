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

#include "mca2_rows.hpp"

namespace mhimx {

int64_t merge2_part_floats() { return M2_PART_FLOATS; }
int64_t merge2_rows_tiles(int64_t R) { return cdiv(R, m2_tile_rows(R)); }
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
// 2. rows forward: LayerNorm, scores against the J slots, per-tile softmax partials, pooled rows.   grid = ceil(R / 32)
// ----------------------------------------------------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(M2_THREADS) void merge2_rows_fwd_kernel(const float* __restrict__ X, const int64_t* __restrict__ xrows, int64_t R,
                                                                    const float* __restrict__ ln_w, const float* __restrict__ ln_b, int J,
                                                                    float drop_p, uint64_t seed0, const uint64_t* __restrict__ tick, Merge2Ws w, BagBatch bb) {
  extern __shared__ __attribute__((aligned(16))) float m2sm[];
  if (blockIdx.z) {
    MHIMX_BAG(X); MHIMX_BAG(xrows);
    seed0 = bag_mca_seed(seed0, bb);
    bag_move(w, bb);
  }
  merge2_rows_fwd_body<RT>((int)blockIdx.x, m2sm, X, xrows, R, ln_w, ln_b, J, drop_p, seed0, tick, w);
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
                                                                    float* __restrict__ stats, float* __restrict__ raw_stats, BagBatch bb) {
  __shared__ float lds[M2_PARTIALS_LDS];
  if (blockIdx.z) {
    in.pm = bag_ptr(in.pm, bb); in.pl = bag_ptr(in.pl, bb); in.psd = bag_ptr(in.psd, bb); in.y = bag_ptr(in.y, bb);
    MHIMX_BAG(out); MHIMX_BAG(stats); MHIMX_BAG(raw_stats);
  }
  merge2_partials_body<MODE>((int)blockIdx.x, lds, in, live_only != 0, ln_w, ln_b, out, stats, raw_stats);
}

// ----------------------------------------------------------------------------------------------------------------------
// 3b. O[i, h*64+d] = Wv[h*64+d, :] . Y[(h,i), :].   grid = 8 heads x 4 quarters of the head's 64 columns
// ----------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(M2_THREADS) void merge2_o_kernel(const float* __restrict__ wkv, int k, Merge2Ws w, BagBatch bb) {
  __shared__ __attribute__((aligned(16))) float ys[6 * M2_E];
  bag_move(w, bb);
  const int tid = threadIdx.x;
  const int h = blockIdx.x >> 2, qd = blockIdx.x & 3;
  // (the weight rows are requested BEFORE the pooled rows go through LDS: one memory round trip on the chain instead of two)
  M2HeadRows wr;
  m2_head_rows_load<16>(wkv + (int64_t)(M2_I + h * 64 + qd * 16) * M2_E, wr);                                  // the V half of to_kv
  m2_zero_tail(ys, k);
  for (int idx = tid; idx < k * (M2_E / 4); idx += M2_THREADS)
    reinterpret_cast<m2_f4*>(ys)[idx] = reinterpret_cast<const m2_f4*>(w.Y + (int64_t)h * k * M2_E)[idx];
  __syncthreads();
  m2_head_dots_use<16>(wr, ys, k, nullptr, 16, w.O + h * 64 + qd * 16);
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
                                                                   Merge2Ws w, float rep, BagBatch bb) {
  __shared__ __attribute__((aligned(16))) float lds[M2_BWD_PRE_LDS];
  if (blockIdx.z) {
    MHIMX_BAG(dz); MHIMX_BAG(wo_t); MHIMX_BAG(d_bo);
    seed0 = bag_mca_seed(seed0, bb);
    bag_move(w, bb);
  }
  merge2_bwd_pre_body((int)blockIdx.x, lds, dz, wo_t, wkv, k, drop_p, seed0, tick, d_bo, accumulate, w, rep);      // (mca2_side.hpp)
}

// ----------------------------------------------------------------------------------------------------------------------
// 5. rows backward (body: mca2_rows.hpp).   grid = ceil(R / 32)
// ----------------------------------------------------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(M2_THREADS) void merge2_rows_bwd_kernel(const float* __restrict__ X, const int64_t* __restrict__ xrows, int64_t R,
                                                                    const float* __restrict__ ln_w, const float* __restrict__ ln_b, int J,
                                                                    float drop_p, uint64_t seed0, const uint64_t* __restrict__ tick,
                                                                    float* __restrict__ dX, Merge2Ws w, BagBatch bb) {
  extern __shared__ __attribute__((aligned(16))) float m2sm[];
  if (blockIdx.z) {
    MHIMX_BAG(X); MHIMX_BAG(xrows); MHIMX_BAG(dX);
    seed0 = bag_mca_seed(seed0, bb);
    bag_move(w, bb);
  }
  merge2_rows_bwd_body<RT>((int)blockIdx.x, m2sm, X, xrows, R, ln_w, ln_b, J, drop_p, seed0, tick, dX, w);
}

// ----------------------------------------------------------------------------------------------------------------------
// 6. rank-k gradients, first launch (U merged by merge2_partials_kernel<false> before).   blocks 0..31: (head, quarter): dQ = scale Wk U and
//    16 + 16 rows of d_wkv (K part: scale Q (x) U, V part: dO (x) Y);   blocks 32..47: 32 rows of d_wo = dz0^T O.
// ----------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(M2_THREADS) void merge2_grads1_kernel(Merge2Side a, BagBatch bb) {
  __shared__ __attribute__((aligned(16))) float lds[M2_GRADS1_LDS];
  bag_move(a, bb);
  merge2_grads1_body((int)blockIdx.x, lds, a);
}

// ----------------------------------------------------------------------------------------------------------------------
// 7. rank-k gradients, second launch (needs all of dQ).   blocks 0..15: 32 rows of d_wq = dQ^T gq;   blocks 16..23: 64 columns of
//    dgq = dQ Wq and their LayerNorm-parameter gradients (the queries themselves are not trained) -> partial row T of lnpart.
// ----------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(M2_THREADS) void merge2_grads2_kernel(Merge2Side a, BagBatch bb) {
  __shared__ __attribute__((aligned(16))) float lds[M2_GRADS2_LDS];
  bag_move(a, bb);
  merge2_grads2_body((int)blockIdx.x, lds, a);
}

// ----------------------------------------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------------------------------------
int mca_out(hipStream_t st, const float* O, const float* wo, const float* bo, int k, int E, int I, float p, uint64_t seed0, const uint64_t* tick,
            float* z, const float* q, float* q_new, float mm);                                      // mca.hip
int reduce_parts2(hipStream_t st, const float* part0, const float* part1, int G, int W, int ld, float* out0, float* out1, int accumulate);   // rows.hip

int merge2_side_launch(hipStream_t st, int stage, const Merge2Side& sd) {
  if (stage == 1) hipLaunchKernelGGL(merge2_partials_kernel<0>, bgrid((unsigned)(sd.J * 4)), dim3(M2_THREADS), 0, st, m2_parts_tiles(sd.w, sd.w.upart),
                                     sd.w.own_n > 0 ? 1 : 0, sd.ln_w, sd.ln_b, const_cast<float*>(sd.U), (float*)nullptr, (float*)nullptr, cur_batch());
  else if (stage == 2) hipLaunchKernelGGL(merge2_grads1_kernel, bgrid(M2_GRADS1_BLOCKS), dim3(M2_THREADS), 0, st, sd, cur_batch());
  else hipLaunchKernelGGL(merge2_grads2_kernel, bgrid(M2_GRADS2_BLOCKS), dim3(M2_THREADS), 0, st, sd, cur_batch());
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
// The rows pass as a rider of another launch (round 5: the student's one-pass scorer launch, scorer_fused.hip - the row tiles of the Merge and
// the scorer's tiles over the rows that stay are independent until the tokens exist): the kernel arguments of the pass for this Merge and
// workspace.  Returns 0 and fills *out when the pass can ride (projection-free form, prepared parameters, one process, 16-row tiles: 53 KB of
// LDS under the scorer's 75 KB), 1 when it cannot (the caller runs mhimx_merge_fwd as always), < 0 on error.
int merge2_fwd_rows_args(const mhimx_merge* m, const float* X, int64_t R, void* ws, int64_t ws_bytes, M2RowsFwd* out) {
  if (!m || !X || !merge2_ok(m, R) || !m->prepared || m->own_n != 0) return 1;
  Arena ar(ws, ws_bytes);
  Merge2Ws w;
  merge2_ws_layout(ar, R, m->k, &w);
  MHIMX_CHECK_ARG(ar.ok(), "merge rows rider: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)ar.off);
  if (w.rt != 16) return 1;
  w.own_lo = 0; w.own_n = 0;
  *out = M2RowsFwd{X, m->x_rows, R, m->ln_w, m->ln_b, M2_H * (int)m->k, m->drop_p, m->drop_seed, m->drop_tick, w};
  return 0;
}

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
  if (m->rows_done) {                  // (the rows pass rode in the caller's previous launch: merge2_fwd_rows_args)
    *wout = w;
    return 0;
  }
  if (!m->prepared) {
    hipLaunchKernelGGL(merge2_prep_kernel, dim3(64), dim3(M2_THREADS), 0, st, Merge2PrepArgs{m->q_param, m->ln_w, m->ln_b, m->wq, m->wkv, k, scale, w});
    MHIMX_LAUNCH_CHECK();
  }
  MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)merge2_rows_fwd_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)m2_fwd_smem(32)));
                        MHIMX_HIP(hipFuncSetAttribute((const void*)merge2_rows_fwd_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)m2_fwd_smem(16))));
  if (w.rt == 16)
    hipLaunchKernelGGL(merge2_rows_fwd_kernel<16>, bgrid((unsigned)w.T), dim3(M2_THREADS), m2_fwd_smem(16), st, X, m->x_rows, R, m->ln_w, m->ln_b, J, m->drop_p,
                       m->drop_seed, m->drop_tick, w, cur_batch());
  else
    hipLaunchKernelGGL(merge2_rows_fwd_kernel<32>, bgrid((unsigned)w.T), dim3(M2_THREADS), m2_fwd_smem(32), st, X, m->x_rows, R, m->ln_w, m->ln_b, J, m->drop_p,
                       m->drop_seed, m->drop_tick, w, cur_batch());
  MHIMX_LAUNCH_CHECK();
  *wout = w;
  return 0;
}
// from the merged pooled rows Y (and the softmax statistics) to the tokens: O = Wv Y, to_out, dropout, the queries' EMA
static int merge2_fwd_tail(hipStream_t st, const mhimx_merge* m, float* z, float* q_new, int update_q, const Merge2Ws& w) {
  const int k = (int)m->k;
  hipLaunchKernelGGL(merge2_o_kernel, bgrid(M2_H * 4), dim3(M2_THREADS), 0, st, m->wkv, k, w, cur_batch());
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
  hipLaunchKernelGGL(merge2_partials_kernel<1>, bgrid((unsigned)(J * 4)), dim3(M2_THREADS), 0, st, m2_parts_tiles(w, w.ypart), 0, m->ln_w, m->ln_b, w.Y,
                     w.stats, (float*)nullptr, cur_batch());
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
                     part + 3 * M2_JP, (float*)nullptr, part, BagBatch{});
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
                     w.stats, (float*)nullptr, BagBatch{});
  MHIMX_LAUNCH_CHECK();
  return merge2_fwd_tail(st, m, z, q_new, update_q, w);
}

int gemm_tn_rider(hipStream_t st, const mhimx_gemm_tn_args& g, const Merge2Side* rider, int stage);      // gemm.hip

// the argument block of the backward's stages (every pointer a stage reads or writes), from the caller's structs
static int merge2_side_of(const mhimx_merge* m, const float* X, int64_t R, const float* dz, float* dX, const mhimx_merge_grad* gr, void* ws,
                          int64_t ws_bytes, Merge2Side* out) {
  Arena ar(ws, ws_bytes);
  Merge2Ws w;
  merge2_ws_layout(ar, R, m->k, &w);
  MHIMX_CHECK_ARG(ar.ok(), "merge_bwd: workspace too small");
  MHIMX_CHECK_ARG(m->own_n >= 0 && (m->own_n == 0 || m->x_rows), "merge_bwd: a shard's row range needs the bag-level row list (x_rows)");
  w.own_lo = m->own_n > 0 ? m->own_lo : 0;           // (an instance-sharded bag: this shard's rows only; dX holds them at row id - own_lo)
  w.own_n = m->own_n;
  MHIMX_CHECK_ARG(m->wo_t && aligned16(m->wo_t), "merge_bwd: transposed to_out weight missing");
  const int k = (int)m->k, J = M2_H * k;
  Merge2Side sd = {};
  sd.w = w; sd.dz = dz; sd.U = w.aq; sd.ln_w = m->ln_w; sd.ln_b = m->ln_b; sd.wkv = m->wkv; sd.wq = m->wq; sd.q_param = m->q_param; sd.wo_t = m->wo_t;
  sd.d_wkv = gr->d_wkv; sd.d_wo = gr->d_wo; sd.d_wq = gr->d_wq; sd.d_ln_w = gr->d_ln_w; sd.d_ln_b = gr->d_ln_b; sd.d_bo = gr->d_bo; sd.tick = m->drop_tick;
  sd.oseed = m->drop_seed + 0x9E3779B97F4A7C15ull; sd.scale = 1.0f / sqrtf((float)M2_DH); sd.drop_p = m->drop_p; sd.k = k; sd.accumulate = gr->accumulate;
  sd.J = J;
  sd.rep = m->own_n > 0 ? m->rep : 1.f;
  sd.X = X; sd.xrows = m->x_rows; sd.R = R; sd.dX = dX; sd.seed0 = m->drop_seed;
  *out = sd;
  return 0;
}

// (round 5) Park the backward's FIRST stage (parameters x dz: merge2_bwd_pre) on the step's list BEFORE the pool backward runs: the pool
// backward's one-pass rows launch (scorer_fused_bwd_kernel) then gives it a ride behind a gate on the tile(s) that produce dz - the merged
// tokens' gradient rows - instead of the stage being a 14 us launch of its own between the pool backward and the rows pass.
int merge2_bwd_park(const mhimx_merge* m, const float* X, int64_t R, const float* dz, float* dX, const mhimx_merge_grad* gr, void* ws, int64_t ws_bytes) {
  if (!gr->defer || gr->defer->pre.pending != 0) return 0;
  static const bool ride = getenv("MHIMX_MERGE_PRE_RIDE") == nullptr || atoi(getenv("MHIMX_MERGE_PRE_RIDE")) != 0;
  if (!ride) return 0;
  Merge2Side sd;
  if (int r = merge2_side_of(m, X, R, dz, dX, gr, ws, ws_bytes, &sd)) return r;
  static_assert(sizeof(Merge2Side) <= sizeof(gr->defer->pre.blob), "Merge2Side must fit mhimx_reduce_list.pre");
  memcpy(gr->defer->pre.blob, &sd, sizeof(sd));
  gr->defer->pre.pending = 1;
  return 0;
}

int merge2_bwd(hipStream_t st, const mhimx_merge* m, const float* X, int64_t R, const float* dz, float* dX, const mhimx_merge_grad* gr, void* ws,
               int64_t ws_bytes) {
  Merge2Side sd;
  if (int r = merge2_side_of(m, X, R, dz, dX, gr, ws, ws_bytes, &sd)) return r;
  const Merge2Ws& w = sd.w;
  const int k = sd.k, J = sd.J;
  const int acc = gr->accumulate;
  const uint64_t oseed = sd.oseed;
  bool pre_done = false, rows_done = false;
  if (gr->defer && gr->defer->pre.pending) {
    pre_done = gr->defer->pre.pending == 2;              // 2: the first stage rode in the pool backward's rows launch (merge2_bwd_park)
    gr->defer->pre.pending = 0;
  }
  // (round 5) The pool backward's scorer-weight-gradient GEMM waits in the list; it feeds the optimiser only.  Round 2-4: launched here with
  // this backward's parameter-only first stage riding (16.6 us), THEN the rows pass (23.9 us on 31 CUs).  Now: the first stage as a small
  // launch of its own (it is all the rows pass waits for), then ONE launch of rows pass + product - the row tiles' 31 CUs beside the
  // product's 224: what was 40.5 us of serial chain is the longer of the two.  MHIMX_MERGE_BWD_FUSE=0: the round-4 order.
  static const bool fuse_rows = getenv("MHIMX_MERGE_BWD_FUSE") == nullptr || atoi(getenv("MHIMX_MERGE_BWD_FUSE")) != 0;
  if (gr->defer && gr->defer->parked.pending) {
    mhimx_gemm_tn_args pg;
    memcpy(&pg, gr->defer->parked.blob, sizeof(pg));
    gr->defer->parked.pending = 0;
    if (fuse_rows && pre_done) {
      const int rc = gemm_tn_rider(st, pg, &sd, 4);
      if (rc < 0) return rc;
      rows_done = rc == 1;
    } else {
      // launch it now, with this backward's parameter-only first stage riding along as its first 64 workgroups (both depend only on the
      // pool backward's outputs)
      const int rc = gemm_tn_rider(st, pg, pre_done ? nullptr : &sd, 0);
      if (rc < 0) return rc;
      pre_done = pre_done || rc == 1;
    }
  }
  if (!pre_done) {
    hipLaunchKernelGGL(merge2_bwd_pre_kernel, bgrid(M2_BWD_PRE_BLOCKS), dim3(M2_THREADS), 0, st, dz, m->wo_t, m->wkv, k, m->drop_p, oseed, m->drop_tick, gr->d_bo, acc, w, sd.rep, cur_batch());
    MHIMX_LAUNCH_CHECK();
  }
  if (!rows_done) {
    MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)merge2_rows_bwd_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)m2_bwd_smem(32)));
                          MHIMX_HIP(hipFuncSetAttribute((const void*)merge2_rows_bwd_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)m2_bwd_smem(16))));
    if (w.rt == 16)
      hipLaunchKernelGGL(merge2_rows_bwd_kernel<16>, bgrid((unsigned)w.T), dim3(M2_THREADS), m2_bwd_smem(16), st, X, m->x_rows, R, m->ln_w, m->ln_b, J, m->drop_p,
                         m->drop_seed, m->drop_tick, dX, w, cur_batch());
    else
      hipLaunchKernelGGL(merge2_rows_bwd_kernel<32>, bgrid((unsigned)w.T), dim3(M2_THREADS), m2_bwd_smem(32), st, X, m->x_rows, R, m->ln_w, m->ln_b, J, m->drop_p,
                         m->drop_seed, m->drop_tick, dX, w, cur_batch());
    MHIMX_LAUNCH_CHECK();
  }
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
