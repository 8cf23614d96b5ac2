// mca.hip — Merge: LayerNorm -> multi-head cross attention of k global queries over the R rows that are
// merged away -> output projection -> EMA of the global queries (mhim_modules/merge.py:43-65,127-144).
// The attention itself has only k (1..10) query rows, so it is row-streaming work (wave per key row,
// lane per head-dim column, coalesced 256-B segments), not a GEMM; the R x E projections around it are
// MFMA GEMMs from gemm.hip.
#include <math.h>

#include "common.hpp"

namespace mhimx {

int gemm_nt(hipStream_t st, const mhimx_gemm_nt_args& g);
int gemm_tn(hipStream_t st, const mhimx_gemm_tn_args& g);
int layernorm_fwd(hipStream_t st, const float* x, int64_t M, int64_t E, const float* w, const float* b, float* y, float* mean, float* rstd);
int layernorm_bwd(hipStream_t st, const float* dy, const float* x, int64_t M, int64_t E, const float* w, const float* mean,
                  const float* rstd, float* dx, float* dw_part, float* db_part, float* d_w, float* d_b, int accumulate,
                  int max_parts, const float* dy2, const float* x2, const float* mean2, const float* rstd2, int64_t M2,
                  mhimx_reduce_list* defer, const int64_t* xrows = nullptr, const float* resid = nullptr);
int skinny_pair(hipStream_t st, const mhimx_gemm_tn_args& t, const mhimx_gemm_nt_args& g, float a_drop_p = 0.f, uint64_t a_seed = 0,
                const uint64_t* a_tick = nullptr, float* a_colsum = nullptr, int a_accumulate = 0, int use_a_drop = 0);
bool mca_fused_ok(int64_t E, int64_t heads, int64_t dh, int64_t k, const float* wkv_frag, const float* xn, const float* KV);
int mca_fused_fwd(hipStream_t st, const float* xn, int64_t R, const float* wkv_frag, const float* Q, int kq, int heads, float scale,
                  float drop_p, uint64_t seed, const uint64_t* tick, float* KV, float* dots, float* pm, float* pl, float* po);
int colsum(hipStream_t st, const float* X, int64_t M, int64_t E, float* out, int accumulate, void* ws, int64_t ws_bytes);
// mca2.hip: the projection-free form (k <= 6 queries, E = 512, 8 x 64)
bool merge2_ok(const mhimx_merge* m, int64_t R);
int64_t merge2_ws_bytes(int64_t R, int64_t k);
int merge2_fwd(hipStream_t st, const mhimx_merge* m, const float* X, int64_t R, float* z, float* q_new, int update_q, void* ws, int64_t ws_bytes);
int merge2_fwd_part(hipStream_t st, const mhimx_merge* m, const float* X, int64_t R, float* part, void* ws, int64_t ws_bytes);
int merge2_fwd_finish(hipStream_t st, const mhimx_merge* m, const float* parts, int W, int64_t R, float* z, float* q_new, int update_q, void* ws,
                      int64_t ws_bytes);
int64_t merge2_part_floats();
int merge2_bwd_park(const mhimx_merge* m, const float* X, int64_t R, const float* dz, float* dX, const mhimx_merge_grad* gr, void* ws, int64_t ws_bytes);
int merge2_bwd(hipStream_t st, const mhimx_merge* m, const float* X, int64_t R, const float* dz, float* dX, const mhimx_merge_grad* gr, void* ws,
               int64_t ws_bytes);

constexpr int MCA_THREADS = 256;
#ifdef MHIMX_MCA_PROF
__device__ unsigned long long mca_prof[16];
#define MCA_STAMP(i) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 100) mca_prof[i] = wall_clock64(); } while (0)
extern "C" int mhimx_mca_prof_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mca_prof), 16 * 8); }
#else
#define MCA_STAMP(i)
#endif

MHIMX_DEV float block_reduce_sum(float v, float* red /*[4]*/) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
MHIMX_DEV float block_reduce_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ---------------------------------------------------------------------------------------------------
// Row-parallel cross attention.  A block owns MCA_ROWS consecutive key rows; its 4 waves split the heads
// (wave w: heads w, w+4, ...), lane = head-dim column, so every K/V row is read once, in 256-B segments.
// Forward is flash-style: per (head, query) running max / sum / weighted V sum, one partial per block,
// merged by a tiny finalize kernel; the n x k probability matrix is never stored, only the raw scores.
// ---------------------------------------------------------------------------------------------------
constexpr int MCA_ROWS = 4;             // key rows per block: small => many blocks, the row loop is latency bound
constexpr int MCA_HPW = 2;            // heads per wave (heads <= 8)

template <int KQ>
__global__ __launch_bounds__(MCA_THREADS) void mca_fwd_part_kernel(const float* __restrict__ KV, const float* __restrict__ Q,
                                                                   int64_t R, int heads, int kq, float scale, float drop_p,
                                                                   uint64_t seed0, const uint64_t* __restrict__ tick, float* __restrict__ dots,
                                                                   float* __restrict__ pm, float* __restrict__ pl,
                                                                   float* __restrict__ po) {
  MCA_STAMP(0);
  const uint64_t seed = eff_seed(seed0, tick);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int inner = heads * 64;
  const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  float q[MCA_HPW][KQ], m[MCA_HPW][KQ], l[MCA_HPW][KQ], o[MCA_HPW][KQ];
#pragma unroll
  for (int hh = 0; hh < MCA_HPW; ++hh)
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
      const int h = wave + 4 * hh;
      q[hh][i] = (h < heads && i < kq) ? Q[(int64_t)i * inner + h * 64 + lane] * scale : 0.f;
      m[hh][i] = -INFINITY; l[hh][i] = 0.f; o[hh][i] = 0.f;
    }
  const int64_t r0 = (int64_t)blockIdx.x * MCA_ROWS;
  const int64_t r1 = r0 + MCA_ROWS < R ? r0 + MCA_ROWS : R;
  float kbuf[MCA_ROWS][MCA_HPW], vbuf[MCA_ROWS][MCA_HPW];
#pragma unroll
  for (int rr = 0; rr < MCA_ROWS; ++rr)
#pragma unroll
    for (int hh = 0; hh < MCA_HPW; ++hh) {
      const int h = wave + 4 * hh;
      const int64_t r = r0 + rr;
      const bool ok = h < heads && r < r1;
      kbuf[rr][hh] = ok ? KV[r * 2 * inner + h * 64 + lane] : 0.f;
      vbuf[rr][hh] = ok ? KV[r * 2 * inner + inner + h * 64 + lane] : 0.f;
    }
  MCA_STAMP(1);
#ifdef MHIMX_MCA_PROF
  if (threadIdx.x == 0 && blockIdx.x == 100) mca_prof[8] = __float_as_uint(kbuf[0][0] + q[0][0]);   // force the loads
#endif
  MCA_STAMP(2);
  // One wave per SIMD: a dependent instruction chain runs at its full latency.  So first ALL the block's dot products - 40
  // independent wave reductions whose DPP steps interleave - then, per (head, query), a plain two-pass softmax over the
  // block's MCA_ROWS rows in registers (no running max to rescale).
  float d[MCA_ROWS][MCA_HPW][KQ];
#pragma unroll
  for (int rr = 0; rr < MCA_ROWS; ++rr)
#pragma unroll
    for (int hh = 0; hh < MCA_HPW; ++hh)
#pragma unroll
      for (int i = 0; i < KQ; ++i) d[rr][hh][i] = wave_sum(q[hh][i] * kbuf[rr][hh]);
#pragma unroll
  for (int hh = 0; hh < MCA_HPW; ++hh) {
    const int h = wave + 4 * hh;
    if (h >= heads) continue;
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
      if (i >= kq) continue;
      float mm = -INFINITY;
#pragma unroll
      for (int rr = 0; rr < MCA_ROWS; ++rr)
        if (r0 + rr < r1) mm = fmaxf(mm, d[rr][hh][i]);
      float ll = 0.f, oo = 0.f;
#pragma unroll
      for (int rr = 0; rr < MCA_ROWS; ++rr) {
        const int64_t r = r0 + rr;
        if (r >= r1) continue;
        if (lane == 0) dots[((int64_t)h * kq + i) * R + r] = d[rr][hh][i];
        float ks = 1.f;
        if (drop_p > 0.f) ks = drop_keep(seed, (uint64_t)(h * kq + i), (uint32_t)r, drop_p) ? keep_scale : 0.f;
        const float p = __expf(d[rr][hh][i] - mm);
        ll += p;
        oo += p * ks * vbuf[rr][hh];
      }
      m[hh][i] = mm; l[hh][i] = ll; o[hh][i] = oo;
    }
  }
  MCA_STAMP(3);
#pragma unroll
  for (int hh = 0; hh < MCA_HPW; ++hh) {
    const int h = wave + 4 * hh;
    if (h >= heads) continue;
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
      if (i >= kq) continue;
      const int64_t slot = (int64_t)blockIdx.x * heads * kq + h * kq + i;
      if (lane == 0) { pm[slot] = m[hh][i]; pl[slot] = l[hh][i]; }
      po[slot * 64 + lane] = o[hh][i];
    }
  }
  MCA_STAMP(4);
}

// grid (heads*k), 1024 threads (16 waves split the partial blocks): merge the per-block partials ->
// stats[h,i] = {max, sum}, O[i, h*64+c]
__global__ __launch_bounds__(1024) void mca_fwd_final_kernel(const float* __restrict__ pm, const float* __restrict__ pl,
                                                             const float* __restrict__ po, int nb, int heads, int kq,
                                                             float* __restrict__ stats, float* __restrict__ O) {
  __shared__ float red[16];
  __shared__ float osum[16][64];
  const int hi = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = hi / kq, i = hi % kq;
  const int HK = heads * kq;
  float mx = -INFINITY;
  for (int b = threadIdx.x; b < nb; b += 1024) mx = fmaxf(mx, pm[(int64_t)b * HK + hi]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) mx = fmaxf(mx, red[w]);
  __syncthreads();
  float L = 0.f;
  for (int b = threadIdx.x; b < nb; b += 1024) {
    const float pmb = pm[(int64_t)b * HK + hi];
    L += (pmb == -INFINITY) ? 0.f : pl[(int64_t)b * HK + hi] * __expf(pmb - mx);
  }
  L = wave_sum(L);
  if (lane == 0) red[wave] = L;
  float acc = 0.f;
#pragma unroll 4
  for (int b = wave; b < nb; b += 16) {                   // unrolled: several partial loads in flight per wave
    const float pmb = pm[(int64_t)b * HK + hi];
    const float w = (pmb != -INFINITY) ? __expf(pmb - mx) : 0.f;
    acc += po[((int64_t)b * HK + hi) * 64 + lane] * w;
  }
  osum[wave][lane] = acc;
  __syncthreads();
  if (wave == 0) {
    float Lt = 0.f, a = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) { Lt += red[w]; a += osum[w][lane]; }     // fixed order
    O[(int64_t)i * heads * 64 + h * 64 + lane] = a / Lt;
    if (lane == 0) { stats[2 * hi] = mx; stats[2 * hi + 1] = Lt; }
  }
}

// The attention backward in ONE pass over the key rows: dP_r = keep_r (dO_i . V_r), dd = scale * P (dP - rowdot),
// dK_r = sum_i dd q_i, dV_r = sum_i Pd dO_i, per-block partial dQ_i += dd k_r.
// The softmax's row dot  sum_r P_r dP_r  needs no pass of its own: sum_r P_r keep_r (dO_i . V_r) = dO_i . O_i with
// O = sum_r Pd_r V_r, the forward output that is kept anyway (a second row kernel + a [heads,k,R] buffer less).
template <int KQ>
__global__ __launch_bounds__(MCA_THREADS) void mca_bwd_dkv_kernel(const float* __restrict__ KV, const float* __restrict__ Q,
                                                                  const float* __restrict__ dO, const float* __restrict__ dots,
                                                                  const float* __restrict__ stats, const float* __restrict__ O,
                                                                  int64_t R, int heads,
                                                                  int kq, float scale, float drop_p, uint64_t seed0,
                                                                  const uint64_t* __restrict__ tick, float* __restrict__ dKV, float* __restrict__ pdq) {
  const uint64_t seed = eff_seed(seed0, tick);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int inner = heads * 64;
  const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  float q[MCA_HPW][KQ], go[MCA_HPW][KQ], rd[MCA_HPW][KQ], mx[MCA_HPW][KQ], il[MCA_HPW][KQ], dq[MCA_HPW][KQ];
#pragma unroll
  for (int hh = 0; hh < MCA_HPW; ++hh)
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
      const int h = wave + 4 * hh;
      const bool ok = h < heads && i < kq;
      q[hh][i] = ok ? Q[(int64_t)i * inner + h * 64 + lane] : 0.f;
      go[hh][i] = ok ? dO[(int64_t)i * inner + h * 64 + lane] : 0.f;
      mx[hh][i] = ok ? stats[2 * (h * kq + i)] : 0.f;
      il[hh][i] = ok ? 1.f / stats[2 * (h * kq + i) + 1] : 0.f;
      rd[hh][i] = wave_sum(ok ? go[hh][i] * O[(int64_t)i * inner + h * 64 + lane] : 0.f);
      dq[hh][i] = 0.f;
    }
  const int64_t r0 = (int64_t)blockIdx.x * MCA_ROWS;
  const int64_t r1 = r0 + MCA_ROWS < R ? r0 + MCA_ROWS : R;
  float kbuf[MCA_ROWS][MCA_HPW], vbuf[MCA_ROWS][MCA_HPW], dbuf[MCA_ROWS][MCA_HPW][KQ];       // loads issued up front
#pragma unroll
  for (int rr = 0; rr < MCA_ROWS; ++rr)
#pragma unroll
    for (int hh = 0; hh < MCA_HPW; ++hh) {
      const int h = wave + 4 * hh;
      const int64_t r = r0 + rr;
      const bool ok = h < heads && r < r1;
      kbuf[rr][hh] = ok ? KV[r * 2 * inner + h * 64 + lane] : 0.f;
      vbuf[rr][hh] = ok ? KV[r * 2 * inner + inner + h * 64 + lane] : 0.f;
#pragma unroll
      for (int i = 0; i < KQ; ++i) dbuf[rr][hh][i] = (ok && i < kq) ? dots[((int64_t)h * kq + i) * R + r] : 0.f;
    }
  // all the block's dO.V dot products first: 40 independent wave reductions whose steps interleave (one wave per SIMD runs a
  // dependent chain at its full latency)
  float dpb[MCA_ROWS][MCA_HPW][KQ];
#pragma unroll
  for (int rr = 0; rr < MCA_ROWS; ++rr)
#pragma unroll
    for (int hh = 0; hh < MCA_HPW; ++hh)
#pragma unroll
      for (int i = 0; i < KQ; ++i) dpb[rr][hh][i] = wave_sum(go[hh][i] * vbuf[rr][hh]);
#pragma unroll
  for (int rr = 0; rr < MCA_ROWS; ++rr) {
    const int64_t r = r0 + rr;
    if (r >= r1) break;
#pragma unroll
    for (int hh = 0; hh < MCA_HPW; ++hh) {
      const int h = wave + 4 * hh;
      if (h >= heads) continue;
      const float kv = kbuf[rr][hh];
      float dk = 0.f, dv = 0.f;
#pragma unroll
      for (int i = 0; i < KQ; ++i) {
        if (i >= kq) continue;
        const float p = __expf(dbuf[rr][hh][i] - mx[hh][i]) * il[hh][i];
        float ks = 1.f;
        if (drop_p > 0.f) ks = drop_keep(seed, (uint64_t)(h * kq + i), (uint32_t)r, drop_p) ? keep_scale : 0.f;
        const float dp = dpb[rr][hh][i] * ks;
        const float dd = scale * p * (dp - rd[hh][i]);
        const float pd = p * ks;
        dk += dd * q[hh][i];
        dv += pd * go[hh][i];
        dq[hh][i] += dd * kv;
      }
      dKV[r * 2 * inner + h * 64 + lane] = dk;
      dKV[r * 2 * inner + inner + h * 64 + lane] = dv;
    }
  }
#pragma unroll
  for (int hh = 0; hh < MCA_HPW; ++hh) {
    const int h = wave + 4 * hh;
    if (h >= heads) continue;
#pragma unroll
    for (int i = 0; i < KQ; ++i)
      if (i < kq) pdq[((int64_t)blockIdx.x * kq + i) * inner + h * 64 + lane] = dq[hh][i];
  }
}

// out[j] = sum_b part[b][j]  (fixed order), 2-D parallel: blockDim (64, 4)
__global__ void mca_reduce_kernel(const float* __restrict__ part, int nb, int W, float* __restrict__ out) {
  __shared__ float red[4][64];
  const int j = blockIdx.x * 64 + threadIdx.x;
  float acc = 0.f;
  if (j < W) {
#pragma unroll 8
    for (int b = threadIdx.y; b < nb; b += 4) acc += part[(int64_t)b * W + j];
  }
  red[threadIdx.y][threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.y == 0 && j < W) out[j] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// ---------------------------------------------------------------------------------------------------
// Forward prologue in ONE launch: blocks [0, nbx) LayerNorm the key rows (one wave per row); the next I/8 blocks handle the
// k global queries - each LayerNorms them into LDS (k x E floats, a few KB: cheaper than a launch of its own and a grid
// dependency) and projects them onto its 8 columns of Wq (exact fp32 FMA, wave per output column, as the skinny GEMM does).
// Replaces layernorm(x), layernorm(q), q-projection: three launch floors on the step's serial chain.
// ---------------------------------------------------------------------------------------------------
constexpr int MCA_PRE_COLS = 8;       // Wq columns per query block (4 waves x 2)
MHIMX_DEV void ln_row(const float* __restrict__ xr, int E, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ y,
                      float* mu_out, float* rs_out) {
  const int lane = threadIdx.x & 63;
  float sum = 0.f;
  for (int e = lane; e < E; e += 64) sum += xr[e];
  const float mu = wave_sum(sum) / (float)E;
  float var = 0.f;
  for (int e = lane; e < E; e += 64) { const float d = xr[e] - mu; var += d * d; }
  const float rs = rsqrtf(wave_sum(var) / (float)E + 1e-5f);
  for (int e = lane; e < E; e += 64) y[e] = (xr[e] - mu) * rs * w[e] + b[e];
  *mu_out = mu;
  *rs_out = rs;
}

__global__ __launch_bounds__(256) void mca_pre_kernel(const float* __restrict__ x, int64_t R, int E, const float* __restrict__ ln_w,
                                                      const float* __restrict__ ln_b, float* __restrict__ xn, float* __restrict__ mean,
                                                      float* __restrict__ rstd, const float* __restrict__ q_param, int k,
                                                      float* __restrict__ gq, float* __restrict__ gmean, float* __restrict__ grstd,
                                                      const float* __restrict__ wq, int I, float* __restrict__ Q, int nbx,
                                                      const int64_t* __restrict__ xrows /* optional: row n of the block is x[xrows[n]] */) {
  extern __shared__ __attribute__((aligned(16))) float gqs[];          // [k][E]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)blockIdx.x < nbx) {
    for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < R; n += (int64_t)nbx * 4) {
      float mu, rs;
      ln_row(x + (xrows ? xrows[n] : n) * (int64_t)E, E, ln_w, ln_b, xn + n * (int64_t)E, &mu, &rs);
      if (lane == 0) { mean[n] = mu; rstd[n] = rs; }
    }
    return;
  }
  const int qb = (int)blockIdx.x - nbx;
  for (int i = wave; i < k; i += 4) {
    float mu, rs;
    ln_row(q_param + (int64_t)i * E, E, ln_w, ln_b, gqs + i * E, &mu, &rs);
    if (qb == 0 && lane == 0) { gmean[i] = mu; grstd[i] = rs; }
  }
  __syncthreads();
  if (qb == 0)
    for (int i = threadIdx.x; i < k * E; i += 256) gq[i] = gqs[i];       // kept for the backward
  // two output columns per wave, both weight rows in flight before the first use (the chain is global-load latency)
  const int c0 = qb * MCA_PRE_COLS + wave * 2;
  if (c0 >= I) return;
  const bool two = c0 + 1 < I;
  const float* w0 = wq + (int64_t)c0 * E;
  const float* w1 = wq + (int64_t)(two ? c0 + 1 : c0) * E;
  float acc0[16], acc1[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
  for (int e = lane * 4; e < E; e += 256) {                             // the summation order of skinny_nt_kernel
    const float4 u = *reinterpret_cast<const float4*>(w0 + e), v = *reinterpret_cast<const float4*>(w1 + e);
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i < k) {
        const float4 a = *reinterpret_cast<const float4*>(gqs + i * E + e);
        acc0[i] += a.x * u.x + a.y * u.y + a.z * u.z + a.w * u.w;
        acc1[i] += a.x * v.x + a.y * v.y + a.z * v.z + a.w * v.w;
      }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (i < k) {
      const float s0 = wave_sum(acc0[i]), s1 = wave_sum(acc1[i]);
      if (lane == 0) {
        Q[(int64_t)i * I + c0] = s0;
        if (two) Q[(int64_t)i * I + c0 + 1] = s1;
      }
    }
}

template <typename F>
static int dispatch_kq(int64_t k, F&& f) {
  if (k <= 1) return f(std::integral_constant<int, 1>());
  if (k <= 5) return f(std::integral_constant<int, 5>());
  if (k <= 10) return f(std::integral_constant<int, 10>());
  if (k <= 16) return f(std::integral_constant<int, 16>());
  return fail(-1, "merge: k=%lld global queries unsupported (max 16)", (long long)k);
}

__global__ void ema_kernel(const float* q, const float* __restrict__ z, float* out, int64_t n, float mm) {   // out may alias q
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = q[i] * mm + z[i] * (1.f - mm);
}

// dz0 = dz * keep/(1-p) with the same (seed,row,col) stream as the forward epilogue
__global__ void drop_bwd_kernel(const float* __restrict__ dz, float* __restrict__ out, int64_t M, int E, float p, uint64_t seed0,
                                const uint64_t* __restrict__ tick) {
  const uint64_t seed = eff_seed(seed0, tick);
  const int64_t n = M * E;
  const float ks = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / E;
    const int e = (int)(i - m * E);
    out[i] = (p > 0.f && !drop_keep(seed, (uint64_t)m, (uint32_t)e, p)) ? 0.f : dz[i] * ks;
  }
}

// dz0 = dz * keep/(1-p) (the forward's output-dropout mask) and d_bo[e] (+)= sum_i dz0[i,e], k <= 16 rows: one launch
__global__ void mca_dz0_kernel(const float* __restrict__ dz, float* __restrict__ dz0, int k, int E, float p, uint64_t seed0,
                               const uint64_t* __restrict__ tick, float* __restrict__ d_bo, int accumulate) {
  const uint64_t seed = eff_seed(seed0, tick);
  const float ks = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  float cs = 0.f;
  for (int i = 0; i < k; ++i) {
    const float v = (p > 0.f && !drop_keep(seed, (uint64_t)i, (uint32_t)e, p)) ? 0.f : dz[(int64_t)i * E + e] * ks;
    dz0[(int64_t)i * E + e] = v;
    cs += v;
  }
  d_bo[e] = accumulate ? d_bo[e] + cs : cs;
}

// z = dropout(O Wo^T + bo) for the k merged tokens and, fused, the EMA of the global queries q_new = mm q + (1-mm) z
// (merge.py:127-129,142-143).  One wave per output column; q_new may alias q.
__global__ __launch_bounds__(256) void mca_out_kernel(const float* __restrict__ O, const float* __restrict__ wo, const float* __restrict__ bo,
                                                     int k, int E, int I, float p, uint64_t seed0, const uint64_t* __restrict__ tick,
                                                     float* __restrict__ z, const float* q, float* q_new, float mm, BagBatch bb) {
  if (blockIdx.z) {
    MHIMX_BAG(O); MHIMX_BAG(z); MHIMX_BAG(q); MHIMX_BAG(q_new);
    seed0 = bag_mca_seed(seed0, bb);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wave;
  if (n >= E) return;
  // Everything the epilogue reads is REQUESTED here, ahead of the dot products: the step counter behind the dropout seed, the bias and the
  // k old query values were three dependent cache misses behind the reduction (~1.5 us each on a cold launch of 128 workgroups).
  const uint64_t seed = p > 0.f ? eff_seed(seed0, tick) : 0;
  const float bias = bo[n];
  float qv[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) qv[i] = (q_new && i < k) ? q[(int64_t)i * E + n] : 0.f;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const float* wrow = wo + (int64_t)n * I;
  for (int c = lane * 4; c < I; c += 256) {
    const float4 w = *reinterpret_cast<const float4*>(wrow + c);
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i < k) {
        const float4 a = *reinterpret_cast<const float4*>(O + (int64_t)i * I + c);
        acc[i] += a.x * w.x + a.y * w.y + a.z * w.z + a.w * w.w;
      }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (i < k) acc[i] = wave_sum(acc[i]);                    // (k is uniform: no reduction for the unused accumulators)
  if (lane != 0) return;
  const float ks = p > 0.f ? 1.f / (1.f - p) : 1.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (i >= k) break;
    float v = acc[i] + bias;
    if (p > 0.f) v = drop_keep(seed, (uint64_t)i, (uint32_t)n, p) ? v * ks : 0.f;
    z[(int64_t)i * E + n] = v;
    if (q_new) q_new[(int64_t)i * E + n] = qv[i] * mm + v * (1.f - mm);
  }
}

int mca_out(hipStream_t st, const float* O, const float* wo, const float* bo, int k, int E, int I, float p, uint64_t seed0, const uint64_t* tick,
            float* z, const float* q, float* q_new, float mm) {
  hipLaunchKernelGGL(mca_out_kernel, bgrid((unsigned)cdiv(E, 4)), dim3(256), 0, st, O, wo, bo, k, E, I, p, seed0, tick, z, q, q_new, mm, cur_batch());
  MHIMX_LAUNCH_CHECK();
  return 0;
}

struct MergeWs {
  float *xn, *mean, *rstd, *gq, *gmean, *grstd, *KV, *Q, *P, *O, *dd, *dKV, *dQ, *dO, *dxn, *dgq, *dz0, *lnp_w, *lnp_b, *scratch;
  float *stats, *pm, *pl, *po, *prd, *pdq;
  int nb;
  int64_t scratch_bytes;
  float* nt_ws;
  int64_t nt_ws_floats;
};

static int64_t merge_ws_layout(Arena& ar, int64_t R, int64_t E, int64_t k, int64_t heads, int64_t dh, MergeWs* out) {
  const int64_t I = heads * dh;
  MergeWs w;
  w.xn = ar.take<float>(R * E);
  w.mean = ar.take<float>(R);
  w.rstd = ar.take<float>(R);
  w.gq = ar.take<float>(k * E);
  w.gmean = ar.take<float>(k);
  w.grstd = ar.take<float>(k);
  w.KV = ar.take<float>(R * 2 * I);
  w.Q = ar.take<float>(k * I);
  w.P = ar.take<float>(heads * k * R);
  w.O = ar.take<float>(k * I);
  w.dd = ar.take<float>(heads * k * R);
  w.dKV = ar.take<float>(R * 2 * I);
  w.dQ = ar.take<float>(k * I);
  w.dO = ar.take<float>(k * I);
  w.dxn = ar.take<float>(R * E);
  w.dgq = ar.take<float>(k * E);
  w.dz0 = ar.take<float>(k * E);
  w.lnp_w = ar.take<float>(512 * E);
  w.lnp_b = ar.take<float>(512 * E);
  w.nb = (int)cdiv(R, MCA_ROWS);
  w.stats = ar.take<float>(2 * heads * k);
  w.pm = ar.take<float>((int64_t)w.nb * heads * k);
  w.pl = ar.take<float>((int64_t)w.nb * heads * k);
  w.po = ar.take<float>((int64_t)w.nb * heads * k * 64);
  w.prd = ar.take<float>((int64_t)w.nb * heads * k);
  w.pdq = ar.take<float>((int64_t)w.nb * k * I);
  w.scratch_bytes = 8 * 2 * I * E * 4;         // split-K slabs for dWkv (8 x [2I,E]) / colsum partials
  w.scratch = (float*)ar.take<char>(w.scratch_bytes);
  w.nt_ws_floats = 8 * R * (2 * I > E ? 2 * I : E);      // split-K slabs of the two row GEMMs (few tiles, long K)
  w.nt_ws = ar.take<float>(w.nt_ws_floats);
  if (out) *out = w;
  return ar.off;
}

static int check_merge(const mhimx_merge* m) {
  MHIMX_CHECK_ARG(m && m->q_param && m->ln_w && m->ln_b && m->wkv && m->wq && m->wo && m->bo, "merge: null weights");
  MHIMX_CHECK_ARG(m->dim_head == 64, "merge: dim_head must be 64");
  MHIMX_CHECK_ARG(m->heads > 0 && m->heads <= 8 && m->k > 0 && m->k <= 16 && m->E % 64 == 0, "merge: bad dims (heads <= 8, k <= 16)");
  return 0;
}

int merge_fwd(hipStream_t st, const mhimx_merge* m, const float* X, int64_t R, float* z, float* q_new, int update_q,
              void* ws, int64_t ws_bytes) {
  if (int r = check_merge(m)) return r;
  MHIMX_CHECK_ARG(X && z && R > 0, "merge_fwd: null args");
  if (merge2_ok(m, R)) return merge2_fwd(st, m, X, R, z, q_new, update_q, ws, ws_bytes);
  MHIMX_CHECK_ARG(m->own_n == 0, "merge_fwd: a shard of an instance-sharded bag needs the projection-free form (E = 512, 8 x 64, k <= 6)");
  const int64_t E = m->E, k = m->k, H = m->heads, I = H * m->dim_head;
  Arena ar(ws, ws_bytes);
  MergeWs w;
  merge_ws_layout(ar, R, E, k, H, m->dim_head, &w);
  MHIMX_CHECK_ARG(ar.ok(), "merge_fwd: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)ar.off);
  const int fprec = m->prec == MHIMX_PREC_F32 ? MHIMX_PREC_F32 : MHIMX_PREC_BF16X3;   // small GEMMs: ~fp32 accuracy
  const bool pre_ok = E % 4 == 0 && E <= 1024 && k * E * 4 <= 64 * 1024 && aligned16(m->wq) && aligned16(w.gq);
  if (pre_ok) {
    int nbx = (int)cdiv(R, 4);
    if (nbx > 512) nbx = 512;
        MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)mca_pre_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)));
    hipLaunchKernelGGL(mca_pre_kernel, dim3((unsigned)(nbx + cdiv(I, MCA_PRE_COLS))), dim3(256), (size_t)(k * E * 4), st, X, R, (int)E, m->ln_w, m->ln_b,
                       w.xn, w.mean, w.rstd, m->q_param, (int)k, w.gq, w.gmean, w.grstd, m->wq, (int)I, w.Q, nbx, m->x_rows);
    MHIMX_LAUNCH_CHECK();
  } else {
    MHIMX_CHECK_ARG(!m->x_rows, "merge_fwd: gathered rows (x_rows) need E %% 4 == 0, E <= 1024 and k E <= 16384");
    if (int r = layernorm_fwd(st, X, R, E, m->ln_w, m->ln_b, w.xn, w.mean, w.rstd)) return r;
    if (int r = layernorm_fwd(st, m->q_param, k, E, m->ln_w, m->ln_b, w.gq, w.gmean, w.grstd)) return r;
  }
  mhimx_gemm_nt_args g = {};
  const float scale = 1.0f / sqrtf((float)m->dim_head);
  const bool fused = pre_ok && mca_fused_ok(E, H, m->dim_head, k, m->wkv_frag, w.xn, w.KV) && m->prec != MHIMX_PREC_F32;
  int nparts = w.nb;
  if (fused) {
    // K/V projection + dots + softmax partials of a row tile in one kernel (mca_fused.hip); Q comes from the prologue launch
    nparts = mca_fused_fwd(st, w.xn, R, m->wkv_frag, w.Q, (int)k, (int)H, scale, m->drop_p, m->drop_seed, m->drop_tick, w.KV, w.P, w.pm,
                           w.pl, w.po);
    if (nparts < 0) return nparts;
  } else {
    g.A = w.xn; g.lda = E; g.B = m->wkv; g.ldb = E; g.C = w.KV; g.ldc = 2 * I; g.M = R; g.N = 2 * I; g.K = E; g.prec = fprec;
    g.ws = w.nt_ws; g.ws_floats = w.nt_ws_floats;
    if (int r = gemm_nt(st, g)) return r;
    if (!pre_ok) {
      g = {};
      g.A = w.gq; g.lda = E; g.B = m->wq; g.ldb = E; g.C = w.Q; g.ldc = I; g.M = k; g.N = I; g.K = E; g.prec = fprec;
      if (int r = gemm_nt(st, g)) return r;
    }
    if (int r = dispatch_kq(k, [&](auto kqc) {
          constexpr int KQ = decltype(kqc)::value;
          hipLaunchKernelGGL(mca_fwd_part_kernel<KQ>, dim3((unsigned)w.nb), dim3(MCA_THREADS), 0, st, w.KV, w.Q, R, (int)H, (int)k, scale,
                             m->drop_p, m->drop_seed, m->drop_tick, w.P, w.pm, w.pl, w.po);
          MHIMX_LAUNCH_CHECK();
          return 0;
        })) return r;
  }
  hipLaunchKernelGGL(mca_fwd_final_kernel, dim3((unsigned)(H * k)), dim3(1024), 0, st, w.pm, w.pl, w.po, nparts, (int)H, (int)k, w.stats, w.O);
  MHIMX_LAUNCH_CHECK();
  MHIMX_CHECK_ARG(!update_q || q_new, "merge_fwd: update_q needs q_new");
  MHIMX_CHECK_ARG(I % 4 == 0, "merge_fwd: inner width must be a multiple of 4");
  hipLaunchKernelGGL(mca_out_kernel, dim3((unsigned)cdiv(E, 4)), dim3(256), 0, st, w.O, m->wo, m->bo, (int)k, (int)E, (int)I, m->drop_p,
                     m->drop_seed + 0x9E3779B97F4A7C15ull, m->drop_tick, z, m->q_param, update_q ? q_new : (float*)nullptr, m->mm, BagBatch{});
  MHIMX_LAUNCH_CHECK();
  return 0;
}

int merge_bwd(hipStream_t st, const mhimx_merge* m, const float* X, int64_t R, const float* dz, float* dX,
              const mhimx_merge_grad* gr, void* ws, int64_t ws_bytes) {
  if (int r = check_merge(m)) return r;
  MHIMX_CHECK_ARG(X && dz && dX && gr && R > 0, "merge_bwd: null args");
  MHIMX_CHECK_ARG(gr->d_ln_w && gr->d_ln_b && gr->d_wkv && gr->d_wq && gr->d_wo && gr->d_bo, "merge_bwd: null grads");
  if (merge2_ok(m, R)) return merge2_bwd(st, m, X, R, dz, dX, gr, ws, ws_bytes);
  MHIMX_CHECK_ARG(m->own_n == 0, "merge_bwd: a shard of an instance-sharded bag needs the projection-free form (E = 512, 8 x 64, k <= 6)");
  MHIMX_CHECK_ARG(m->wkv_t && m->wq_t && m->wo_t, "merge_bwd: transposed weights missing");
  const int64_t E = m->E, k = m->k, H = m->heads, I = H * m->dim_head;
  Arena ar(ws, ws_bytes);
  MergeWs w;
  merge_ws_layout(ar, R, E, k, H, m->dim_head, &w);
  MHIMX_CHECK_ARG(ar.ok(), "merge_bwd: workspace too small");
  const int gprec = m->prec == MHIMX_PREC_F32 ? MHIMX_PREC_F32 : MHIMX_PREC_BF16X3;
  const int acc = gr->accumulate;
  // through the output dropout and projection: d_wo = dz0^T O and dO = dz0 Wo with dz0 = dz * keep/(1-p) applied as the two
  // products read dz (and d_bo = column sums of dz0 on the side) - one launch, no dz0 buffer
  mhimx_gemm_tn_args t = {};
  t.A = dz; t.lda = E; t.B = w.O; t.ldb = I; t.C = gr->d_wo; t.ldc = I; t.M = k; t.K1 = E; t.K2 = I; t.splits = 1;
  t.accumulate = acc; t.prec = gprec;
  mhimx_gemm_nt_args g = {};
  g.A = dz; g.lda = E; g.B = m->wo_t; g.ldb = E; g.C = w.dO; g.ldc = I; g.M = k; g.N = I; g.K = E; g.prec = gprec;
  if (int r = skinny_pair(st, t, g, m->drop_p, m->drop_seed + 0x9E3779B97F4A7C15ull, m->drop_tick, gr->d_bo, acc, 1)) return r;
  // attention
  const float scale = 1.0f / sqrtf((float)m->dim_head);
  if (int r = dispatch_kq(k, [&](auto kqc) {
        constexpr int KQ = decltype(kqc)::value;
        hipLaunchKernelGGL(mca_bwd_dkv_kernel<KQ>, dim3((unsigned)w.nb), dim3(MCA_THREADS), 0, st, w.KV, w.Q, w.dO, w.P, w.stats, w.O,
                           R, (int)H, (int)k, scale, m->drop_p, m->drop_seed, m->drop_tick, w.dKV, w.pdq);
        MHIMX_LAUNCH_CHECK();
        return 0;
      })) return r;
  hipLaunchKernelGGL(mca_reduce_kernel, dim3((unsigned)cdiv(k * I, 64)), dim3(64, 4), 0, st, w.pdq, w.nb, (int)(k * I), w.dQ);
  MHIMX_LAUNCH_CHECK();
  // projections
  g = {};
  g.A = w.dKV; g.lda = 2 * I; g.B = m->wkv_t; g.ldb = 2 * I; g.C = w.dxn; g.ldc = E; g.M = R; g.N = E; g.K = 2 * I; g.prec = gprec;
  g.ws = w.nt_ws; g.ws_floats = w.nt_ws_floats;
  if (int r = gemm_nt(st, g)) return r;
  t = {};
  t.A = w.dKV; t.lda = 2 * I; t.B = w.xn; t.ldb = E; t.C = gr->d_wkv; t.ldc = E; t.M = R; t.K1 = 2 * I; t.K2 = E;
  t.splits = 1; t.ws = w.scratch; t.ws_floats = w.scratch_bytes / 4; t.accumulate = acc; t.prec = gprec;
  t.defer = gr->defer;                           // nothing else uses the scratch slabs before the flush
  if (int r = gemm_tn(st, t)) return r;
  t = {};
  t.A = w.dQ; t.lda = I; t.B = w.gq; t.ldb = E; t.C = gr->d_wq; t.ldc = E; t.M = k; t.K1 = I; t.K2 = E; t.splits = 1;
  t.accumulate = acc; t.prec = gprec;
  g = {};
  g.A = w.dQ; g.lda = I; g.B = m->wq_t; g.ldb = I; g.C = w.dgq; g.ldc = E; g.M = k; g.N = E; g.K = I; g.prec = gprec;
  if (int r = skinny_pair(st, t, g)) return r;                    // d_wq = dQ^T LN(q)  and  d LN(q) = dQ Wq: one launch
  // LayerNorm: rows (dX + weight grads), then the global queries (weight grads only; the queries are not trained)
  if (R > 16) {                                  // the k query rows ride along as the last block of the row launch
    if (int r = layernorm_bwd(st, w.dxn, X, R, E, m->ln_w, w.mean, w.rstd, dX, w.lnp_w, w.lnp_b, gr->d_ln_w, gr->d_ln_b, acc, 256, w.dgq,
                              m->q_param, w.gmean, w.grstd, k, gr->defer, m->x_rows)) return r;
  } else {
    if (int r = layernorm_bwd(st, w.dxn, X, R, E, m->ln_w, w.mean, w.rstd, dX, w.lnp_w, w.lnp_b, gr->d_ln_w, gr->d_ln_b, acc, 256, nullptr,
                              nullptr, nullptr, nullptr, 0, nullptr, m->x_rows)) return r;
    if (int r = layernorm_bwd(st, w.dgq, m->q_param, k, E, m->ln_w, w.gmean, w.grstd, nullptr, w.lnp_w, w.lnp_b, gr->d_ln_w, gr->d_ln_b, 1, 256,
                              nullptr, nullptr, nullptr, nullptr, 0, nullptr)) return r;
  }
  return 0;
}

}  // namespace mhimx

using namespace mhimx;

// out = x * keep(seed + tick, row, col) / (1-p): the same counter-based mask the GEMM epilogue applies (forward) so a
// backward can re-apply it to the incoming gradient without any stored mask.
extern "C" int mhimx_dropout_apply(void* stream, const float* x, float* out, int64_t M, int64_t E, float p, uint64_t seed,
                                   const uint64_t* tick) {
  MHIMX_CHECK_ARG(x && out && M >= 0 && E > 0 && p >= 0.f && p < 1.f, "dropout_apply: bad args");
  if (M == 0) return 0;
  hipLaunchKernelGGL(drop_bwd_kernel, dim3((unsigned)(cdiv(M * E, 256) < 4096 ? cdiv(M * E, 256) : 4096)), dim3(256), 0, (hipStream_t)stream, x,
                     out, M, (int)E, p, seed, tick);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t mhimx_merge_ws_bytes(int64_t R, int64_t E, int64_t k, int64_t heads, int64_t dim_head) {
  Arena ar(nullptr, 0);
  const int64_t a = merge_ws_layout(ar, R, E, k, heads, dim_head, nullptr);
  const int64_t b = (E == 512 && heads == 8 && dim_head == 64 && heads * k <= 48) ? merge2_ws_bytes(R, k) : 0;   // either form may run
  return a > b ? a : b;
}
extern "C" int mhimx_merge_fwd(void* stream, const mhimx_merge* m, const float* X, int64_t R, float* z, float* q_new,
                               int32_t update_q, void* ws, int64_t ws_bytes) {
  return merge_fwd((hipStream_t)stream, m, X, R, z, q_new, update_q, ws, ws_bytes);
}
extern "C" int64_t mhimx_merge_part_floats(void) { return merge2_part_floats(); }
extern "C" int mhimx_merge_fwd_part(void* stream, const mhimx_merge* m, const float* X, int64_t R, float* part, void* ws, int64_t ws_bytes) {
  if (int r = check_merge(m)) return r;
  MHIMX_CHECK_ARG(X && R > 0 && ws && merge2_ok(m, R), "merge_fwd_part: needs the projection-free form (E = 512, 8 x 64, k <= 6, R <= 32768, not exact f32)");
  return merge2_fwd_part((hipStream_t)stream, m, X, R, part, ws, ws_bytes);
}
extern "C" int mhimx_merge_fwd_finish(void* stream, const mhimx_merge* m, const float* parts, int32_t W, int64_t R, float* z, float* q_new,
                                      int32_t update_q, void* ws, int64_t ws_bytes) {
  if (int r = check_merge(m)) return r;
  MHIMX_CHECK_ARG(R > 0 && ws && merge2_ok(m, R), "merge_fwd_finish: needs the projection-free form");
  return merge2_fwd_finish((hipStream_t)stream, m, parts, W, R, z, q_new, update_q, ws, ws_bytes);
}
extern "C" int mhimx_merge_bwd_park(const mhimx_merge* m, const float* X, int64_t R, const float* dz, float* dX, const mhimx_merge_grad* g, void* ws,
                                    int64_t ws_bytes) {
  if (check_merge(m)) return 0;
  if (!(X && dz && dX && g && R > 0 && g->d_ln_w && g->d_ln_b && g->d_wkv && g->d_wq && g->d_wo && g->d_bo && m->wo_t)) return 0;
  if (!merge2_ok(m, R) || m->own_n != 0) return 0;        // (the general path / a shard of a sharded bag: the stage runs where it always did)
  return merge2_bwd_park(m, X, R, dz, dX, g, ws, ws_bytes);
}
extern "C" int mhimx_merge_bwd(void* stream, const mhimx_merge* m, const float* X, int64_t R, const float* dz, float* dX,
                               const mhimx_merge_grad* g, void* ws, int64_t ws_bytes) {
  return merge_bwd((hipStream_t)stream, m, X, R, dz, dX, g, ws, ws_bytes);
}
