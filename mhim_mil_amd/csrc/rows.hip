// rows.hip — row-streaming kernels of the ABMIL path: scorer tail + online-softmax pool (forward and
// backward), pseudo score, LayerNorm, activation backward, column sums.  All HBM-bound: one wave owns
// one token row at a time, lanes stride the row (coalesced 256-B segments), wave reductions by DPP
// shuffles, cross-wave/-block merging by log-sum-exp partials (deterministic fixed-order finalize).
#include <math.h>
#include <string.h>

#include "mca2_rows.hpp"
#include "reduce_jobs.hpp"

namespace mhimx {

int gemm_nt(hipStream_t st, const mhimx_gemm_nt_args& g);
int gemm_tn(hipStream_t st, const mhimx_gemm_tn_args& g);
bool scorer_fused_ok(int64_t E, int64_t A, int gated, int prec, const float* T, const float* wa, const float* wp, int64_t C);
int scorer_fused_fwd(hipStream_t st, const float* T, int64_t M, const float* wa, const float* wa_frag, const float* ba, int act, const float* wc,
                     const float* bc, const float* wp, int C, float* u_pre, float* s_out, float* cproj, float* pm, float* pl,
                     float* pz, int max_parts, const int64_t* rows, const uint8_t* excl, const mhimx_prep_job* ride_jobs, int n_ride_jobs,
                     const void* merge_rows);
int merge2_fwd_rows_args(const mhimx_merge* m, const float* X, int64_t R, void* ws, int64_t ws_bytes, M2RowsFwd* out);      // mca2.hip
int prep_batch(hipStream_t st, const mhimx_prep_job* jobs, int n);      // gemm_dma.hip
int scorer_fused_bwd(hipStream_t st, const float* T, int64_t M, const float* u_pre, const float* s_in, const float* stats,
                     const float* g_z, const float* z, const float* wc, int act, const float* wa_t, const float* wa_t_frag, float* du,
                     float* dT, float* dwc_part, float* dbc_part, int max_parts, const int64_t* rows, const void* pre_side, int64_t gate_row0,
                     void* img, const void* img_dact, float* img_part, int64_t img_rows);

constexpr int ROWS_THREADS = 256;
constexpr int MAX_PART = 512;          // partial blocks per segment
constexpr int TN_SLABS = 128;          // split-reduction slabs for d_wa = du^T T (4 tiles x 128 = 512 workgroups: two per CU once the row table is <= 12 KiB)

MHIMX_DEV float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// ------------------------------------------------------------------------------------------------
// forward: s[n] = wc . (act(a_n) [* sigmoid(b_n)]) + bc ; per-block LSE partial of sum_n e^{s_n} T[n,:]
// ------------------------------------------------------------------------------------------------
constexpr int FWD_WAVES = 16;            // waves per block of the forward row pass: 16 => 4x fewer partials for pool_finalize
template <int EPL>   // E = 64*EPL
__global__ __launch_bounds__(64 * FWD_WAVES) void score_rows_fwd_kernel(
    const float* __restrict__ T, int64_t M, int E, int A, int act, int gated, const float* __restrict__ u_pre,
    const float* __restrict__ wc, const float* __restrict__ bc, const float* __restrict__ wp, int C,
    float* __restrict__ s_out, float* __restrict__ cproj, float* __restrict__ pm, float* __restrict__ pl,
    float* __restrict__ pz, float gdrop_p, uint64_t gdrop_seed, int64_t row0 /* dropout row id of the segment's first row */) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  // sm: [FWD_WAVES][E] wave z-accumulators, then [FWD_WAVES] m, [FWD_WAVES] l
  constexpr int NWV = FWD_WAVES;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ldu = A * (1 + gated);
  const float bias_c = bc ? bc[0] : 0.f;

  float mw = -INFINITY, lw = 0.f, zacc[EPL];
#pragma unroll
  for (int q = 0; q < EPL; ++q) zacc[q] = 0.f;

  for (int64_t n = (int64_t)blockIdx.x * NWV + wave; n < M; n += (int64_t)gridDim.x * NWV) {
    const float* up = u_pre + n * ldu;
    float part = 0.f;
    const bool gdrop = gated && gdrop_p > 0.f;                  // abmil.py:96-98: dropout after the tanh and after the gate
    const uint32_t rkey = gdrop ? drop_row_key(gdrop_seed, (uint64_t)(row0 + n)) : 0u;
    const float ginv = gdrop ? 1.f / (1.f - gdrop_p) : 1.f;
    for (int j = lane; j < A; j += 64) {
      float u = act_fwd(up[j], act);
      if (gated) {
        float sg = sigmoidf_(up[A + j]);
        if (gdrop) {
          u = drop_keep_k(rkey, (uint32_t)j, gdrop_p) ? u * ginv : 0.f;
          sg = drop_keep_k(rkey, (uint32_t)(A + j), gdrop_p) ? sg * ginv : 0.f;
        }
        u *= sg;
      }
      part += wc[j] * u;
    }
    const float s = wave_sum(part) + bias_c;
    const float* h = T + n * (int64_t)E;
    float hv[EPL];
#pragma unroll
    for (int q = 0; q < EPL; ++q) hv[q] = h[lane + 64 * q];
    if (wp) {
      for (int c = 0; c < C; ++c) {
        float d = 0.f;
#pragma unroll
        for (int q = 0; q < EPL; ++q) d += hv[q] * wp[c * E + lane + 64 * q];
        d = wave_sum(d);
        if (lane == 0) cproj[n * C + c] = d;
      }
    }
    if (lane == 0) s_out[n] = s;
    if (s > mw) {
      const float sc = (mw == -INFINITY) ? 0.f : __expf(mw - s);
      lw = lw * sc + 1.f;
#pragma unroll
      for (int q = 0; q < EPL; ++q) zacc[q] = zacc[q] * sc + hv[q];
      mw = s;
    } else {
      const float p = __expf(s - mw);
      lw += p;
#pragma unroll
      for (int q = 0; q < EPL; ++q) zacc[q] += p * hv[q];
    }
  }
  float* zs = sm;
  float* ms = sm + NWV * E;
  float* ls = ms + NWV;
#pragma unroll
  for (int q = 0; q < EPL; ++q) zs[wave * E + lane + 64 * q] = zacc[q];
  if (lane == 0) { ms[wave] = mw; ls[wave] = lw; }
  __syncthreads();
  float mb = ms[0];
#pragma unroll
  for (int i = 1; i < NWV; ++i) mb = fmaxf(mb, ms[i]);
  float w[NWV];
#pragma unroll
  for (int i = 0; i < NWV; ++i) w[i] = (ms[i] == -INFINITY) ? 0.f : __expf(ms[i] - mb);
  for (int e = threadIdx.x; e < E; e += 64 * NWV) {
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < NWV; ++i) a += zs[i * E + e] * w[i];          // fixed order: deterministic
    pz[(int64_t)blockIdx.x * E + e] = a;
  }
  if (threadIdx.x == 0) {
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < NWV; ++i) l += ls[i] * w[i];
    pm[blockIdx.x] = mb;
    pl[blockIdx.x] = l;
  }
}

// merge G partials: stats = {max, sumexp}, z[e] = sum_b pz[b][e] e^{pm[b]-max} / sumexp.
// grid = E/64 blocks of 1024 threads (64 columns x 16 partial groups) + the blocks of the pseudo score; every block re-derives the G weights.
constexpr int FIN_THREADS = 1024;
__global__ __launch_bounds__(FIN_THREADS) void pool_finalize_kernel(const float* __restrict__ pm, const float* __restrict__ pl,
                                                                     const float* __restrict__ pz, int G, int E,
                                                                     float* __restrict__ stats, float* __restrict__ z,
                                                                     const float* __restrict__ s, const float* __restrict__ cproj,
                                                                     const float* __restrict__ bp, int C, int64_t M1,
                                                                     float* __restrict__ pscore, BagBatch bb) {
  __shared__ float red[16];
  __shared__ float wgt[2 * MAX_PART];
  __shared__ float acc16[16][64];
  if (blockIdx.z) {
    MHIMX_BAG(pm); MHIMX_BAG(pl); MHIMX_BAG(pz); MHIMX_BAG(stats); MHIMX_BAG(z); MHIMX_BAG(s); MHIMX_BAG(cproj); MHIMX_BAG(pscore);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // (G <= 2 MAX_PART = FIN_THREADS: one partial per thread, its max and its sum requested together)
  const int b1 = threadIdx.x;
  const float pm1 = b1 < G ? pm[b1] : -INFINITY, pl1 = b1 < G ? pl[b1] : 0.f;
  float m = wave_max(pm1);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  float mx = red[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) mx = fmaxf(mx, red[w]);
  __syncthreads();
  float lp = 0.f;
  if (b1 < G) {
    const float w = (pm1 == -INFINITY) ? 0.f : __expf(pm1 - mx);
    wgt[b1] = w;
    lp = pl1 * w;
  }
  lp = wave_sum(lp);
  if (lane == 0) red[wave] = lp;
  __syncthreads();
  float L = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) L += red[w];                   // fixed order: deterministic
  const int zblocks = (E + 63) / 64;
  if ((int)blockIdx.x >= zblocks) {            // the extra blocks of the launch: the pseudo score of the instances (pseudo_score_kernel's arithmetic)
    const float invL = 1.f / L, b0 = bp ? bp[0] : 0.f;
    const int nb = (int)gridDim.x - zblocks;
    for (int64_t n = (int64_t)((int)blockIdx.x - zblocks) * FIN_THREADS + threadIdx.x; n < M1; n += (int64_t)nb * FIN_THREADS) {
      const float an = __expf(s[n] - mx) * invL;
      float cm = -INFINITY;
      for (int c = 0; c < C; ++c) cm = fmaxf(cm, an * cproj[n * C + c] + b0);
      float den = 0.f;
      for (int c = 0; c < C; ++c) den += expf((an * cproj[n * C + c] + b0) - cm);
      pscore[n] = 1.f / den;
    }
    return;
  }
  const int e = blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (e < E) {
#pragma unroll 8
    for (int b = wave; b < G; b += 16) acc += pz[(int64_t)b * E + e] * wgt[b];       // (8 partial rows in flight per lane)
  }
  acc16[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && e < E) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) a += acc16[w][lane];
    z[e] = a / L;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { stats[0] = mx; stats[1] = L; }
}

// The finalize of a forward whose last K tokens were NOT scored by the scorer launch (mhimx_pool_io.phase 2: the student's merged tokens,
// produced by launches issued after it): every block first scores the K <= 6 tokens itself - u = Wa t (+ ba), s = wc . act(u) (+ bc), the
// scorer's arithmetic in fp32 FMAs, 0.33 MFLOP - and merges them with the G partials as K more one-row partials (fixed order).  Block 0 also
// writes the tokens' u_pre and s where the scorer would have (the backward reads them there).   grid = E / 64, 1024 threads.
constexpr int FIN_MAXTOK = 6;
#ifdef MHIMX_FT_PROF                                            // phase stamps of pool_finalize_tok_kernel (block 0): tools/exp_fintok.py
__device__ unsigned long long ft_prof[16];
#define FT_STAMP(i) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) ft_prof[i] = wall_clock64(); } while (0)
#else
#define FT_STAMP(i)
#endif
__global__ __launch_bounds__(FIN_THREADS) void pool_finalize_tok_kernel(const float* __restrict__ pm, const float* __restrict__ pl,
                                                                         const float* __restrict__ pz, int G, int E, float* __restrict__ stats,
                                                                         float* __restrict__ z, const float* __restrict__ T,
                                                                         const int64_t* __restrict__ rows_tail, int64_t row0, int K,
                                                                         const float* __restrict__ wa, const float* __restrict__ wa_t,
                                                                         const float* __restrict__ ba, const float* __restrict__ wc,
                                                                         const float* __restrict__ bc, int A, int act,
                                                                         float* __restrict__ u_pre_tail, float* __restrict__ s_tail, BagBatch bb) {
  if (blockIdx.z) {
    MHIMX_BAG(pm); MHIMX_BAG(pl); MHIMX_BAG(pz); MHIMX_BAG(stats); MHIMX_BAG(z); MHIMX_BAG(T); MHIMX_BAG(rows_tail); MHIMX_BAG(wa_t); MHIMX_BAG(u_pre_tail);
    MHIMX_BAG(s_tail);
  }
  __shared__ float red[16];
  __shared__ float wgt[2 * MAX_PART];
  __shared__ float acc16[16][64];
  __shared__ __attribute__((aligned(16))) float tks[FIN_MAXTOK][512];
  __shared__ float up[8][FIN_MAXTOK][128];                    // the 8 e-chunks' partial products
  __shared__ float us[FIN_MAXTOK][128];
  __shared__ float stok[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  FT_STAMP(0);
  // the partial statistics and this wave's first 8 pooled rows: requested now, in flight under the token scoring
  const float pm1 = tid < G ? pm[tid] : -INFINITY, pl1 = tid < G ? pl[tid] : 0.f;
  const int e = blockIdx.x * 64 + lane;
  float pzv[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int b = wave + 16 * q;
    pzv[q] = (e < E && b < G) ? pz[(int64_t)b * E + e] : 0.f;
  }
  // ---- token rows -> LDS (rows K.. zero)
  for (int idx = tid; idx < FIN_MAXTOK * (E / 4); idx += FIN_THREADS) {
    const int i = idx / (E / 4), c4 = idx - i * (E / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < K) {
      const int64_t r = rows_tail ? rows_tail[i] : row0 + i;
      v = reinterpret_cast<const float4*>(T + r * E)[c4];
    }
    reinterpret_cast<float4*>(&tks[i][0])[c4] = v;
  }
  // Round 6, stamped (tools/exp_fintok.py, block 0 inside a c2 step, us): weights + token rows landed 4.0 | products + chunk sums 3.3 | token
  // scores 1.0 | statistics 1.2 | pooled rows + z 1.9 = 11.4 of the launch's ~12.  Measured and not kept: the token product on the matrix cores
  // (the one-pass scorer's 3-term bf16 form; wave = 16 scorer columns x half of K, 16 float4 of weights per lane): loads 4.0 -> 8.2 us (a
  // lane's 32-byte pieces of 16 different weight rows against 128 coalesced lanes of the transposed weight), products 3.3 -> 5.0 us (the bf16
  // splits); the second batch of pooled rows requested ahead of the scores: the last phase 2.1 -> 1.7 us, the phase that issues them 1.0 ->
  // 1.6 us - nothing at step level (0.3047-0.3056 ms either way).
  if (wa_t && A == 128 && E == 512) {
    // thread = (scorer column a, chunk of 64 feature dims): the transposed weight is read coalesced (128 threads x 4 B per dim), 16 loads in
    // flight, the tokens broadcast from LDS; no cross-lane reduction - the 8 chunk partials meet in LDS
    const int a = tid & 127, ch = tid >> 7;
    const float* wp = wa_t + (int64_t)(ch * 64) * A + a;
    float acc[FIN_MAXTOK];
#pragma unroll
    for (int i = 0; i < FIN_MAXTOK; ++i) acc[i] = 0.f;
    float wv[64];                                             // ALL of this thread's weights in flight at once: one memory round trip
#pragma unroll
    for (int u = 0; u < 64; ++u) wv[u] = wp[(int64_t)u * A];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (a fence for the scheduler: the 64 loads are issued above it, as one batch)
    __syncthreads();                                          // the token rows are in LDS
    FT_STAMP(1);
#pragma unroll
    for (int u4 = 0; u4 < 64; u4 += 4) {
#pragma unroll
      for (int i = 0; i < FIN_MAXTOK; ++i) {
        const float4 tv = *reinterpret_cast<const float4*>(&tks[i][ch * 64 + u4]);      // (broadcast read: every lane the same address)
        acc[i] += tv.x * wv[u4] + tv.y * wv[u4 + 1] + tv.z * wv[u4 + 2] + tv.w * wv[u4 + 3];
      }
    }
#pragma unroll
    for (int i = 0; i < FIN_MAXTOK; ++i) up[ch][i][a] = acc[i];
    __syncthreads();
    if (tid < K * 128) {
      const int i = tid >> 7, aa = tid & 127;
      float v = ((up[0][i][aa] + up[1][i][aa]) + (up[2][i][aa] + up[3][i][aa])) + ((up[4][i][aa] + up[5][i][aa]) + (up[6][i][aa] + up[7][i][aa]));
      us[i][aa] = v + (ba ? ba[aa] : 0.f);
    }
  } else {
    // (no transposed weight at hand: wave = 8 scorer rows, lanes along the feature dims, a wave reduction per product)
    float4 w0[8], w1[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int a = wave * 8 + q;
      const float* wr = wa + (int64_t)(a < A ? a : 0) * E + lane * 8;
      w0[q] = *reinterpret_cast<const float4*>(wr);
      w1[q] = *reinterpret_cast<const float4*>(wr + 4);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int a = wave * 8 + q;
      for (int i = 0; i < K; ++i) {
        const float4 t0 = *reinterpret_cast<const float4*>(&tks[i][lane * 8]), t1 = *reinterpret_cast<const float4*>(&tks[i][lane * 8 + 4]);
        float d = t0.x * w0[q].x + t0.y * w0[q].y + t0.z * w0[q].z + t0.w * w0[q].w + t1.x * w1[q].x + t1.y * w1[q].y + t1.z * w1[q].z + t1.w * w1[q].w;
        d = wave_sum(d);
        if (lane == 0 && a < A) us[i][a] = d + (ba ? ba[a] : 0.f);
      }
    }
  }
  __syncthreads();
  FT_STAMP(2);
  if (blockIdx.x == 0)
    for (int idx = tid; idx < K * A; idx += FIN_THREADS) u_pre_tail[idx] = us[idx / A][idx % A];
  if (wave < K) {                                             // wave i: s_i = wc . act(u_i) + bc   (A = 128: two per lane)
    float v = 0.f;
    for (int a = lane; a < A; a += 64) v += wc[a] * act_fwd(us[wave][a], act);
    v = wave_sum(v) + (bc ? bc[0] : 0.f);
    if (lane == 0) {
      stok[wave] = v;
      if (blockIdx.x == 0) s_tail[wave] = v;
    }
  }
  // ---- statistics over the G partials and the K tokens
  FT_STAMP(3);
  float m = wave_max(pm1);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  float mx = red[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) mx = fmaxf(mx, red[w]);
  for (int i = 0; i < K; ++i) mx = fmaxf(mx, stok[i]);
  __syncthreads();
  float lp = 0.f;
  if (tid < G) {
    const float w = (pm1 == -INFINITY) ? 0.f : __expf(pm1 - mx);
    wgt[tid] = w;
    lp = pl1 * w;
  }
  lp = wave_sum(lp);
  if (lane == 0) red[wave] = lp;
  __syncthreads();
  float L = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) L += red[w];
  float wt[FIN_MAXTOK];
#pragma unroll
  for (int i = 0; i < FIN_MAXTOK; ++i) {
    wt[i] = i < K ? __expf(stok[i] - mx) : 0.f;
    L += wt[i];
  }
  FT_STAMP(4);
  float acc = 0.f;
  if (e < E) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (wave + 16 * q < G) acc += pzv[q] * wgt[wave + 16 * q];
#pragma unroll 8
    for (int b = wave + 128; b < G; b += 16) acc += pz[(int64_t)b * E + e] * wgt[b];
  }
  acc16[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && e < E) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) a += acc16[w][lane];
#pragma unroll
    for (int i = 0; i < FIN_MAXTOK; ++i)
      if (i < K) a += wt[i] * tks[i][e];
    z[e] = a / L;
  }
  if (blockIdx.x == 0 && tid == 0) { stats[0] = mx; stats[1] = L; }
  FT_STAMP(5);
}

// ------------------------------------------------------------------------------------------------
// backward rows: ds_n = A_n (g_z.h_n - g_z.z); du = ds * wc * act'(a) [gated forms]; attn_n = A_n
// ------------------------------------------------------------------------------------------------
template <int EPL>
__global__ __launch_bounds__(ROWS_THREADS) void score_rows_bwd_kernel(
    const float* __restrict__ T, int64_t M, int E, int A, int act, int gated, const float* __restrict__ u_pre,
    const float* __restrict__ wc, const float* __restrict__ s_in, const float* __restrict__ stats,
    const float* __restrict__ g_z, const float* __restrict__ z, float* __restrict__ du, float* __restrict__ attn,
    float* __restrict__ dwc_part, float* __restrict__ dbc_part, float gdrop_p, uint64_t gdrop_seed, int64_t row0) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [4][A] dwc, [4] dbc
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ldu = A * (1 + gated);
  float gz[EPL];
  float c0 = 0.f;
#pragma unroll
  for (int q = 0; q < EPL; ++q) {
    gz[q] = g_z[lane + 64 * q];
    c0 += gz[q] * z[lane + 64 * q];
  }
  c0 = wave_sum(c0);
  const float mx = stats[0], invL = 1.f / stats[1];
  constexpr int MAXJ = 8;                      // A <= 512
  float dwc[MAXJ];
#pragma unroll
  for (int i = 0; i < MAXJ; ++i) dwc[i] = 0.f;
  float dbc = 0.f;

  for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < M; n += (int64_t)gridDim.x * 4) {
    const float* h = T + n * (int64_t)E;
    float gh = 0.f;
#pragma unroll
    for (int q = 0; q < EPL; ++q) gh += gz[q] * h[lane + 64 * q];
    gh = wave_sum(gh);
    const float an = __expf(s_in[n] - mx) * invL;
    const float ds = an * (gh - c0);
    if (lane == 0) attn[n] = an;
    dbc += ds;
    const float* up = u_pre + n * ldu;
    float* dp = du + n * ldu;
    const bool gdrop = gated && gdrop_p > 0.f;
    const uint32_t rkey = gdrop ? drop_row_key(gdrop_seed, (uint64_t)(row0 + n)) : 0u;
    const float ginv = gdrop ? 1.f / (1.f - gdrop_p) : 1.f;
#pragma unroll
    for (int i = 0; i < MAXJ; ++i) {
      const int j = lane + 64 * i;
      if (j < A) {
        const float a = up[j];
        const float ya = act_fwd(a, act);
        const float ga = act_grad(a, ya, act);
        if (gated) {
          const float sg = sigmoidf_(up[A + j]);
          const float ma = gdrop ? (drop_keep_k(rkey, (uint32_t)j, gdrop_p) ? ginv : 0.f) : 1.f;       // the forward's two masks
          const float mb = gdrop ? (drop_keep_k(rkey, (uint32_t)(A + j), gdrop_p) ? ginv : 0.f) : 1.f;
          dp[j] = ds * wc[j] * (sg * mb) * ga * ma;
          dp[A + j] = ds * wc[j] * (ya * ma) * sg * (1.f - sg) * mb;
          dwc[i] += ds * (ya * ma) * (sg * mb);
        } else {
          dp[j] = ds * wc[j] * ga;
          dwc[i] += ds * ya;
        }
      }
    }
  }
  float* ws = sm;
  float* bs = sm + 4 * A;
#pragma unroll
  for (int i = 0; i < MAXJ; ++i) {
    const int j = lane + 64 * i;
    if (j < A) ws[wave * A + j] = dwc[i];
  }
  if (lane == 0) bs[wave] = dbc;       // every lane holds the same dbc
  __syncthreads();
  for (int j = threadIdx.x; j < A; j += ROWS_THREADS)
    dwc_part[(int64_t)blockIdx.x * A + j] = ws[j] + ws[A + j] + ws[2 * A + j] + ws[3 * A + j];
  if (threadIdx.x == 0) dbc_part[blockIdx.x] = bs[0] + bs[1] + bs[2] + bs[3];
}

// out[j] (+)= sum_b part[b][j]   (fixed order)
// out[j] (+)= sum_b part[b*ld + j].  Launched with RP_THREADS = 32 columns x 32 row groups per block: the G partial rows
// are read 32 at a time (fixed summation order -> deterministic), instead of one serial G-long chain per column.
constexpr int RP_THREADS = 1024;
__global__ __launch_bounds__(RP_THREADS) void reduce_parts_kernel(const float* __restrict__ part, int G, int W, int ld,
                                                                   float* __restrict__ out, int accumulate) {
  __shared__ float red[32][33];
  const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
  for (int j0 = blockIdx.x * 32; j0 < W; j0 += gridDim.x * 32) {
    const int j = j0 + c;
    float acc = 0.f;
    if (j < W) {
#pragma unroll 8
      for (int b = rg; b < G; b += 32) acc += part[(int64_t)b * ld + j];
    }
    red[rg][c] = acc;
    __syncthreads();
    if (rg == 0 && j < W) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < 32; ++q) v += red[q][c];
      out[j] = accumulate ? out[j] + v : v;
    }
    __syncthreads();
  }
}

int reduce_parts_now(hipStream_t st, const float* part, int64_t G, int64_t W, int64_t ld, float* out, int accumulate) {
  MHIMX_CHECK_ARG(cur_batch().n == 0, "reduce: a bag-batched launch queues its reductions (mhimx_reduce_list full?)");
  hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)cdiv(W, 32)), dim3(RP_THREADS), 0, st, part, (int)G, (int)W, (int)ld, out, accumulate);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

// every queued final reduction of a step in one launch (mhimx_reduce_flush): job jb owns blocks [first[jb], first[jb+1]).
// Same arithmetic as reduce_parts_kernel (kind 0) and reduce_slabs_kernel (kind 1): queued or not, the bits are the same.
struct ReduceJobs { ReduceTable t; int side_blocks; Merge2Side side; };
__global__ __launch_bounds__(RP_THREADS) void reduce_batch_kernel(ReduceJobs rj, BagBatch bb) {
  // the last stage of a parked Merge-backward tail rides along (256 of the 1024 threads).  Its workgroups come FIRST: they are a ~9 us
  // latency chain, and behind 600 reduction workgroups (two resident per CU) they started a round late (reduce 11 us alone, 16 with them last)
  if ((int)blockIdx.x < rj.side_blocks) {
    __shared__ __attribute__((aligned(16))) float side_lds[M2_GRADS2_LDS];
    Merge2Side side = rj.side;             // (a copy of the stage's block alone: moving pointers inside rj sends the whole table through scratch)
    bag_move(side, bb);
    if (threadIdx.x < M2_THREADS) merge2_side_stage(3, (int)blockIdx.x, side_lds, side);
    return;
  }
  __shared__ float red[32][33];
  reduce_jobs_block<RP_THREADS, true>(rj.t, (int)blockIdx.x - rj.side_blocks, red, bb);      // (reduce_jobs.hpp)
}

// two independent partial sets in one launch (blockIdx.y selects): LayerNorm's d_w and d_b
__global__ __launch_bounds__(RP_THREADS) void reduce_parts2_kernel(const float* __restrict__ part0, const float* __restrict__ part1,
                                                                    int G, int W, int ld, float* __restrict__ out0,
                                                                    float* __restrict__ out1, int accumulate) {
  __shared__ float red[32][33];
  const float* part = blockIdx.y ? part1 : part0;
  float* out = blockIdx.y ? out1 : out0;
  const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
  for (int j0 = blockIdx.x * 32; j0 < W; j0 += gridDim.x * 32) {
    const int j = j0 + c;
    float acc = 0.f;
    if (j < W) {
#pragma unroll 8
      for (int b = rg; b < G; b += 32) acc += part[(int64_t)b * ld + j];
    }
    red[rg][c] = acc;
    __syncthreads();
    if (rg == 0 && j < W) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < 32; ++q) v += red[q][c];
      out[j] = accumulate ? out[j] + v : v;
    }
    __syncthreads();
  }
}

__global__ void softmax_from_stats_kernel(const float* __restrict__ s, const float* __restrict__ stats,
                                          float* __restrict__ attn, int64_t M, BagBatch bb) {
  if (blockIdx.z) { MHIMX_BAG(s); MHIMX_BAG(stats); MHIMX_BAG(attn); }
  const float mx = stats[0], invL = 1.f / stats[1];
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < M; n += (int64_t)gridDim.x * blockDim.x)
    attn[n] = __expf(s[n] - mx) * invL;
}

// Merge of W partial softmax pools (one per shard of an instance-sharded bag), fixed order => deterministic:
//   parts[w] = (max_w, L_w = sum_n exp(s_n - max_w), z_w[E] = sum_n exp(s_n - max_w) h_n / L_w)
//   M = max_w max_w ; L = sum_w L_w e^{max_w - M} ; z = sum_w z_w L_w e^{max_w - M} / L.   An empty shard sends L_w = 0.
__global__ void lse_merge_kernel(const float* __restrict__ parts, int W, int E, float* __restrict__ stats, float* __restrict__ z) {
  const int ld = E + 2;
  float M = -INFINITY;
  for (int w = 0; w < W; ++w)
    if (parts[w * ld + 1] > 0.f) M = fmaxf(M, parts[w * ld]);
  float Lsum = 0.f;
  for (int w = 0; w < W; ++w)
    if (parts[w * ld + 1] > 0.f) Lsum += parts[w * ld + 1] * __expf(parts[w * ld] - M);
  const float inv = 1.f / Lsum;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int w = 0; w < W; ++w) {
      const float Lw = parts[w * ld + 1];
      if (Lw > 0.f) acc += parts[w * ld + 2 + e] * (Lw * __expf(parts[w * ld] - M));
    }
    z[e] = acc * inv;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { stats[0] = M; stats[1] = Lsum; }
}

// score_n = max_c softmax_c(attn_n * cproj[n,c] + bp[0]) = 1 / sum_c exp(cam_c - max_c cam)
__global__ void pseudo_score_kernel(const float* __restrict__ s, const float* __restrict__ stats,
                                    const float* __restrict__ cproj, const float* __restrict__ bp,
                                    float* __restrict__ score, float* __restrict__ attn_out, int64_t M, int C) {
  const float mx = s ? stats[0] : 0.f, invL = s ? 1.f / stats[1] : 1.f;
  const float b0 = bp ? bp[0] : 0.f;
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < M; n += (int64_t)gridDim.x * blockDim.x) {
    const float an = s ? __expf(s[n] - mx) * invL : 1.f;      // s == NULL: the TransMIL form, cam = cproj + b0 (scoring.py:9-34)
    if (attn_out) attn_out[n] = an;
    float cm = -INFINITY;
    for (int c = 0; c < C; ++c) cm = fmaxf(cm, an * cproj[n * C + c] + b0);
    float den = 0.f;
    for (int c = 0; c < C; ++c) den += expf((an * cproj[n * C + c] + b0) - cm);
    score[n] = 1.f / den;
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm (eps 1e-5, biased variance), one wave per row
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ROWS_THREADS) void layernorm_fwd_kernel(const float* __restrict__ x, int64_t M, int E,
                                                                     const float* __restrict__ w,
                                                                     const float* __restrict__ b, float* __restrict__ y,
                                                                     float* __restrict__ mean, float* __restrict__ rstd) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < M; n += (int64_t)gridDim.x * 4) {
    const float* xr = x + n * (int64_t)E;
    float sum = 0.f;
    for (int e = lane; e < E; e += 64) sum += xr[e];
    const float mu = wave_sum(sum) / (float)E;
    float var = 0.f;
    for (int e = lane; e < E; e += 64) { const float d = xr[e] - mu; var += d * d; }
    const float rs = rsqrtf(wave_sum(var) / (float)E + 1e-5f);
    for (int e = lane; e < E; e += 64) y[n * (int64_t)E + e] = (xr[e] - mu) * rs * w[e] + b[e];
    if (lane == 0) { mean[n] = mu; rstd[n] = rs; }
  }
}

// E = 256 Q (the 512-wide TransMIL tokens): the row is read ONCE, 16 bytes per lane per 256 columns, and stays in registers for the
// mean, the variance and the output - one wave per row, as many blocks as rows / 4 (no partial-count cap: nothing is reduced across rows)
template <int Q>
__global__ __launch_bounds__(ROWS_THREADS) void layernorm_fwd_vec_kernel(const float* __restrict__ x, int64_t M, const float* __restrict__ w,
                                                                         const float* __restrict__ b, float* __restrict__ y,
                                                                         float* __restrict__ mean, float* __restrict__ rstd) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  constexpr int E = 256 * Q;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f4 wv[Q], bv[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    wv[q] = *reinterpret_cast<const f4*>(w + 256 * q + 4 * lane);
    bv[q] = *reinterpret_cast<const f4*>(b + 256 * q + 4 * lane);
  }
  for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < M; n += (int64_t)gridDim.x * 4) {
    f4 v[Q];
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      v[q] = *reinterpret_cast<const f4*>(x + n * E + 256 * q + 4 * lane);
      sum += (v[q][0] + v[q][1]) + (v[q][2] + v[q][3]);
    }
    const float mu = wave_sum(sum) / (float)E;
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      v[q] -= mu;
      var += (v[q][0] * v[q][0] + v[q][1] * v[q][1]) + (v[q][2] * v[q][2] + v[q][3] * v[q][3]);
    }
    const float rs = rsqrtf(wave_sum(var) / (float)E + 1e-5f);
#pragma unroll
    for (int q = 0; q < Q; ++q) *reinterpret_cast<f4*>(y + n * E + 256 * q + 4 * lane) = v[q] * rs * wv[q] + bv[q];
    if (lane == 0) { mean[n] = mu; rstd[n] = rs; }
  }
}

struct LnSeg2 { const float* dy; const float* x; const float* mean; const float* rstd; int64_t M; };

// dx = rstd*(dy*w - mean(dy*w) - xhat*mean(dy*w*xhat)); per-block partial dw = sum dy*xhat, db = sum dy
__global__ __launch_bounds__(ROWS_THREADS) void layernorm_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, int64_t M, int E, const float* __restrict__ w,
    const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dx,
    float* __restrict__ dw_part, float* __restrict__ db_part, int direct_accumulate /* -1: partial rows; 0/1: one block writes d_w, d_b */,
    LnSeg2 s2, const int64_t* __restrict__ xrows /* optional: x and dx rows of the main segment live at xrows[n] */,
    const float* __restrict__ resid /* optional: dx = (LayerNorm backward) + resid, the residual branch's gradient */) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [4][E] dw, [4][E] db
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int MAXQ = 16;                    // E <= 1024
  float dwa[MAXQ], dba[MAXQ];
#pragma unroll
  for (int q = 0; q < MAXQ; ++q) dwa[q] = dba[q] = 0.f;
  // a second row set that shares the weight (the k global queries of Merge; weight gradients only) rides along as the
  // LAST block of the launch instead of as a launch of its own
  int64_t nblk = gridDim.x, first = blockIdx.x;
  if (s2.M > 0) {
    if (blockIdx.x == gridDim.x - 1) { dy = s2.dy; x = s2.x; mean = s2.mean; rstd = s2.rstd; M = s2.M; dx = nullptr; nblk = 1; first = 0; xrows = nullptr; }
    else nblk = gridDim.x - 1;
  }
  for (int64_t n = first * 4 + wave; n < M; n += nblk * 4) {
    const float mu = mean[n], rs = rstd[n];
    const int64_t xn = xrows ? xrows[n] : n;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
      const int e = lane + 64 * q;
      if (e < E) {
        const float g = dy[n * (int64_t)E + e];
        const float xh = (x[xn * (int64_t)E + e] - mu) * rs;
        const float gw = g * w[e];
        s1 += gw; s2 += gw * xh;
        dwa[q] += g * xh; dba[q] += g;
      }
    }
    s1 = wave_sum(s1) / (float)E;
    s2 = wave_sum(s2) / (float)E;
    if (dx) {
#pragma unroll
      for (int q = 0; q < MAXQ; ++q) {
        const int e = lane + 64 * q;
        if (e < E) {
          const float xh = (x[xn * (int64_t)E + e] - mu) * rs;
          float o = rs * (dy[n * (int64_t)E + e] * w[e] - s1 - xh * s2);
          if (resid) o += resid[xn * (int64_t)E + e];
          dx[xn * (int64_t)E + e] = o;
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < MAXQ; ++q) {
    const int e = lane + 64 * q;
    if (e < E) { sm[wave * E + e] = dwa[q]; sm[(4 + wave) * E + e] = dba[q]; }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += ROWS_THREADS) {
    const float a = sm[e] + sm[E + e] + sm[2 * E + e] + sm[3 * E + e];
    const float b = sm[4 * E + e] + sm[5 * E + e] + sm[6 * E + e] + sm[7 * E + e];
    if (direct_accumulate < 0) {
      dw_part[(int64_t)blockIdx.x * E + e] = a;
      db_part[(int64_t)blockIdx.x * E + e] = b;
    } else {                                          // single block (a handful of rows): the final gradients, no reduce launch
      dw_part[e] = direct_accumulate ? dw_part[e] + a : a;
      db_part[e] = direct_accumulate ? db_part[e] + b : b;
    }
  }
}

// E = 512 rows without a ride-along segment or row map (the TransMIL token stream): dy, x (and the residual) are read ONCE, 16 bytes per
// lane per 256 columns, and stay in registers for the two row sums, dx and the weight-gradient partials (lane-owned columns).
__global__ __launch_bounds__(ROWS_THREADS) void layernorm_bwd_vec512_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, int64_t M, const float* __restrict__ w, const float* __restrict__ mean,
    const float* __restrict__ rstd, float* __restrict__ dx, float* __restrict__ dw_part, float* __restrict__ db_part,
    const float* __restrict__ resid) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  constexpr int E = 512;
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [4][E] dw, [4][E] db
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f4 wv[2], dwa[2], dba[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    wv[q] = *reinterpret_cast<const f4*>(w + 256 * q + 4 * lane);
    dwa[q] = f4{0.f, 0.f, 0.f, 0.f};
    dba[q] = f4{0.f, 0.f, 0.f, 0.f};
  }
  for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < M; n += (int64_t)gridDim.x * 4) {
    const float mu = mean[n], rs = rstd[n];
    f4 g[2], xh[2], rr[2];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      g[q] = *reinterpret_cast<const f4*>(dy + n * E + 256 * q + 4 * lane);
      xh[q] = (*reinterpret_cast<const f4*>(x + n * E + 256 * q + 4 * lane) - mu) * rs;
      if (resid) rr[q] = *reinterpret_cast<const f4*>(resid + n * E + 256 * q + 4 * lane);
      const f4 gw = g[q] * wv[q], gx = gw * xh[q];
      s1 += (gw[0] + gw[1]) + (gw[2] + gw[3]);
      s2 += (gx[0] + gx[1]) + (gx[2] + gx[3]);
      dwa[q] += g[q] * xh[q];
      dba[q] += g[q];
    }
    s1 = wave_sum(s1) / (float)E;
    s2 = wave_sum(s2) / (float)E;
    if (dx) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f4 o = (g[q] * wv[q] - s1 - xh[q] * s2) * rs;
        if (resid) o += rr[q];
        *reinterpret_cast<f4*>(dx + n * E + 256 * q + 4 * lane) = o;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    *reinterpret_cast<f4*>(sm + wave * E + 256 * q + 4 * lane) = dwa[q];
    *reinterpret_cast<f4*>(sm + (4 + wave) * E + 256 * q + 4 * lane) = dba[q];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += ROWS_THREADS) {
    dw_part[(int64_t)blockIdx.x * E + e] = sm[e] + sm[E + e] + sm[2 * E + e] + sm[3 * E + e];
    db_part[(int64_t)blockIdx.x * E + e] = sm[4 * E + e] + sm[5 * E + e] + sm[6 * E + e] + sm[7 * E + e];
  }
}

// ------------------------------------------------------------------------------------------------
// feature activation backward (in place on dH) and column sums
// ------------------------------------------------------------------------------------------------
// one block = a chunk of rows, thread = a pair of columns: the backward through act+dropout and the bias gradient
// (column sums of dPre) share one pass over dH.
__global__ __launch_bounds__(256) void act_bwd_kernel(float* __restrict__ dH, const float* __restrict__ H,
                                                     const float* __restrict__ pre, int64_t M, int E, int act, float drop_p,
                                                     uint64_t seed0, const uint8_t* __restrict__ drop_mask,
                                                     const int64_t* __restrict__ rows, int64_t chunk, float* __restrict__ part,
                                                     const uint64_t* __restrict__ tick) {
  const uint64_t seed = eff_seed(seed0, tick);
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const int64_t mb = (int64_t)blockIdx.x * chunk;
  const int64_t me = mb + chunk < M ? mb + chunk : M;
  for (int e = threadIdx.x; e < E; e += 256) {
    float colsum = 0.f;
    for (int64_t m = mb; m < me; ++m) {
      const int64_t i = m * E + e;
      float g = dH[i];
      bool keep = true;
      if (drop_mask) keep = drop_mask[i] != 0;
      else if (drop_p > 0.f) keep = drop_keep(seed, rows ? (uint64_t)rows[m] : (uint64_t)m, (uint32_t)e, drop_p);
      if (!keep) {
        g = 0.f;
      } else {
        g *= inv_keep;
        if (act == MHIMX_ACT_RELU) {
          g = (pre ? pre[i] > 0.f : H[i] > 0.f) ? g : 0.f;        // H = relu(pre)*keep/(1-p) > 0 exactly where pre > 0
        } else if (act != MHIMX_ACT_NONE) {
          const float x = pre[i];
          g *= act_grad(x, act == MHIMX_ACT_TANH ? tanhf(x) : 0.f, act);
        }
      }
      dH[i] = g;
      colsum += g;
    }
    if (part) part[(int64_t)blockIdx.x * E + e] = colsum;
  }
}

// dH *= dact (float4 per thread, 128 threads per 512-wide row) + per-block column sums
__global__ __launch_bounds__(256) void mul_colsum_kernel(float* __restrict__ dH, const float* __restrict__ dact, int64_t M, int E,
                                                        int64_t chunk, float* __restrict__ part) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  const int e4 = E / 4;
  const int64_t mb = (int64_t)blockIdx.x * chunk;
  const int64_t me = mb + chunk < M ? mb + chunk : M;
  for (int c = threadIdx.x; c < e4; c += 256) {
    f4v cs = f4v{0.f, 0.f, 0.f, 0.f};
    for (int64_t m = mb; m < me; ++m) {
      f4v* p = reinterpret_cast<f4v*>(dH + m * E) + c;
      const f4v g = *p * reinterpret_cast<const f4v*>(dact + m * E)[c];
      *p = g;
      cs += g;
    }
    if (part) *(reinterpret_cast<f4v*>(part + (int64_t)blockIdx.x * E) + c) = cs;
  }
}

// dpre[p,:] = dH[rows[p],:] * dact16[rows[p],:] (fp16 d out / d pre from bag_project) -> compact [L,E] + per-block column sums:
// the backward through activation + dropout of the rows that took part in the step, gathered out of the bag-ordered buffers
__global__ __launch_bounds__(256) void rows_dpre_kernel(const float* __restrict__ dH, const _Float16* __restrict__ dact,
                                                       const int64_t* __restrict__ rows, int64_t L, int E, int64_t chunk,
                                                       float* __restrict__ dpre, float* __restrict__ part, int side_blocks, Merge2Side side) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  if ((int)blockIdx.x < side_blocks) {          // a parked Merge-backward tail rides along (stage 1), its workgroups first
    __shared__ float side_lds[M2_PARTIALS_LDS];
    merge2_side_stage(1, (int)blockIdx.x, side_lds, side);
    return;
  }
  const int bid = (int)blockIdx.x - side_blocks;
  typedef _Float16 h4v __attribute__((ext_vector_type(4)));
  const int e4 = E / 4;
  const int64_t mb = (int64_t)bid * chunk;
  const int64_t me = mb + chunk < L ? mb + chunk : L;
  for (int c = threadIdx.x; c < e4; c += 256) {
    f4v cs = f4v{0.f, 0.f, 0.f, 0.f};
    for (int64_t m = mb; m < me; ++m) {
      const int64_t r = rows ? rows[m] : m;
      const h4v d = reinterpret_cast<const h4v*>(dact + r * E)[c];
      const f4v g = reinterpret_cast<const f4v*>(dH + r * E)[c] * f4v{(float)d[0], (float)d[1], (float)d[2], (float)d[3]};
      reinterpret_cast<f4v*>(dpre + m * E)[c] = g;
      cs += g;
    }
    if (part) *(reinterpret_cast<f4v*>(part + (int64_t)bid * E) + c) = cs;
  }
}

// per-column max and arg-max (lowest row on ties) of x[M,C], C <= 16: DSMIL's critical instance per class
// (baseline.py:139-140) and its max-instance logits (:172).  One block.
__global__ __launch_bounds__(1024) void colmax_kernel(const float* __restrict__ x, int64_t M, int C, float* __restrict__ vals,
                                                      int64_t* __restrict__ idx) {
  __shared__ float bv[16][16];
  __shared__ int64_t bi[16][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = 0; c < C; ++c) {
    float best = -INFINITY;
    int64_t arg = 0;
    for (int64_t m = threadIdx.x; m < M; m += 1024) {
      const float v = x[m * C + c];
      if (v > best) { best = v; arg = m; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int64_t oi = __shfl_xor(arg, o, 64);
      if (ov > best || (ov == best && oi < arg)) { best = ov; arg = oi; }
    }
    if (lane == 0) { bv[c][wave] = best; bi[c][wave] = arg; }
  }
  __syncthreads();
  if (threadIdx.x < C) {
    const int c = threadIdx.x;
    float best = bv[c][0];
    int64_t arg = bi[c][0];
    for (int w = 1; w < 16; ++w)
      if (bv[c][w] > best || (bv[c][w] == best && bi[c][w] < arg)) { best = bv[c][w]; arg = bi[c][w]; }
    vals[c] = best;
    idx[c] = arg;
  }
}
// out[m] = max_c x[m,c]  (DSMIL's instance score with cls_attn, baseline.py:176)
__global__ void rowmax_kernel(const float* __restrict__ x, int64_t M, int C, float* __restrict__ out) {
  for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
    float v = x[m * C];
    for (int c = 1; c < C; ++c) v = fmaxf(v, x[m * C + c]);
    out[m] = v;
  }
}

__global__ void colsum_part_kernel(const float* __restrict__ X, int64_t M, int E, int64_t chunk, float* __restrict__ part) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t mb = (int64_t)blockIdx.y * chunk;
  const int64_t me = mb + chunk < M ? mb + chunk : M;
  float acc = 0.f;
#pragma unroll 8
  for (int64_t m = mb; m < me; ++m) acc += X[m * (int64_t)E + e];
  part[(int64_t)blockIdx.y * E + e] = acc;
}

__global__ void compose_ids_kernel(const int64_t* __restrict__ a, const int64_t* __restrict__ b, int64_t* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = a[b[i]];
}

// =================================================================================================
// host side
// =================================================================================================
static int grid_for_ln(int64_t M) {       // LayerNorm: one row per wave, no partial-count limit issues (<= 512 blocks)
  int64_t g = cdiv(M, 4);
  if (g < 1) g = 1;
  if (g > MAX_PART) g = MAX_PART;
  return (int)g;
}

static int grid_for_rows(int64_t M) {
  int64_t g = cdiv(M, 20);              // 5 rows per wave: the per-row chain (2 KB read, dot, exp) is latency bound
  if (g < 1) g = 1;
  if (g > MAX_PART) g = MAX_PART;
  return (int)g;
}

struct PoolWs {
  float *pm, *pl, *pz, *attn, *du, *dwc_part, *dbc_part, *u_pre, *tn_ws, *nt_ws;
};

static int64_t pool_ws_layout(Arena& ar, int64_t M, int64_t E, int64_t A, int gated, PoolWs* w) {
  const int64_t G = 2 * MAX_PART;
  PoolWs t;
  t.pm = ar.take<float>(G);
  t.pl = ar.take<float>(G);
  t.pz = ar.take<float>(G * E);
  t.attn = ar.take<float>(M);
  t.du = ar.take<float>(M * A * (1 + gated));
  t.dwc_part = ar.take<float>(G * A);
  t.dbc_part = ar.take<float>(G);
  t.u_pre = ar.take<float>(M * A * (1 + gated));
  t.tn_ws = ar.take<float>((int64_t)TN_SLABS * A * E);
  t.nt_ws = ar.take<float>(4 * M * A);                  // split-K slabs of the scorer GEMM (79 tiles at N = 10 000)
  if (w) *w = t;
  return ar.off;
}

template <typename F>
static int dispatch_epl(int64_t E, F&& f) {
  switch (E) {
    case 256: return f(std::integral_constant<int, 4>());
    case 512: return f(std::integral_constant<int, 8>());
    case 1024: return f(std::integral_constant<int, 16>());
    default: return fail(-1, "token width E=%lld unsupported (256, 512 or 1024)", (long long)E);
  }
}

static int check_scorer(const mhimx_scorer* sc) {
  MHIMX_CHECK_ARG(sc && sc->wa && sc->wc, "scorer: null weights");
  MHIMX_CHECK_ARG(sc->A > 0 && sc->A <= 512 && sc->A % 4 == 0, "scorer: A must be a multiple of 4, <= 512");
  MHIMX_CHECK_ARG(!sc->gated || sc->wb, "scorer: gated needs wb");
  return 0;
}

int abmil_pool_fwd(hipStream_t st, const mhimx_scorer* sc, mhimx_pool_io* io) {
  if (int r = check_scorer(sc)) return r;
  MHIMX_CHECK_ARG(io && io->T1 && io->M1 > 0 && io->s && io->stats && io->z, "pool_fwd: null io");
  MHIMX_CHECK_ARG(io->M2 == 0 || io->T2, "pool_fwd: M2>0 needs T2");
  MHIMX_CHECK_ARG(!io->cproj || (io->wp && io->C > 0), "pool_fwd: cproj needs wp, C");
  MHIMX_CHECK_ARG(!io->rows1 || (io->M2 == 0 && scorer_fused_ok(sc->E, sc->A, sc->gated, sc->prec, io->T1, sc->wa, io->cproj ? io->wp : nullptr, io->C)),
                  "pool_fwd: gathered tokens (rows1) need the one-pass scorer (E = 512, A = 128, plain form, not f32) and a single segment");
  MHIMX_CHECK_ARG(!io->excl || scorer_fused_ok(sc->E, sc->A, sc->gated, sc->prec, io->T1, sc->wa, io->cproj ? io->wp : nullptr, io->C),
                  "pool_fwd: excluded rows (excl) need the one-pass scorer (E = 512, A = 128, plain form, not f32)");
  const int64_t M = io->M1 + io->M2, E = sc->E, A = sc->A;
  const int gated = sc->gated ? 1 : 0;
  Arena ar(io->ws, io->ws_bytes);
  PoolWs w;
  pool_ws_layout(ar, M, E, A, gated, &w);
  MHIMX_CHECK_ARG(ar.ok(), "pool_fwd: workspace too small (%lld < %lld)", (long long)io->ws_bytes, (long long)ar.off);
  float* u_pre = io->u_pre ? io->u_pre : w.u_pre;
  const int64_t ldu = A * (1 + gated);
  if (io->phase != 0) {
    // the forward in two calls around the launches that produce the last tail_tokens tokens (include/mhimx.h)
    const int64_t K = io->tail_tokens, Ms1 = io->M1 - K;
    MHIMX_CHECK_ARG((io->phase == 1 || io->phase == 2) && K >= 1 && K <= FIN_MAXTOK && Ms1 >= 1 && io->M2 == 0 && !io->cproj && !io->pscore && !io->excl &&
                        E == 512 && A == 128 && scorer_fused_ok(E, A, gated, sc->prec, io->T1, sc->wa, nullptr, 0),
                    "pool_fwd: the two-call forward takes the one-pass scorer shapes, one segment, 1..%d tail tokens, no class projections", FIN_MAXTOK);
    const int tiles = (int)cdiv(Ms1, 32);
    const int G1 = tiles < MAX_PART ? tiles : MAX_PART;
    if (io->phase == 1) {
      M2RowsFwd mfbuf;
      const void* mf = nullptr;
      io->rode_merge = 0;
      static const bool ride_ok = getenv("MHIMX_MERGE_FWD_RIDE") == nullptr || atoi(getenv("MHIMX_MERGE_FWD_RIDE")) != 0;
      if (io->ride_merge && ride_ok) {
        const int rc = merge2_fwd_rows_args(reinterpret_cast<const mhimx_merge*>(io->ride_merge), io->ride_X, io->ride_R, io->ride_ws, io->ride_ws_bytes,
                                            &mfbuf);
        if (rc < 0) return rc;
        if (rc == 0) { mf = &mfbuf; io->rode_merge = 1; }
      }
      const int g1 = scorer_fused_fwd(st, io->T1, Ms1, sc->wa, sc->wa_frag, sc->ba, sc->act, sc->wc, sc->bc, nullptr, 0, u_pre, io->s, nullptr, w.pm, w.pl,
                                      w.pz, MAX_PART, io->rows1, nullptr, io->ride_jobs, io->n_ride_jobs, mf);
      return g1 < 0 ? g1 : 0;
    }
    hipLaunchKernelGGL(pool_finalize_tok_kernel, bgrid((unsigned)cdiv(E, 64)), dim3(FIN_THREADS), 0, st, w.pm, w.pl, w.pz, G1, (int)E, io->stats, io->z, io->T1,
                       (io->rows1 && io->tail_row0 < 0) ? io->rows1 + Ms1 : nullptr, io->rows1 ? io->tail_row0 : Ms1, (int)K, sc->wa, io->tail_wa_t, sc->ba, sc->wc,
                       sc->bc, (int)A, sc->act, u_pre + Ms1 * ldu, io->s + Ms1, cur_batch());
    MHIMX_LAUNCH_CHECK();
    return 0;
  }

  const float* Ts[2] = {io->T1, io->T2};
  const int64_t Ms[2] = {io->M1, io->M2};
  int64_t off = 0;
  int G = 0;
  bool rode = false;                            // (io->ride_jobs go with the first one-pass scorer launch; no such launch: a launch of their own)
  for (int seg = 0; seg < 2; ++seg) {
    if (Ms[seg] == 0) continue;
    if (scorer_fused_ok(E, A, gated, sc->prec, Ts[seg], sc->wa, io->cproj ? io->wp : nullptr, io->C)) {
      // one pass over the rows: GEMM + scores + class projections + pool partials (scorer_fused.hip)
      const int g1 = scorer_fused_fwd(st, Ts[seg], Ms[seg], sc->wa, sc->wa_frag, sc->ba, sc->act, sc->wc, sc->bc, io->cproj ? io->wp : nullptr,
                                      (int)io->C, io->no_backward ? nullptr : u_pre + off * ldu, io->s + off, io->cproj ? io->cproj + off * io->C : nullptr,
                                      w.pm + G, w.pl + G, w.pz + (int64_t)G * E, MAX_PART, seg == 0 ? io->rows1 : nullptr,
                                      seg == 0 ? io->excl : nullptr, rode ? nullptr : io->ride_jobs, rode ? 0 : io->n_ride_jobs, nullptr);
      if (g1 < 0) return g1;
      rode = true;
      G += g1;
      off += Ms[seg];
      continue;
    }
    MHIMX_CHECK_ARG(cur_batch().n == 0, "pool_fwd: a bag-batched launch needs the one-pass scorer");
    mhimx_gemm_nt_args g = {};
    g.A = Ts[seg]; g.lda = E; g.B = sc->wa; g.ldb = E; g.C = u_pre + off * ldu; g.ldc = ldu;
    g.M = Ms[seg]; g.N = A; g.K = E; g.bias = sc->ba;
    g.ws = w.nt_ws; g.ws_floats = 4 * M * A;
    // instance-level scores feed the top-k: keep them at ~fp32 accuracy (3-term bf16) unless exact f32 was asked for
    g.prec = sc->prec == MHIMX_PREC_F32 ? MHIMX_PREC_F32 : MHIMX_PREC_BF16X3;
    if (int r = gemm_nt(st, g)) return r;
    if (gated) {
      g.B = sc->wb; g.bias = sc->bb; g.C = u_pre + off * ldu + A;
      if (int r = gemm_nt(st, g)) return r;
    }
    int grid = (int)cdiv(Ms[seg], 5 * FWD_WAVES);        // 5 rows per wave: the per-row chain (2 KB read, dot, exp) is latency bound
    if (grid < 1) grid = 1;
    if (grid > MAX_PART) grid = MAX_PART;
    const float* Tseg = Ts[seg];
    const int64_t Mseg = Ms[seg];
    int r = dispatch_epl(E, [&](auto epl) {
      constexpr int EPL = decltype(epl)::value;
      const size_t smem = (size_t)(FWD_WAVES * E + 2 * FWD_WAVES) * sizeof(float);
      if (smem > 48 * 1024)                                  // one flag set per EPL instantiation
        MHIMX_ONCE_PER_DEVICE(MHIMX_HIP(hipFuncSetAttribute((const void*)score_rows_fwd_kernel<EPL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)));
      hipLaunchKernelGGL(score_rows_fwd_kernel<EPL>, dim3(grid), dim3(64 * FWD_WAVES), smem, st, Tseg, Mseg, (int)E, (int)A,
                         sc->act, gated, u_pre + off * ldu, sc->wc, sc->bc, io->cproj ? io->wp : nullptr, (int)io->C,
                         io->s + off, io->cproj ? io->cproj + off * io->C : nullptr, w.pm + G, w.pl + G,
                         w.pz + (int64_t)G * E, sc->gate_drop_p, sc->gate_drop_seed, off);
      MHIMX_LAUNCH_CHECK();
      return 0;
    });
    if (r) return r;
    G += grid;
    off += Ms[seg];
  }
  if (!rode && io->ride_jobs && io->n_ride_jobs > 0)
    if (int r = prep_batch(st, io->ride_jobs, io->n_ride_jobs)) return r;
  MHIMX_CHECK_ARG(!io->pscore || io->cproj, "pool_fwd: pscore needs cproj (and wp)");
  static_assert(2 * MAX_PART <= FIN_THREADS, "pool_finalize_kernel reads one partial per thread");
  // the pseudo score of the instances (if asked for) on blocks of its own, beside the E/64 blocks that merge the pooled row
  const int64_t ps_blocks = io->pscore ? (cdiv(io->M1, FIN_THREADS) < 64 ? cdiv(io->M1, FIN_THREADS) : 64) : 0;
  hipLaunchKernelGGL(pool_finalize_kernel, bgrid((unsigned)(cdiv(E, 64) + ps_blocks)), dim3(FIN_THREADS), 0, st, w.pm, w.pl, w.pz, G, (int)E, io->stats,
                     io->z, io->s, io->cproj, io->bp, (int)io->C, io->M1, io->pscore, cur_batch());
  MHIMX_LAUNCH_CHECK();
  return 0;
}

int abmil_pool_bwd(hipStream_t st, const mhimx_scorer* sc, const mhimx_pool_io* io, const mhimx_pool_grad* gr) {
  if (int r = check_scorer(sc)) return r;
  MHIMX_CHECK_ARG(io && gr && gr->g_z && gr->dT1 && gr->d_wa && gr->d_wc && gr->wa_t, "pool_bwd: null args");
  MHIMX_CHECK_ARG(io->M2 == 0 || gr->dT2, "pool_bwd: M2>0 needs dT2");
  const int64_t M = io->M1 + io->M2, E = sc->E, A = sc->A;
  const int gated = sc->gated ? 1 : 0;
  MHIMX_CHECK_ARG(!gated || (gr->d_wb && gr->wb_t), "pool_bwd: gated needs d_wb, wb_t");
  Arena ar(io->ws, io->ws_bytes);
  PoolWs w;
  pool_ws_layout(ar, M, E, A, gated, &w);
  MHIMX_CHECK_ARG(ar.ok(), "pool_bwd: workspace too small");
  const float* u_pre = io->u_pre ? io->u_pre : w.u_pre;
  const int64_t ldu = A * (1 + gated);
  const float* Ts[2] = {io->T1, io->T2};
  float* dTs[2] = {gr->dT1, gr->dT2};
  const int64_t Ms[2] = {io->M1, io->M2};

  int64_t off = 0;
  int G = 0;
  // one pass over the rows for ds / du / dT / the d_wc, d_bc partials (scorer_fused.hip) where the shapes allow
  const bool fused = scorer_fused_ok(E, A, gated, sc->prec, io->T1, gr->wa_t, nullptr, 0) && aligned16(gr->dT1) &&
                     (io->M2 == 0 || (aligned16(io->T2) && aligned16(gr->dT2))) && aligned16(u_pre) && aligned16(w.du);
  MHIMX_CHECK_ARG(!io->rows1 || (fused && io->M2 == 0), "pool_bwd: gathered tokens (rows1) need the one-pass backward and a single segment");
  MHIMX_CHECK_ARG(!gr->img || (fused && io->M2 == 0 && E == 512), "pool_bwd: the dPRE image (img) needs the one-pass backward and a single segment");
  MHIMX_CHECK_ARG(cur_batch().n == 0 || (fused && io->M2 == 0 && !gated && gr->defer && !gr->d_bc && !gr->d_ba && !gr->d_bb),
                  "pool_bwd: a bag-batched launch needs the one-pass backward with its reductions queued");
  for (int seg = 0; seg < 2 && fused; ++seg) {
    if (Ms[seg] == 0) continue;
    // a Merge backward's first stage parked on the list (mhimx_merge_bwd_park) rides in this launch when its dz is a block of whole rows of
    // THIS launch's gradient buffer (the k token rows behind a bag's feature gradients)
    const void* pre_side = nullptr;
    int64_t gate_row0 = -1;
    if (seg == 0 && io->M2 == 0 && gr->defer && gr->defer->pre.pending == 1) {
      Merge2Side psd;
      memcpy(&psd, gr->defer->pre.blob, sizeof(psd));
      const int64_t d = psd.dz - dTs[0];
      if (psd.dz >= dTs[0] && d % E == 0) {
        pre_side = gr->defer->pre.blob;
        gate_row0 = d / E;
        gr->defer->pre.pending = 2;
      }
    }
    const int g1 = scorer_fused_bwd(st, Ts[seg], Ms[seg], u_pre + off * ldu, io->s + off, io->stats, gr->g_z, io->z, sc->wc, sc->act,
                                    gr->wa_t, gr->wa_t_frag, w.du + off * ldu, dTs[seg], w.dwc_part + (int64_t)G * A, w.dbc_part + G,
                                    MAX_PART, seg == 0 ? io->rows1 : nullptr, pre_side, gate_row0, seg == 0 ? gr->img : nullptr, gr->img_dact, gr->img_part,
                                    gr->img_rows);
    if (g1 < 0) return g1;
    G += g1;
    off += Ms[seg];
  }
  for (int seg = 0; seg < 2 && !fused; ++seg) {
    if (Ms[seg] == 0) continue;
    const int grid = grid_for_rows(Ms[seg]);
    const float* Tseg = Ts[seg];
    const int64_t Mseg = Ms[seg];
    int r = dispatch_epl(E, [&](auto epl) {
      constexpr int EPL = decltype(epl)::value;
      const size_t smem = (size_t)(4 * A + 8) * sizeof(float);
      hipLaunchKernelGGL(score_rows_bwd_kernel<EPL>, dim3(grid), dim3(ROWS_THREADS), smem, st, Tseg, Mseg, (int)E, (int)A,
                         sc->act, gated, u_pre + off * ldu, sc->wc, io->s + off, io->stats, gr->g_z, io->z,
                         w.du + off * ldu, w.attn + off, w.dwc_part + (int64_t)G * A, w.dbc_part + G, sc->gate_drop_p,
                         sc->gate_drop_seed, off);
      MHIMX_LAUNCH_CHECK();
      return 0;
    });
    if (r) return r;
    G += grid;
    off += Ms[seg];
  }
  if (!defer_push(gr->defer, reduce_job_parts(w.dwc_part, G, A, A, gr->d_wc, gr->accumulate)))
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)cdiv(A, 32)), dim3(RP_THREADS), 0, st, w.dwc_part, G, (int)A, (int)A, gr->d_wc, gr->accumulate);
  MHIMX_LAUNCH_CHECK();
  if (gr->d_bc) {
    if (!defer_push(gr->defer, reduce_job_parts(w.dbc_part, G, 1, 1, gr->d_bc, gr->accumulate)))
      hipLaunchKernelGGL(reduce_parts_kernel, dim3(1), dim3(RP_THREADS), 0, st, w.dbc_part, G, 1, 1, gr->d_bc, gr->accumulate);
    MHIMX_LAUNCH_CHECK();
  }
  off = 0;
  for (int seg = 0; seg < 2; ++seg) {
    if (Ms[seg] == 0) continue;
    // dT = du_a . Wa + attn (x) g_z  [+ du_b . Wb]
    mhimx_gemm_nt_args g = {};
    g.A = w.du + off * ldu; g.lda = ldu; g.B = gr->wa_t; g.ldb = A; g.C = dTs[seg]; g.ldc = E;
    g.M = Ms[seg]; g.N = E; g.K = A; g.rowv = w.attn + off; g.colv = gr->g_z; g.prec = sc->prec == MHIMX_PREC_F32 ? MHIMX_PREC_F32 : MHIMX_PREC_BF16X3;
    if (!fused)
      if (int r = gemm_nt(st, g)) return r;
    if (gated) {
      g.A = w.du + off * ldu + A; g.B = gr->wb_t; g.rowv = nullptr; g.colv = nullptr; g.accumulate = 1;
      if (int r = gemm_nt(st, g)) return r;
    }
    // d_wa (+)= du_a^T T ; d_wb likewise
    mhimx_gemm_tn_args t = {};
    t.A = w.du + off * ldu; t.lda = ldu; t.B = Ts[seg]; t.ldb = E; t.C = gr->d_wa; t.ldc = E;
    t.rows = seg == 0 ? io->rows1 : nullptr;
    t.M = Ms[seg]; t.K1 = A; t.K2 = E; t.splits = 1; t.accumulate = (gr->accumulate || seg > 0) ? 1 : 0;
    t.prec = sc->prec == MHIMX_PREC_F32 ? MHIMX_PREC_F32 : MHIMX_PREC_BF16X3;
    // split the long reduction (the library raises the slab count until the launch fills the chip, up to the workspace)
    int splits = gr->splits > 1 ? gr->splits : 1;
    if (splits > TN_SLABS) splits = TN_SLABS;
    if (Ms[seg] < 65536 && splits > 64) splits = 64;
    if (Ms[seg] < 2048) splits = 1;
    // (the library raises the slab count up to the workspace it is shown: 128 slabs pay at c5 - 200 000 rows, the row table of a slab
    // shrinks to 12 KiB and two workgroups share a CU, 3.96 -> 3.92 ms per step - and cost 2 us at c2, so short bags are shown 64)
    t.splits = splits; t.ws = w.tn_ws; t.ws_floats = (int64_t)(Ms[seg] >= 65536 ? TN_SLABS : 64) * A * E;
    t.defer = (io->M2 == 0 && !gated) ? gr->defer : nullptr;        // one GEMM per workspace: its slabs may wait for the flush
    static_assert(sizeof(mhimx_gemm_tn_args) <= sizeof(((mhimx_parked_gemm*)nullptr)->blob), "mhimx_parked_gemm.blob too small");
    if (t.defer && fused && !t.defer->parked.pending) {
      // nothing on the data path reads d_wa: the launch waits in the list for the Merge backward (which gives its first stage a ride in
      // it) or for mhimx_reduce_flush
      memcpy(t.defer->parked.blob, &t, sizeof(t));
      t.defer->parked.pending = 1;
    } else if (int r = gemm_tn(st, t)) return r;
    if (gated) {
      t.A = w.du + off * ldu + A; t.C = gr->d_wb;
      if (int r = gemm_tn(st, t)) return r;
    }
    off += Ms[seg];
  }
  // bias gradients of the standalone (biased) scorers: column sums of du
  if (gr->d_ba || gr->d_bb) {
    const int64_t W = ldu;
    const int64_t chunk = cdiv(M, 64);
    const int gy = (int)cdiv(M, chunk);
    hipLaunchKernelGGL(colsum_part_kernel, dim3((unsigned)cdiv(W, 128), gy), dim3(128), 0, st, w.du, M, (int)W, chunk, w.pz);
    MHIMX_LAUNCH_CHECK();
    if (gr->d_ba) {
      hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)cdiv(A, 32)), dim3(RP_THREADS), 0, st, w.pz, gy, (int)A, (int)W, gr->d_ba, gr->accumulate);
      MHIMX_LAUNCH_CHECK();
    }
    if (gr->d_bb && gated) {
      // columns [A, 2A): strided view handled by offsetting the partial pointer (row pitch W)
      hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)cdiv(A, 32)), dim3(RP_THREADS), 0, st, w.pz + A, gy, (int)A, (int)W, gr->d_bb, gr->accumulate);
      MHIMX_LAUNCH_CHECK();
    }
  }
  return 0;
}

int layernorm_fwd(hipStream_t st, const float* x, int64_t M, int64_t E, const float* w, const float* b, float* y,
                  float* mean, float* rstd) {
  MHIMX_CHECK_ARG(E <= 1024 && E % 64 == 0, "layernorm: E must be a multiple of 64, <= 1024");
  if (M == 0) return 0;
  if (E == 512 && M >= 64 && aligned16(x) && aligned16(y) && aligned16(w) && aligned16(b)) {
    const int64_t g = cdiv(M, 8) < 8192 ? cdiv(M, 8) : 8192;
    hipLaunchKernelGGL(layernorm_fwd_vec_kernel<2>, dim3((unsigned)g), dim3(ROWS_THREADS), 0, st, x, M, w, b, y, mean, rstd);
    MHIMX_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(grid_for_ln(M)), dim3(ROWS_THREADS), 0, st, x, M, (int)E, w, b, y, mean, rstd);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

// partials: dw_part/db_part [grid][E] scratch; d_w/d_b (+)= reduced
int layernorm_bwd(hipStream_t st, const float* dy, const float* x, int64_t M, int64_t E, const float* w, const float* mean,
                  const float* rstd, float* dx, float* dw_part, float* db_part, float* d_w, float* d_b, int accumulate,
                  int max_parts, const float* dy2, const float* x2, const float* mean2, const float* rstd2, int64_t M2,
                  mhimx_reduce_list* defer, const int64_t* xrows = nullptr, const float* resid = nullptr) {
  MHIMX_CHECK_ARG(E <= 1024 && E % 64 == 0, "layernorm: E must be a multiple of 64, <= 1024");
  if (M == 0) return 0;
  if (M2 > 0 && M > 16) {                   // rows + a few extra rows (weight gradients only) in one launch
    int grid = (int)cdiv(M, 4);
    if (grid > max_parts - 1) grid = max_parts - 1;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(grid + 1), dim3(ROWS_THREADS), (size_t)8 * E * sizeof(float), st, dy, x, M, (int)E,
                       w, mean, rstd, dx, dw_part, db_part, -1, LnSeg2{dy2, x2, mean2, rstd2, M2}, xrows, resid);
    MHIMX_LAUNCH_CHECK();
    if (defer && defer->n + 2 <= MHIMX_REDUCE_MAX) {
      defer_push(defer, reduce_job_parts(dw_part, grid + 1, E, E, d_w, accumulate));
      defer_push(defer, reduce_job_parts(db_part, grid + 1, E, E, d_b, accumulate));
      return 0;
    }
    hipLaunchKernelGGL(reduce_parts2_kernel, dim3((unsigned)cdiv(E, 32), 2), dim3(RP_THREADS), 0, st, dw_part, db_part, grid + 1, (int)E, (int)E,
                       d_w, d_b, accumulate);
    MHIMX_LAUNCH_CHECK();
    return 0;
  }
  MHIMX_CHECK_ARG(M2 == 0, "layernorm_bwd: the ride-along rows need a main segment of more than 16 rows");
  // one wave per row, 4 rows per block pass: the row loop is a latency chain, so use as many blocks as the partial
  // workspace (max_parts rows of dw / db partials) allows
  int grid = (int)cdiv(M, 4);
  if (grid < 1) grid = 1;
  if (grid > max_parts) grid = max_parts;
  if (M <= 16) {                            // a few rows (the k global queries): ONE block writes d_w / d_b itself
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(1), dim3(ROWS_THREADS), (size_t)8 * E * sizeof(float), st, dy, x, M, (int)E, w, mean,
                       rstd, dx, d_w, d_b, accumulate ? 1 : 0, LnSeg2{nullptr, nullptr, nullptr, nullptr, 0}, xrows, resid);
    MHIMX_LAUNCH_CHECK();
    return 0;
  }
  if (E == 512 && !xrows && aligned16(dy) && aligned16(x) && aligned16(w) && (!dx || aligned16(dx)) && (!resid || aligned16(resid)))
    hipLaunchKernelGGL(layernorm_bwd_vec512_kernel, dim3(grid), dim3(ROWS_THREADS), (size_t)8 * E * sizeof(float), st, dy, x, M, w, mean, rstd, dx,
                       dw_part, db_part, resid);
  else
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(grid), dim3(ROWS_THREADS), (size_t)8 * E * sizeof(float), st, dy, x, M, (int)E,
                       w, mean, rstd, dx, dw_part, db_part, -1, LnSeg2{nullptr, nullptr, nullptr, nullptr, 0}, xrows, resid);
  MHIMX_LAUNCH_CHECK();
  if (defer && defer->n + 2 <= MHIMX_REDUCE_MAX) {
    defer_push(defer, reduce_job_parts(dw_part, grid, E, E, d_w, accumulate));
    defer_push(defer, reduce_job_parts(db_part, grid, E, E, d_b, accumulate));
    return 0;
  }
  hipLaunchKernelGGL(reduce_parts2_kernel, dim3((unsigned)cdiv(E, 32), 2), dim3(RP_THREADS), 0, st, dw_part, db_part, grid, (int)E, (int)E,
                     d_w, d_b, accumulate);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

int reduce_parts2(hipStream_t st, const float* part0, const float* part1, int G, int W, int ld, float* out0, float* out1, int accumulate) {
  hipLaunchKernelGGL(reduce_parts2_kernel, dim3((unsigned)cdiv(W, 32), 2), dim3(RP_THREADS), 0, st, part0, part1, G, W, ld, out0, out1, accumulate);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

int merge2_side_finish(hipStream_t st, mhimx_side_work* side, int upto_stage);      // mca2.hip
int gemm_tn_rider(hipStream_t st, const mhimx_gemm_tn_args& g, const Merge2Side* rider, int stage);      // gemm.hip

int reduce_flush(hipStream_t st, mhimx_reduce_list* list) {
  MHIMX_CHECK_ARG(list && list->n >= 0 && list->n <= MHIMX_REDUCE_MAX, "reduce_flush: bad list");
  if (list->parked.pending) {                                   // a GEMM nobody gave a launch to (no Merge backward followed)
    mhimx_gemm_tn_args pg;
    memcpy(&pg, list->parked.blob, sizeof(pg));
    list->parked.pending = 0;
    if (list->parked.reserved > 0) {                            // (round 6, the step as a DAG: sized as beside `reserved - 1` Merge row tiles)
      Merge2Side sz = {};
      sz.w.T = list->parked.reserved - 1;
      list->parked.reserved = 0;
      const int rc = gemm_tn_rider(st, pg, &sz, 5);
      if (rc < 0) return rc;
    } else if (int r = gemm_tn(st, pg)) return r;               // (queues its slab sum on this same list)
  }
  // a parked Merge-backward tail: the stages that found no ride run now, in order; the last one may ride in the reduction launch
  if (int r = merge2_side_finish(st, &list->side, list->n == 0 ? 3 : 2)) return r;
  if (list->n == 0) return 0;
  ReduceJobs rj;
  rj.side_blocks = 0;
  if (list->side.pending == 3) {
    memcpy(&rj.side, list->side.blob, sizeof(rj.side));
    rj.side_blocks = merge2_side_blocks(3, rj.side);
    list->side.pending = 0;
  }
  for (int i = 0; i < list->n; ++i) {
    const mhimx_reduce_job& j = list->j[i];
    MHIMX_CHECK_ARG(j.parts && j.out && j.G > 0 && (j.kind == 0 ? j.W > 0 : (j.kind == 1 && j.K1 > 0 && j.K2 > 0)), "reduce_flush: bad job %d", i);
  }
  const int first = reduce_table_fill(rj.t, list->j, list->n, RP_THREADS);
  list->n = 0;
  hipLaunchKernelGGL(reduce_batch_kernel, bgrid((unsigned)(first + rj.side_blocks)), dim3(RP_THREADS), 0, st, rj, cur_batch());
  MHIMX_LAUNCH_CHECK();
  return 0;
}

int colsum(hipStream_t st, const float* X, int64_t M, int64_t E, float* out, int accumulate, void* ws, int64_t ws_bytes) {
  MHIMX_CHECK_ARG(X && out && E > 0 && M >= 0, "colsum: bad args");
  // row chunks of >= 16 rows, up to 512 of them if the workspace holds that many (128 x E floats are always enough): a thread's rows are
  // one serial chain of loads, the chunk partials are summed 32 at a time per column
  const int64_t nrows = M > 0 ? M : 1, avail = ws ? ws_bytes / (E * 4) : 0;
  int64_t want = cdiv(nrows, 16) < 512 ? cdiv(nrows, 16) : 512;
  MHIMX_CHECK_ARG(avail >= (want < 128 ? want : 128), "colsum: workspace too small (need %lld bytes)", (long long)((want < 128 ? want : 128) * E * 4));
  if (want > avail) want = avail;
  const int64_t chunk = cdiv(M > 0 ? M : 1, want);
  const int gy = (int)cdiv(M > 0 ? M : 1, chunk);
  hipLaunchKernelGGL(colsum_part_kernel, dim3((unsigned)cdiv(E, 128), gy), dim3(128), 0, st, X, M, (int)E, chunk, (float*)ws);
  MHIMX_LAUNCH_CHECK();
  hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)cdiv(E, 32)), dim3(RP_THREADS), 0, st, (const float*)ws, gy, (int)E, (int)E, out, accumulate);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

}  // namespace mhimx

using namespace mhimx;

extern "C" int64_t mhimx_abmil_pool_ws_bytes(int64_t M, int64_t E, int64_t A, int32_t gated) {
  Arena ar(nullptr, 0);
  return pool_ws_layout(ar, M, E, A, gated ? 1 : 0, nullptr);
}
extern "C" int mhimx_abmil_pool_fwd(void* stream, const mhimx_scorer* sc, mhimx_pool_io* io) {
  return abmil_pool_fwd((hipStream_t)stream, sc, io);
}
extern "C" int mhimx_abmil_pool_bwd(void* stream, const mhimx_scorer* sc, const mhimx_pool_io* io, const mhimx_pool_grad* g) {
  return abmil_pool_bwd((hipStream_t)stream, sc, io, g);
}
extern "C" int mhimx_softmax_from_stats(void* stream, const float* s, const float* stats, float* attn, int64_t M) {
  if (M <= 0) return 0;
  hipLaunchKernelGGL(softmax_from_stats_kernel, bgrid((unsigned)(cdiv(M, 256) < 1024 ? cdiv(M, 256) : 1024)), dim3(256), 0,
                     (hipStream_t)stream, s, stats, attn, M, cur_batch());
  MHIMX_LAUNCH_CHECK();
  return 0;
}
extern "C" int mhimx_lse_merge(void* stream, const float* parts, int64_t W, int64_t E, float* stats, float* z) {
  MHIMX_CHECK_ARG(parts && stats && z && W >= 1 && E >= 1, "lse_merge: bad args");
  hipLaunchKernelGGL(lse_merge_kernel, dim3((unsigned)cdiv(E, 256)), dim3(256), 0, (hipStream_t)stream, parts, (int)W, (int)E, stats, z);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
extern "C" int mhimx_pseudo_score(void* stream, const float* s, const float* stats, const float* cproj, const float* bp,
                                  float* score, float* attn_out, int64_t M, int64_t C) {
  MHIMX_CHECK_ARG((!s || stats) && cproj && score && C > 0, "pseudo_score: null args");
  if (M <= 0) return 0;
  hipLaunchKernelGGL(pseudo_score_kernel, dim3((unsigned)(cdiv(M, 256) < 1024 ? cdiv(M, 256) : 1024)), dim3(256), 0,
                     (hipStream_t)stream, s, stats, cproj, bp, score, attn_out, M, (int)C);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
extern "C" int mhimx_act_bwd(void* stream, float* dH, const float* H, const float* pre, int64_t M, int64_t E, int32_t act,
                             float drop_p, uint64_t drop_seed, const uint8_t* drop_mask, const int64_t* rows, float* colsum_out,
                             int32_t accumulate, void* ws, int64_t ws_bytes, const uint64_t* drop_tick) {
  MHIMX_CHECK_ARG(dH && (H || pre), "act_bwd: null args");
  MHIMX_CHECK_ARG(act == MHIMX_ACT_NONE || act == MHIMX_ACT_RELU || pre, "act_bwd: gelu/tanh need the pre-activation");
  if (M <= 0) return 0;
  int64_t nblk = cdiv(M, 16);
  if (nblk > 1024) nblk = 1024;
  const int64_t chunk = cdiv(M, nblk);
  nblk = cdiv(M, chunk);
  MHIMX_CHECK_ARG(!colsum_out || (ws && ws_bytes >= nblk * E * 4), "act_bwd: workspace too small for the column sums (%lld bytes)",
                  (long long)(nblk * E * 4));
  hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, dH, H, pre, M, (int)E, act, drop_p,
                     drop_seed, drop_mask, rows, chunk, colsum_out ? (float*)ws : nullptr, drop_tick);
  MHIMX_LAUNCH_CHECK();
  if (colsum_out) {
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)cdiv(E, 32)), dim3(RP_THREADS), 0, (hipStream_t)stream, (const float*)ws, (int)nblk,
                       (int)E, (int)E, colsum_out, accumulate);
    MHIMX_LAUNCH_CHECK();
  }
  return 0;
}
extern "C" int mhimx_colmax(void* stream, const float* x, int64_t M, int64_t C, float* vals, int64_t* idx) {
  MHIMX_CHECK_ARG(x && vals && idx && M > 0 && C > 0 && C <= 16, "colmax: bad args (C <= 16)");
  hipLaunchKernelGGL(colmax_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, M, (int)C, vals, idx);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
extern "C" int mhimx_rowmax(void* stream, const float* x, int64_t M, int64_t C, float* out) {
  MHIMX_CHECK_ARG(x && out && M > 0 && C > 0, "rowmax: bad args");
  hipLaunchKernelGGL(rowmax_kernel, dim3((unsigned)(cdiv(M, 256) < 1024 ? cdiv(M, 256) : 1024)), dim3(256), 0, (hipStream_t)stream, x, M,
                     (int)C, out);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
extern "C" int mhimx_rows_dpre(void* stream, const float* dH, const void* dact16, const int64_t* rows, int64_t L, int64_t E, float* dpre,
                               float* colsum_out, int32_t accumulate, void* ws, int64_t ws_bytes, mhimx_reduce_list* defer) {
  MHIMX_CHECK_ARG(dH && dact16 && dpre && E % 4 == 0 && aligned16(dH) && aligned16(dpre) && (reinterpret_cast<uintptr_t>(dact16) & 7) == 0,
                  "rows_dpre: bad args");
  if (L <= 0) return 0;
  int64_t nblk = cdiv(L, 8);
  if (nblk > 512) nblk = 512;
  const int64_t chunk = cdiv(L, nblk);
  nblk = cdiv(L, chunk);
  MHIMX_CHECK_ARG(!colsum_out || (ws && ws_bytes >= nblk * E * 4), "rows_dpre: workspace too small (%lld bytes)", (long long)(nblk * E * 4));
  Merge2Side side = {};
  int side_blocks = 0;
  if (defer && defer->side.pending == 1) {       // a parked Merge-backward tail: stage 1 rides in this launch
    memcpy(&side, defer->side.blob, sizeof(side));
    side_blocks = merge2_side_blocks(1, side);
    defer->side.pending = 2;
  }
  hipLaunchKernelGGL(rows_dpre_kernel, dim3((unsigned)(nblk + side_blocks)), dim3(256), 0, (hipStream_t)stream, dH, (const _Float16*)dact16, rows, L,
                     (int)E, chunk, dpre, colsum_out ? (float*)ws : nullptr, side_blocks, side);
  MHIMX_LAUNCH_CHECK();
  if (colsum_out && !defer_push(defer, reduce_job_parts((const float*)ws, nblk, E, E, colsum_out, accumulate))) {
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)cdiv(E, 32)), dim3(RP_THREADS), 0, (hipStream_t)stream, (const float*)ws, (int)nblk,
                       (int)E, (int)E, colsum_out, accumulate);
    MHIMX_LAUNCH_CHECK();
  }
  return 0;
}
extern "C" int mhimx_mul_colsum(void* stream, float* dH, const float* dact, int64_t M, int64_t E, float* colsum_out, int32_t accumulate,
                                void* ws, int64_t ws_bytes, mhimx_reduce_list* defer) {
  MHIMX_CHECK_ARG(dH && dact && E % 4 == 0 && aligned16(dH) && aligned16(dact), "mul_colsum: bad args");
  if (M <= 0) return 0;
  int64_t nblk = cdiv(M, 8);
  if (nblk > 512) nblk = 512;
  const int64_t chunk = cdiv(M, nblk);
  nblk = cdiv(M, chunk);
  MHIMX_CHECK_ARG(!colsum_out || (ws && ws_bytes >= nblk * E * 4), "mul_colsum: workspace too small (%lld bytes)", (long long)(nblk * E * 4));
  hipLaunchKernelGGL(mul_colsum_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, dH, dact, M, (int)E, chunk,
                     colsum_out ? (float*)ws : nullptr);
  MHIMX_LAUNCH_CHECK();
  if (colsum_out && !defer_push(defer, reduce_job_parts((const float*)ws, nblk, E, E, colsum_out, accumulate))) {
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)cdiv(E, 32)), dim3(RP_THREADS), 0, (hipStream_t)stream, (const float*)ws, (int)nblk,
                       (int)E, (int)E, colsum_out, accumulate);
    MHIMX_LAUNCH_CHECK();
  }
  return 0;
}
// the pool's finalize launch on caller-held partials (the scored projection's: mhimx_proj_score) - the same kernel mhimx_abmil_pool_fwd ends with
extern "C" int mhimx_pool_finalize(void* stream, const float* pm, const float* pl, const float* pz, int64_t G, int64_t E, float* stats, float* z,
                                   const float* s, const float* cproj, const float* bp, int64_t C, int64_t M1, float* pscore) {
  MHIMX_CHECK_ARG(pm && pl && pz && stats && z && G >= 1 && G <= 2 * MAX_PART && E >= 1, "pool_finalize: 1..%d partials", 2 * MAX_PART);
  MHIMX_CHECK_ARG(!pscore || (s && cproj && C >= 1 && M1 >= 1), "pool_finalize: the pseudo score needs s, cproj and the instance count");
  const int64_t ps_blocks = pscore ? (cdiv(M1, FIN_THREADS) < 64 ? cdiv(M1, FIN_THREADS) : 64) : 0;
  hipLaunchKernelGGL(pool_finalize_kernel, dim3((unsigned)(cdiv(E, 64) + ps_blocks)), dim3(FIN_THREADS), 0, (hipStream_t)stream, pm, pl, pz, (int)G, (int)E,
                     stats, z, s, cproj, bp, (int)C, M1, pscore, BagBatch{});
  MHIMX_LAUNCH_CHECK();
  return 0;
}

#ifdef MHIMX_FT_PROF
extern "C" int mhimx_ft_prof_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mhimx::ft_prof), 16 * 8); }
#endif
extern "C" int mhimx_reduce_flush(void* stream, mhimx_reduce_list* list) { return reduce_flush((hipStream_t)stream, list); }
extern "C" int mhimx_colsum(void* stream, const float* X, int64_t M, int64_t E, float* out, int32_t accumulate, void* ws,
                            int64_t ws_bytes) {
  return colsum((hipStream_t)stream, X, M, E, out, accumulate, ws, ws_bytes);
}
// ws: 2*512*E floats (partial rows of the weight/bias gradients)
extern "C" int mhimx_layernorm_fwd(void* stream, const float* x, int64_t M, int64_t E, const float* w, const float* b, float* y,
                                   float* mean, float* rstd) {
  MHIMX_CHECK_ARG(x && w && b && y && mean && rstd, "layernorm_fwd: null args");
  return layernorm_fwd((hipStream_t)stream, x, M, E, w, b, y, mean, rstd);
}
extern "C" int mhimx_layernorm_bwd(void* stream, const float* dy, const float* x, int64_t M, int64_t E, const float* w, const float* mean,
                                   const float* rstd, float* dx, float* d_w, float* d_b, int32_t accumulate, float* ws) {
  MHIMX_CHECK_ARG(dy && x && w && mean && rstd && d_w && d_b && ws, "layernorm_bwd: null args");
  return layernorm_bwd((hipStream_t)stream, dy, x, M, E, w, mean, rstd, dx, ws, ws + 512 * E, d_w, d_b, accumulate, 512, nullptr, nullptr, nullptr,
                       nullptr, 0, nullptr);
}
extern "C" int mhimx_layernorm_bwd_res(void* stream, const float* dy, const float* x, int64_t M, int64_t E, const float* w, const float* mean,
                                       const float* rstd, const float* resid, float* dx, float* d_w, float* d_b, int32_t accumulate, float* ws) {
  MHIMX_CHECK_ARG(dy && x && w && mean && rstd && resid && dx && d_w && d_b && ws, "layernorm_bwd_res: null args");
  return layernorm_bwd((hipStream_t)stream, dy, x, M, E, w, mean, rstd, dx, ws, ws + 512 * E, d_w, d_b, accumulate, 512, nullptr, nullptr, nullptr,
                       nullptr, 0, nullptr, nullptr, resid);
}
// A keyed pseudo-random PERMUTATION of 0 .. n-1 computed per element - no sort, no generator state: a 6-round Feistel network on the
// smallest even-bit domain 2^b >= n (round function: one 32-bit mix of the half, the round and the key), cycle-walked into [0, n)
// (x = E(x) until x < n: a permutation of the domain restricted to [0, n) this way is a permutation of [0, n); fewer than 4 steps on
// average).  out[j] = pi(j), or src[pi(j)] with a source list: the random subsets of masking.py:67 / merge.py:165-170 (torch.randperm there)
// are the first entries of such a list.  torch.randperm is 13 rocprim launches (~60 us at n = 200 000).
__global__ __launch_bounds__(256) void random_perm_kernel(int64_t n, int bits, uint64_t seed0, const uint64_t* __restrict__ tick,
                                                         const int64_t* __restrict__ src, int64_t* __restrict__ out) {
  const uint64_t seed = eff_seed(seed0, tick);
  const uint32_t k0 = mix32((uint32_t)seed ^ 0x9E3779B9u), k1 = mix32((uint32_t)(seed >> 32) + 0x85EBCA6Bu + k0 * 0x632BE5ABu);   // (both keys: the whole seed)
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256) {
    const uint64_t x = feistel_index((uint64_t)j, (uint64_t)n, bits, k0, k1);
    out[j] = src ? src[x] : (int64_t)x;
  }
}
extern "C" int mhimx_random_perm(void* stream, int64_t n, uint64_t seed, const uint64_t* tick, const int64_t* src, int64_t* out) {
  if (n == 0) return 0;
  MHIMX_CHECK_ARG(out && n > 0 && n < ((int64_t)1 << 40) && out != src, "random_perm: 0 <= n < 2^40, out != src");
  int bits = 2;
  while (((int64_t)1 << bits) < n) bits += 2;
  hipLaunchKernelGGL(random_perm_kernel, dim3((unsigned)(cdiv(n, 256) < 2048 ? cdiv(n, 256) : 2048)), dim3(256), 0, (hipStream_t)stream, n, bits,
                     seed, tick, src, out);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
extern "C" int mhimx_compose_ids(void* stream, const int64_t* a, const int64_t* b, int64_t* out, int64_t n) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(compose_ids_kernel, dim3((unsigned)(cdiv(n, 256) < 1024 ? cdiv(n, 256) : 1024)), dim3(256), 0,
                     (hipStream_t)stream, a, b, out, n);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
