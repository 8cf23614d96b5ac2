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
                  const float* rstd, float* dx, float* dw_part, float* db_part, float* d_w, float* d_b, int accumulate);
int colsum(hipStream_t st, const float* X, int64_t M, int64_t E, float* out, int accumulate, void* ws, int64_t ws_bytes);

constexpr int MCA_THREADS = 256;

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

// grid (heads, k).  dim_head must be 64 (lane = column).  P[h,i,:] = softmax_r(scale * q_i . k_r)
__global__ __launch_bounds__(MCA_THREADS) void mca_attend_fwd_kernel(const float* __restrict__ KV, const float* __restrict__ Q,
                                                                     int64_t R, int heads, int kq, float scale,
                                                                     float drop_p, uint64_t seed, float* __restrict__ P,
                                                                     float* __restrict__ O) {
  __shared__ float red[4];
  __shared__ float osum[4][64];
  const int h = blockIdx.x, i = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int inner = heads * 64;
  const float q = Q[(int64_t)i * inner + h * 64 + lane];
  float* Prow = P + ((int64_t)h * kq + i) * R;
  float mx = -INFINITY;
  for (int64_t r = wave; r < R; r += 4) {
    const float d = wave_sum(q * KV[r * 2 * inner + h * 64 + lane]) * scale;
    if (lane == 0) Prow[r] = d;
    mx = fmaxf(mx, d);
  }
  mx = block_reduce_max(mx, red);          // includes the barrier that publishes Prow within the block
  __threadfence_block();
  float sum = 0.f;
  for (int64_t r = threadIdx.x; r < R; r += MCA_THREADS) {
    const float e = __expf(Prow[r] - mx);
    Prow[r] = e;
    sum += e;
  }
  sum = block_reduce_sum(sum, red);
  const float inv = 1.f / sum;
  for (int64_t r = threadIdx.x; r < R; r += MCA_THREADS) Prow[r] *= inv;
  __threadfence_block();
  __syncthreads();
  const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  float acc = 0.f;
  for (int64_t r = wave; r < R; r += 4) {
    float p = Prow[r];
    if (drop_p > 0.f) p = drop_keep(seed, (uint64_t)(h * kq + i), (uint32_t)r, drop_p) ? p * keep_scale : 0.f;
    acc += p * KV[r * 2 * inner + inner + h * 64 + lane];
  }
  osum[wave][lane] = acc;
  __syncthreads();
  if (wave == 0) O[(int64_t)i * inner + h * 64 + lane] = osum[0][lane] + osum[1][lane] + osum[2][lane] + osum[3][lane];
}

// grid (heads, k): dP_r = dO_i . v_r (dropout-scaled); dd[h,i,r] = P_r (dP_r - sum_r P_r dP_r)
__global__ __launch_bounds__(MCA_THREADS) void mca_dp_kernel(const float* __restrict__ KV, const float* __restrict__ dO,
                                                             const float* __restrict__ P, int64_t R, int heads, int kq,
                                                             float drop_p, uint64_t seed, float* __restrict__ dd) {
  __shared__ float red[4];
  const int h = blockIdx.x, i = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int inner = heads * 64;
  const float go = dO[(int64_t)i * inner + h * 64 + lane];
  const float* Prow = P + ((int64_t)h * kq + i) * R;
  float* drow = dd + ((int64_t)h * kq + i) * R;
  const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  float dot = 0.f;
  for (int64_t r = wave; r < R; r += 4) {
    float dp = wave_sum(go * KV[r * 2 * inner + inner + h * 64 + lane]);
    if (drop_p > 0.f) dp = drop_keep(seed, (uint64_t)(h * kq + i), (uint32_t)r, drop_p) ? dp * keep_scale : 0.f;
    if (lane == 0) drow[r] = dp;
    dot += (lane == 0) ? Prow[r] * dp : 0.f;
  }
  dot = block_reduce_sum(dot, red);
  __threadfence_block();
  __syncthreads();
  for (int64_t r = threadIdx.x; r < R; r += MCA_THREADS) drow[r] = Prow[r] * (drow[r] - dot);
}

// grid (heads, row blocks): dK[r] = scale sum_i dd[h,i,r] q_i ; dV[r] = sum_i Pd[h,i,r] dO_i
__global__ __launch_bounds__(MCA_THREADS) void mca_dkv_kernel(const float* __restrict__ Q, const float* __restrict__ dO,
                                                              const float* __restrict__ P, const float* __restrict__ dd,
                                                              int64_t R, int heads, int kq, float scale, float drop_p,
                                                              uint64_t seed, float* __restrict__ dKV) {
  const int h = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int inner = heads * 64;
  const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  for (int64_t r = (int64_t)blockIdx.y * 4 + wave; r < R; r += (int64_t)gridDim.y * 4) {
    float dk = 0.f, dv = 0.f;
    for (int i = 0; i < kq; ++i) {
      const int64_t o = ((int64_t)h * kq + i) * R + r;
      float p = P[o];
      if (drop_p > 0.f) p = drop_keep(seed, (uint64_t)(h * kq + i), (uint32_t)r, drop_p) ? p * keep_scale : 0.f;
      dk += dd[o] * Q[(int64_t)i * inner + h * 64 + lane];
      dv += p * dO[(int64_t)i * inner + h * 64 + lane];
    }
    dKV[r * 2 * inner + h * 64 + lane] = dk * scale;
    dKV[r * 2 * inner + inner + h * 64 + lane] = dv;
  }
}

// grid (heads, k): dQ[i,h,:] = scale sum_r dd[h,i,r] k_r
__global__ __launch_bounds__(MCA_THREADS) void mca_dq_kernel(const float* __restrict__ KV, const float* __restrict__ dd,
                                                             int64_t R, int heads, int kq, float scale,
                                                             float* __restrict__ dQ) {
  __shared__ float osum[4][64];
  const int h = blockIdx.x, i = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int inner = heads * 64;
  const float* drow = dd + ((int64_t)h * kq + i) * R;
  float acc = 0.f;
  for (int64_t r = wave; r < R; r += 4) acc += drow[r] * KV[r * 2 * inner + h * 64 + lane];
  osum[wave][lane] = acc;
  __syncthreads();
  if (wave == 0) dQ[(int64_t)i * inner + h * 64 + lane] = scale * (osum[0][lane] + osum[1][lane] + osum[2][lane] + osum[3][lane]);
}

__global__ void ema_kernel(const float* __restrict__ q, const float* __restrict__ z, float* __restrict__ out, int64_t n, float mm) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = q[i] * mm + z[i] * (1.f - mm);
}

// dz0 = dz * keep/(1-p) with the same (seed,row,col) stream as the forward epilogue
__global__ void drop_bwd_kernel(const float* __restrict__ dz, float* __restrict__ out, int64_t M, int E, float p, uint64_t seed) {
  const int64_t n = M * E;
  const float ks = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / E;
    const int e = (int)(i - m * E);
    out[i] = (p > 0.f && !drop_keep(seed, (uint64_t)m, (uint32_t)e, p)) ? 0.f : dz[i] * ks;
  }
}

struct MergeWs {
  float *xn, *mean, *rstd, *gq, *gmean, *grstd, *KV, *Q, *P, *O, *dd, *dKV, *dQ, *dO, *dxn, *dgq, *dz0, *lnp_w, *lnp_b, *scratch;
  int64_t scratch_bytes;
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
  w.scratch_bytes = 8 * 2 * I * E * 4;         // split-K slabs for dWkv (8 x [2I,E]) / colsum partials
  w.scratch = (float*)ar.take<char>(w.scratch_bytes);
  if (out) *out = w;
  return ar.off;
}

static int check_merge(const mhimx_merge* m) {
  MHIMX_CHECK_ARG(m && m->q_param && m->ln_w && m->ln_b && m->wkv && m->wq && m->wo && m->bo, "merge: null weights");
  MHIMX_CHECK_ARG(m->dim_head == 64, "merge: dim_head must be 64");
  MHIMX_CHECK_ARG(m->heads > 0 && m->k > 0 && m->E % 64 == 0, "merge: bad dims");
  return 0;
}

int merge_fwd(hipStream_t st, const mhimx_merge* m, const float* X, int64_t R, float* z, float* q_new, int update_q,
              void* ws, int64_t ws_bytes) {
  if (int r = check_merge(m)) return r;
  MHIMX_CHECK_ARG(X && z && R > 0, "merge_fwd: null args");
  const int64_t E = m->E, k = m->k, H = m->heads, I = H * m->dim_head;
  Arena ar(ws, ws_bytes);
  MergeWs w;
  merge_ws_layout(ar, R, E, k, H, m->dim_head, &w);
  MHIMX_CHECK_ARG(ar.ok(), "merge_fwd: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)ar.off);
  const int fprec = m->prec == MHIMX_PREC_F32 ? MHIMX_PREC_F32 : MHIMX_PREC_BF16X3;   // small GEMMs: ~fp32 accuracy
  if (int r = layernorm_fwd(st, X, R, E, m->ln_w, m->ln_b, w.xn, w.mean, w.rstd)) return r;
  if (int r = layernorm_fwd(st, m->q_param, k, E, m->ln_w, m->ln_b, w.gq, w.gmean, w.grstd)) return r;
  mhimx_gemm_nt_args g = {};
  g.A = w.xn; g.lda = E; g.B = m->wkv; g.ldb = E; g.C = w.KV; g.ldc = 2 * I; g.M = R; g.N = 2 * I; g.K = E; g.prec = fprec;
  if (int r = gemm_nt(st, g)) return r;
  g = {};
  g.A = w.gq; g.lda = E; g.B = m->wq; g.ldb = E; g.C = w.Q; g.ldc = I; g.M = k; g.N = I; g.K = E; g.prec = fprec;
  if (int r = gemm_nt(st, g)) return r;
  const float scale = 1.0f / sqrtf((float)m->dim_head);
  hipLaunchKernelGGL(mca_attend_fwd_kernel, dim3((unsigned)H, (unsigned)k), dim3(MCA_THREADS), 0, st, w.KV, w.Q, R, (int)H, (int)k,
                     scale, m->drop_p, m->drop_seed, w.P, w.O);
  MHIMX_LAUNCH_CHECK();
  g = {};
  g.A = w.O; g.lda = I; g.B = m->wo; g.ldb = I; g.C = z; g.ldc = E; g.M = k; g.N = E; g.K = I; g.bias = m->bo; g.prec = fprec;
  g.drop_p = m->drop_p; g.drop_seed = m->drop_seed + 0x9E3779B97F4A7C15ull;
  if (int r = gemm_nt(st, g)) return r;
  if (update_q) {
    MHIMX_CHECK_ARG(q_new, "merge_fwd: update_q needs q_new");
    hipLaunchKernelGGL(ema_kernel, dim3((unsigned)cdiv(k * E, 256)), dim3(256), 0, st, m->q_param, z, q_new, k * E, m->mm);
    MHIMX_LAUNCH_CHECK();
  }
  return 0;
}

int merge_bwd(hipStream_t st, const mhimx_merge* m, const float* X, int64_t R, const float* dz, float* dX,
              const mhimx_merge_grad* gr, void* ws, int64_t ws_bytes) {
  if (int r = check_merge(m)) return r;
  MHIMX_CHECK_ARG(X && dz && dX && gr && R > 0, "merge_bwd: null args");
  MHIMX_CHECK_ARG(m->wkv_t && m->wq_t && m->wo_t, "merge_bwd: transposed weights missing");
  MHIMX_CHECK_ARG(gr->d_ln_w && gr->d_ln_b && gr->d_wkv && gr->d_wq && gr->d_wo && gr->d_bo, "merge_bwd: null grads");
  const int64_t E = m->E, k = m->k, H = m->heads, I = H * m->dim_head;
  Arena ar(ws, ws_bytes);
  MergeWs w;
  merge_ws_layout(ar, R, E, k, H, m->dim_head, &w);
  MHIMX_CHECK_ARG(ar.ok(), "merge_bwd: workspace too small");
  const int gprec = m->prec == MHIMX_PREC_F32 ? MHIMX_PREC_F32 : MHIMX_PREC_BF16X3;
  const int acc = gr->accumulate;
  // through the output dropout and projection
  hipLaunchKernelGGL(drop_bwd_kernel, dim3((unsigned)cdiv(k * E, 256)), dim3(256), 0, st, dz, w.dz0, k, (int)E, m->drop_p,
                     m->drop_seed + 0x9E3779B97F4A7C15ull);
  MHIMX_LAUNCH_CHECK();
  if (int r = colsum(st, w.dz0, k, E, gr->d_bo, acc, w.scratch, w.scratch_bytes)) return r;
  mhimx_gemm_tn_args t = {};
  t.A = w.dz0; t.lda = E; t.B = w.O; t.ldb = I; t.C = gr->d_wo; t.ldc = I; t.M = k; t.K1 = E; t.K2 = I; t.splits = 1;
  t.accumulate = acc; t.prec = gprec;
  if (int r = gemm_tn(st, t)) return r;
  mhimx_gemm_nt_args g = {};
  g.A = w.dz0; g.lda = E; g.B = m->wo_t; g.ldb = E; g.C = w.dO; g.ldc = I; g.M = k; g.N = I; g.K = E; g.prec = gprec;
  if (int r = gemm_nt(st, g)) return r;
  // attention
  const float scale = 1.0f / sqrtf((float)m->dim_head);
  hipLaunchKernelGGL(mca_dp_kernel, dim3((unsigned)H, (unsigned)k), dim3(MCA_THREADS), 0, st, w.KV, w.dO, w.P, R, (int)H, (int)k,
                     m->drop_p, m->drop_seed, w.dd);
  MHIMX_LAUNCH_CHECK();
  const unsigned rb = (unsigned)(cdiv(R, 16) < 64 ? cdiv(R, 16) : 64);
  hipLaunchKernelGGL(mca_dkv_kernel, dim3((unsigned)H, rb), dim3(MCA_THREADS), 0, st, w.Q, w.dO, w.P, w.dd, R, (int)H, (int)k, scale,
                     m->drop_p, m->drop_seed, w.dKV);
  MHIMX_LAUNCH_CHECK();
  hipLaunchKernelGGL(mca_dq_kernel, dim3((unsigned)H, (unsigned)k), dim3(MCA_THREADS), 0, st, w.KV, w.dd, R, (int)H, (int)k, scale, w.dQ);
  MHIMX_LAUNCH_CHECK();
  // projections
  g = {};
  g.A = w.dKV; g.lda = 2 * I; g.B = m->wkv_t; g.ldb = 2 * I; g.C = w.dxn; g.ldc = E; g.M = R; g.N = E; g.K = 2 * I; g.prec = gprec;
  if (int r = gemm_nt(st, g)) return r;
  t = {};
  t.A = w.dKV; t.lda = 2 * I; t.B = w.xn; t.ldb = E; t.C = gr->d_wkv; t.ldc = E; t.M = R; t.K1 = 2 * I; t.K2 = E;
  t.splits = (R >= 2048 && gr->splits > 1) ? (gr->splits > 8 ? 8 : gr->splits) : 1; t.ws = w.scratch; t.accumulate = acc; t.prec = gprec;
  if (int r = gemm_tn(st, t)) return r;
  t = {};
  t.A = w.dQ; t.lda = I; t.B = w.gq; t.ldb = E; t.C = gr->d_wq; t.ldc = E; t.M = k; t.K1 = I; t.K2 = E; t.splits = 1;
  t.accumulate = acc; t.prec = gprec;
  if (int r = gemm_tn(st, t)) return r;
  g = {};
  g.A = w.dQ; g.lda = I; g.B = m->wq_t; g.ldb = I; g.C = w.dgq; g.ldc = E; g.M = k; g.N = E; g.K = I; g.prec = gprec;
  if (int r = gemm_nt(st, g)) return r;
  // LayerNorm: rows (dX + weight grads), then the global queries (weight grads only; the queries are not trained)
  if (int r = layernorm_bwd(st, w.dxn, X, R, E, m->ln_w, w.mean, w.rstd, dX, w.lnp_w, w.lnp_b, gr->d_ln_w, gr->d_ln_b, acc)) return r;
  if (int r = layernorm_bwd(st, w.dgq, m->q_param, k, E, m->ln_w, w.gmean, w.grstd, nullptr, w.lnp_w, w.lnp_b, gr->d_ln_w, gr->d_ln_b, 1)) return r;
  return 0;
}

}  // namespace mhimx

using namespace mhimx;

extern "C" int64_t mhimx_merge_ws_bytes(int64_t R, int64_t E, int64_t k, int64_t heads, int64_t dim_head) {
  Arena ar(nullptr, 0);
  return merge_ws_layout(ar, R, E, k, heads, dim_head, nullptr);
}
extern "C" int mhimx_merge_fwd(void* stream, const mhimx_merge* m, const float* X, int64_t R, float* z, float* q_new,
                               int32_t update_q, void* ws, int64_t ws_bytes) {
  return merge_fwd((hipStream_t)stream, m, X, R, z, q_new, update_q, ws, ws_bytes);
}
extern "C" int mhimx_merge_bwd(void* stream, const mhimx_merge* m, const float* X, int64_t R, const float* dz, float* dX,
                               const mhimx_merge_grad* g, void* ws, int64_t ws_bytes) {
  return merge_bwd((hipStream_t)stream, m, X, R, dz, dX, g, ws, ws_bytes);
}
