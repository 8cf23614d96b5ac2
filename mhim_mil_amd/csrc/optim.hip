// optim.hip — the bag-level tail of a train step: predictor + losses (+ their gradients) in one launch,
// and the fused Adam + EMA-teacher update over flat parameter buffers.
#include <math.h>
#include <stdlib.h>

#include "common.hpp"

namespace mhimx {

int reduce_flush(hipStream_t st, mhimx_reduce_list* list);      // rows.hip

constexpr int HEAD_THREADS = 256;

MHIMX_DEV float blk_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
MHIMX_DEV float blk_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// One block.  C <= 16 classes.
__global__ __launch_bounds__(HEAD_THREADS) void head_kernel(const float* __restrict__ z, const float* __restrict__ t,
                                                            const float* __restrict__ wp, const float* __restrict__ bp,
                                                            const int64_t* __restrict__ label, int E, int C, float temp_t,
                                                            float main_alpha, float aux_alpha, float inv_accum,
                                                            float* __restrict__ logits, float* __restrict__ losses,
                                                            float* __restrict__ g_z, float* __restrict__ d_wp,
                                                            float* __restrict__ d_bp, int accumulate,
                                                            const float* __restrict__ g_logits_in,
                                                            const float* __restrict__ g_cl_in) {
  __shared__ float red[4];
  __shared__ float lg[16], gl[16];
  const int tid = threadIdx.x;
  // logits
  for (int c = 0; c < C; ++c) {
    float p = 0.f;
    for (int e = tid; e < E; e += HEAD_THREADS) p += wp[c * E + e] * z[e];
    p = blk_sum(p, red);
    if (tid == 0) lg[c] = p + (bp ? bp[c] : 0.f);
  }
  __syncthreads();
  // cross entropy on the logits (criterion = nn.CrossEntropyLoss, base_engine.py:99)
  float ce = 0.f;
  if (label) {
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, lg[c]);
    float den = 0.f;
    for (int c = 0; c < C; ++c) den += expf(lg[c] - mx);
    const int y = (int)label[0];
    ce = -(lg[y] - mx - logf(den));
    if (tid < C) gl[tid] = main_alpha * inv_accum * (expf(lg[tid] - mx) / den - (tid == y ? 1.f : 0.f));
  } else if (tid < C) {
    gl[tid] = g_logits_in ? g_logits_in[tid] : 0.f;      // upstream dLoss/dlogits supplied by the caller (autograd path)
  }
  if (g_cl_in) aux_alpha = g_cl_in[0];                  // upstream dLoss/dcl
  __syncthreads();
  // soft-target CE over the E feature dims: cl = -sum softmax(t/temp_t) * log_softmax(z)   (losses.py:40-43)
  float cl = 0.f;
  float zmx = 0.f, zden = 1.f, tmx = 0.f, tden = 1.f;
  const bool aux = (t != nullptr);
  if (aux) {
    float a = -INFINITY, b = -INFINITY;
    for (int e = tid; e < E; e += HEAD_THREADS) { a = fmaxf(a, z[e]); b = fmaxf(b, t[e] / temp_t); }
    zmx = blk_max(a, red);
    tmx = blk_max(b, red);
    float sa = 0.f, sb = 0.f;
    for (int e = tid; e < E; e += HEAD_THREADS) { sa += expf(z[e] - zmx); sb += expf(t[e] / temp_t - tmx); }
    zden = blk_sum(sa, red);
    tden = blk_sum(sb, red);
    const float lz = logf(zden);
    float acc = 0.f;
    for (int e = tid; e < E; e += HEAD_THREADS) acc += (expf(t[e] / temp_t - tmx) / tden) * (z[e] - zmx - lz);
    cl = -blk_sum(acc, red);
  }
  if (tid == 0) {
    for (int c = 0; c < C; ++c) logits[c] = lg[c];
    losses[0] = main_alpha * ce + aux_alpha * cl;
    losses[1] = ce;
    losses[2] = cl;
  }
  // gradients
  for (int e = tid; e < E; e += HEAD_THREADS) {
    float g = 0.f;
    for (int c = 0; c < C; ++c) g += wp[c * E + e] * gl[c];
    if (aux) g += aux_alpha * inv_accum * (expf(z[e] - zmx) / zden - expf(t[e] / temp_t - tmx) / tden);
    g_z[e] = g;
    if (d_wp)
      for (int c = 0; c < C; ++c) {
        const float v = gl[c] * z[e];
        d_wp[c * E + e] = accumulate ? d_wp[c * E + e] + v : v;
      }
  }
  if (d_bp && tid < C) d_bp[tid] = accumulate ? d_bp[tid] + gl[tid] : gl[tid];
}

// The same head for the production shapes (E <= 4 x 256, C <= 4) as a SHORT dependency chain: the kernel runs on one workgroup in the middle
// of the step's serial chain, and the generic form above is three dependent memory round trips (z / wp, then label / bp, then t) and
// 2 + C + 3 block reductions of two barriers each.  Here every input is requested at entry (registers), and the block reductions are
// batched: {C logit sums, max z, max t} -> {sum e^z, sum e^t} -> {soft-target CE}: three barrier pairs.  Same sums in the same order
// (wave_sum, then the four wave values left to right): the bits of head_kernel.
template <int Q>
__global__ __launch_bounds__(HEAD_THREADS) void head_fast_kernel(const float* __restrict__ z, const float* __restrict__ t,
                                                                 const float* __restrict__ wp, const float* __restrict__ bp,
                                                                 const int64_t* __restrict__ label, int E, int C, float temp_t,
                                                                 float main_alpha, float aux_alpha, float inv_accum,
                                                                 float* __restrict__ logits, float* __restrict__ losses,
                                                                 float* __restrict__ g_z, float* __restrict__ d_wp,
                                                                 float* __restrict__ d_bp, int accumulate,
                                                                 const float* __restrict__ g_logits_in,
                                                                 const float* __restrict__ g_cl_in, BagBatch bb) {
  __shared__ float red[8][4];
  if (blockIdx.z) {      // (common.hpp: one workgroup per bag of an accumulation window; the label list is one of the moving ranges)
    MHIMX_BAG(z); MHIMX_BAG(t); MHIMX_BAG(label); MHIMX_BAG(logits); MHIMX_BAG(losses); MHIMX_BAG(g_z); MHIMX_BAG(d_wp); MHIMX_BAG(d_bp);
    MHIMX_BAG(g_logits_in); MHIMX_BAG(g_cl_in);
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool aux = (t != nullptr);
  // ---- every load of the kernel, in flight together
  float zv[Q], tv[Q], wv[4][Q], dw_old[4][Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int e = tid + q * HEAD_THREADS;
    const bool in = e < E;
    zv[q] = in ? z[e] : 0.f;
    tv[q] = (in && aux) ? t[e] : 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      wv[c][q] = (in && c < C) ? wp[c * E + e] : 0.f;
      dw_old[c][q] = (in && c < C && d_wp && accumulate) ? d_wp[c * E + e] : 0.f;
    }
  }
  const int y = label ? (int)label[0] : 0;
  float bpv[4], glin[4], dbp_old[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    bpv[c] = (bp && c < C) ? bp[c] : 0.f;
    glin[c] = (!label && g_logits_in && c < C) ? g_logits_in[c] : 0.f;
    dbp_old[c] = (d_bp && accumulate && c < C) ? d_bp[c] : 0.f;
  }
  if (g_cl_in) aux_alpha = g_cl_in[0];
  // ---- reduction 1: the C logit sums, max z, max t / temp_t
  float p[4] = {0.f, 0.f, 0.f, 0.f}, a = -INFINITY, b = -INFINITY;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const bool in = tid + q * HEAD_THREADS < E;
#pragma unroll
    for (int c = 0; c < 4; ++c) p[c] += wv[c][q] * zv[q];
    if (in) { a = fmaxf(a, zv[q]); b = fmaxf(b, tv[q] / temp_t); }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) p[c] = wave_sum(p[c]);
  a = wave_max(a);
  b = wave_max(b);
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) red[c][wave] = p[c];
    red[4][wave] = a;
    red[5][wave] = b;
  }
  __syncthreads();
  float lg[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) lg[c] = (red[c][0] + red[c][1] + red[c][2] + red[c][3]) + bpv[c];
  const float zmx = aux ? fmaxf(fmaxf(red[4][0], red[4][1]), fmaxf(red[4][2], red[4][3])) : 0.f;
  const float tmx = aux ? fmaxf(fmaxf(red[5][0], red[5][1]), fmaxf(red[5][2], red[5][3])) : 0.f;
  // cross entropy on the logits (every thread: C <= 4 values)
  float ce = 0.f, gl[4] = {0.f, 0.f, 0.f, 0.f};
  if (label) {
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < 4; ++c) if (c < C) mx = fmaxf(mx, lg[c]);
    float den = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) if (c < C) den += expf(lg[c] - mx);
    float ly = lg[0];
#pragma unroll
    for (int c = 1; c < 4; ++c) if (c == y) ly = lg[c];
    ce = -(ly - mx - logf(den));
#pragma unroll
    for (int c = 0; c < 4; ++c) if (c < C) gl[c] = main_alpha * inv_accum * (expf(lg[c] - mx) / den - (c == y ? 1.f : 0.f));
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) gl[c] = glin[c];
  }
  // ---- reduction 2: the two softmax denominators
  float cl = 0.f, zden = 1.f, tden = 1.f;
  float ez[Q], et[Q];
  if (aux) {
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const bool in = tid + q * HEAD_THREADS < E;
      ez[q] = in ? expf(zv[q] - zmx) : 0.f;
      et[q] = in ? expf(tv[q] / temp_t - tmx) : 0.f;
      sa += ez[q];
      sb += et[q];
    }
    sa = wave_sum(sa);
    sb = wave_sum(sb);
    if (lane == 0) { red[6][wave] = sa; red[7][wave] = sb; }
    __syncthreads();
    zden = red[6][0] + red[6][1] + red[6][2] + red[6][3];
    tden = red[7][0] + red[7][1] + red[7][2] + red[7][3];
    // ---- reduction 3: cl = -sum softmax(t / temp_t) log_softmax(z)
    const float lz = logf(zden);
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q)
      if (tid + q * HEAD_THREADS < E) acc += (et[q] / tden) * (zv[q] - zmx - lz);
    acc = wave_sum(acc);
    if (lane == 0) red[0][wave] = acc;                          // (red[0]'s logit sums were read by everybody before the barrier above)
    __syncthreads();
    cl = -(red[0][0] + red[0][1] + red[0][2] + red[0][3]);
  }
  if (tid == 0) {
    for (int c = 0; c < C; ++c) logits[c] = lg[c];
    losses[0] = main_alpha * ce + aux_alpha * cl;
    losses[1] = ce;
    losses[2] = cl;
  }
  // ---- gradients
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int e = tid + q * HEAD_THREADS;
    if (e >= E) continue;
    float g = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) if (c < C) g += wv[c][q] * gl[c];
    if (aux) g += aux_alpha * inv_accum * (ez[q] / zden - et[q] / tden);
    g_z[e] = g;
    if (d_wp) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < C) {
          const float v = gl[c] * zv[q];
          d_wp[c * E + e] = accumulate ? dw_old[c][q] + v : v;
        }
    }
  }
  if (d_bp && tid < C) {
    float gv = gl[0], ov = dbp_old[0];
#pragma unroll
    for (int c = 1; c < 4; ++c) if (c == tid) { gv = gl[c]; ov = dbp_old[c]; }
    d_bp[tid] = accumulate ? ov + gv : gv;
  }
}

// DSMIL head (common_mil.py:26-28, mhim.py:355-364, losses.py:26-45): logits = 0.5 (bag + max-instance), CE on them, and the
// distillation loss cl = mean_c [ -sum_v softmax(Bt[c]/temp_t)_v log_softmax(Bs[c])_v ] on the per-class bag features.
// Label given: loss and all gradients in one launch.  label == NULL: the autograd form (cl only, upstream scale g_cl_in).
__global__ __launch_bounds__(HEAD_THREADS) void dsmil_head_kernel(const float* __restrict__ lb, const float* __restrict__ li,
                                                                  const int64_t* __restrict__ label, const float* __restrict__ Bs,
                                                                  const float* __restrict__ Bt, int C, int V, float temp_t,
                                                                  float main_alpha, float aux_alpha, float inv_accum,
                                                                  float* __restrict__ losses, float* __restrict__ g_lb,
                                                                  float* __restrict__ g_li, float* __restrict__ g_B,
                                                                  const float* __restrict__ g_cl_in) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  float ce = 0.f;
  if (label) {
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, 0.5f * (lb[c] + li[c]));
    float den = 0.f;
    for (int c = 0; c < C; ++c) den += expf(0.5f * (lb[c] + li[c]) - mx);
    const int y = (int)label[0];
    ce = -(0.5f * (lb[y] + li[y]) - mx - logf(den));
    if (tid < C) {
      const float g = 0.5f * main_alpha * inv_accum * (expf(0.5f * (lb[tid] + li[tid]) - mx) / den - (tid == y ? 1.f : 0.f));
      g_lb[tid] = g;
      g_li[tid] = g;
    }
  }
  if (g_cl_in) aux_alpha = g_cl_in[0];
  float cl = 0.f;
  if (Bt) {
    for (int c = 0; c < C; ++c) {
      const float* s = Bs + (int64_t)c * V;
      const float* t = Bt + (int64_t)c * V;
      float a = -INFINITY, b = -INFINITY;
      for (int e = tid; e < V; e += HEAD_THREADS) { a = fmaxf(a, s[e]); b = fmaxf(b, t[e] / temp_t); }
      const float smx = blk_max(a, red), tmx = blk_max(b, red);
      float sa = 0.f, sb = 0.f;
      for (int e = tid; e < V; e += HEAD_THREADS) { sa += expf(s[e] - smx); sb += expf(t[e] / temp_t - tmx); }
      const float sden = blk_sum(sa, red), tden = blk_sum(sb, red);
      const float ls = logf(sden);
      float acc = 0.f;
      for (int e = tid; e < V; e += HEAD_THREADS) {
        const float pt = expf(t[e] / temp_t - tmx) / tden;
        acc += pt * (s[e] - smx - ls);
        if (g_B) g_B[(int64_t)c * V + e] = aux_alpha * inv_accum / (float)C * (expf(s[e] - smx) / sden - pt);
      }
      cl -= blk_sum(acc, red) / (float)C;
    }
  } else if (g_B) {
    for (int i = tid; i < C * V; i += HEAD_THREADS) g_B[i] = 0.f;
  }
  if (tid == 0) {
    losses[0] = main_alpha * ce + aux_alpha * cl;
    losses[1] = ce;
    losses[2] = cl;
  }
}

// optional parts of an optimiser step (mhimx_optim_step): learning-rate table, gradient slabs to add, global-norm clipping
// (fold: split-K slab sums the update kernel performs itself - gradient elements [off, off + n) = sum_z parts[z * n + i])
constexpr int FOLD_MAX = 4;
struct FoldJobs { const float* parts[FOLD_MAX]; int64_t off[FOLD_MAX], n[FOLD_MAX]; int G[FOLD_MAX], accumulate[FOLD_MAX]; int count; };
struct OptimExtra { const float* lr_table; int64_t lr_len; const float* g_extra; int64_t n_extra, extra_pitch; float clip_norm;
                    const float* sq_parts; int n_parts; FoldJobs fold; int64_t extra_lo; int extra_only; };

// the slab sum of reduce_jobs.hpp (kind 1) for four neighbouring elements: four running sums, eight slabs in flight, then four, then one -
// the same additions in the same order, so a folded gradient has the bits of a reduced one
MHIMX_DEV float4 fold_sum4(const float* __restrict__ parts, int64_t n, int64_t idx, int G) {
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  auto ld = [&](int z) { return *reinterpret_cast<const float4*>(parts + (int64_t)z * n + idx); };
  auto add = [](float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
  int z = 0;
  for (; z + 8 <= G; z += 8) {
    const float4 p0 = ld(z), p1 = ld(z + 1), p2 = ld(z + 2), p3 = ld(z + 3), p4 = ld(z + 4), p5 = ld(z + 5), p6 = ld(z + 6), p7 = ld(z + 7);
    add(s0, p0); add(s1, p1); add(s2, p2); add(s3, p3);
    add(s0, p4); add(s1, p5); add(s2, p6); add(s3, p7);
  }
  for (; z + 4 <= G; z += 4) { add(s0, ld(z)); add(s1, ld(z + 1)); add(s2, ld(z + 2)); add(s3, ld(z + 3)); }
  for (; z < G; ++z) add(s0, ld(z));
  return make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w));
}

// per-block partial sums of squares of the (scaled, slab-summed) gradient: the first stage of clip_grad_norm_ (base_engine.py:115-119)
__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float* __restrict__ g, int64_t n, float gscale, OptimExtra ex, float* __restrict__ parts) {
  __shared__ float red[256];
  float a = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const bool slabs = i >= ex.extra_lo;
    float v = (slabs && ex.extra_only) ? 0.f : g[i];
    if (slabs)
      for (int64_t z = 0; z < ex.n_extra; ++z) v += ex.g_extra[z * ex.extra_pitch + i];
    v *= gscale;
    a += v * v;
  }
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) parts[blockIdx.x] = red[0];
}

// torch.optim.Adam semantics (weight decay folded into the gradient, bias-corrected, eps outside the sqrt of
// the corrected second moment) + EMA teacher.  bc1 = 1-beta1^t, bc2s = sqrt(1-beta2^t) come from the host in fp64.
MHIMX_DEV void adam_one(float& w, float& gi, float& mi, float& vi, float lr_over_bc1, float bc2s, float beta1, float beta2,
                        float eps, float wd, float gscale) {
  gi = gi * gscale + wd * w;
  mi = beta1 * mi + (1.f - beta1) * gi;
  vi = beta2 * vi + (1.f - beta2) * gi * gi;
  w = w - lr_over_bc1 * (mi / (sqrtf(vi) / bc2s + eps));
}

__global__ __launch_bounds__(256) void adam_ema_kernel(
    float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, float* __restrict__ teacher,
    int64_t n_train, int64_t n_all, float lr_over_bc1, float bc2s, float beta1, float beta2, float eps, float wd, float gscale,
    float mm, int zero_grad, const uint64_t* __restrict__ step_dev, float lr, const float* __restrict__ mm_table, int64_t mm_len,
    int64_t step_host, double ln_beta1, double ln_beta2, OptimExtra ex) {
  __shared__ float sc[4];
  __shared__ float clip_red[256];
  if (ex.clip_norm > 0.f) {             // clip_grad_norm_: every block sums the per-block partial sums of squares in the same fixed order
    float a = 0.f;
    for (int i = threadIdx.x; i < ex.n_parts; i += 256) a += ex.sq_parts[i];
    clip_red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) clip_red[threadIdx.x] += clip_red[threadIdx.x + o];
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) {               // the fp64 pow()s of the bias corrections: once per block, not once per thread
    int64_t step = step_host;
    if (step_dev) step = (int64_t)step_dev[0];      // graph replay: the step count lives on the device
    if (ex.lr_table) {                  // per-update learning-rate schedule: entry of this update, last one held
      int64_t i = step - 1;
      i = i < 0 ? 0 : (i >= ex.lr_len ? ex.lr_len - 1 : i);
      lr = ex.lr_table[i];
    }
    if (step_dev || ex.lr_table) {
      const double t = (double)step;
      lr_over_bc1 = (float)((double)lr / (1.0 - exp(t * ln_beta1)));       // beta^t = e^(t ln beta), ln beta from the host in fp64
      bc2s = (float)sqrt(1.0 - exp(t * ln_beta2));
    }
    float coef = 1.f;
    if (ex.clip_norm > 0.f) {           // torch.nn.utils.clip_grad_norm_: g *= min(1, max_norm / (total_norm + 1e-6))
      coef = ex.clip_norm / (sqrtf(clip_red[0]) + 1e-6f);
      coef = coef > 1.f ? 1.f : coef;
    }
    sc[3] = coef;
    if (mm_table) {                     // EMA momentum schedule (base_engine.py:160-161): entry of this iteration, last one held
      int64_t i = step - 1;
      i = i < 0 ? 0 : (i >= mm_len ? mm_len - 1 : i);
      mm = mm_table[i];
    }
    sc[0] = lr_over_bc1; sc[1] = bc2s; sc[2] = mm;
  }
  __syncthreads();
  lr_over_bc1 = sc[0];
  bc2s = sc[1];
  mm = sc[2];
  gscale *= sc[3];
  const int64_t n4 = n_all / 4;
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                        reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(teacher)) & 15) == 0;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q * 4 < n_all; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i0 = q * 4;
    if (vec_ok && q < n4 && (i0 + 4 <= n_train || i0 >= n_train)) {
      float4 w = reinterpret_cast<float4*>(p)[q];
      if (i0 < n_train) {
        const bool slabs = i0 >= ex.extra_lo;                 // (extra_lo % 4 == 0: the four elements are on one side)
        float4 gi = (slabs && ex.extra_only) ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<float4*>(g)[q];
        float4 mi = reinterpret_cast<float4*>(m)[q], vi = reinterpret_cast<float4*>(v)[q];
        for (int f = 0; f < ex.fold.count; ++f)           // a folded split-K slab sum: this gradient element is summed here, not by a reduction pass
          if (i0 >= ex.fold.off[f] && i0 < ex.fold.off[f] + ex.fold.n[f]) {
            const float4 sv = fold_sum4(ex.fold.parts[f], ex.fold.n[f], i0 - ex.fold.off[f], ex.fold.G[f]);
            if (ex.fold.accumulate[f]) { gi.x += sv.x; gi.y += sv.y; gi.z += sv.z; gi.w += sv.w; }
            else gi = sv;
          }
        for (int64_t z = 0; slabs && z < ex.n_extra; ++z) {       // gradient slabs of an accumulation window's other streams / bags (fixed order)
          const float4 e = *reinterpret_cast<const float4*>(ex.g_extra + z * ex.extra_pitch + i0);
          gi.x += e.x; gi.y += e.y; gi.z += e.z; gi.w += e.w;
        }
        adam_one(w.x, gi.x, mi.x, vi.x, lr_over_bc1, bc2s, beta1, beta2, eps, wd, gscale);
        adam_one(w.y, gi.y, mi.y, vi.y, lr_over_bc1, bc2s, beta1, beta2, eps, wd, gscale);
        adam_one(w.z, gi.z, mi.z, vi.z, lr_over_bc1, bc2s, beta1, beta2, eps, wd, gscale);
        adam_one(w.w, gi.w, mi.w, vi.w, lr_over_bc1, bc2s, beta1, beta2, eps, wd, gscale);
        reinterpret_cast<float4*>(m)[q] = mi;
        reinterpret_cast<float4*>(v)[q] = vi;
        reinterpret_cast<float4*>(p)[q] = w;
        if (zero_grad) reinterpret_cast<float4*>(g)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (teacher) {
        float4 t = reinterpret_cast<float4*>(teacher)[q];
        t.x = t.x * mm + w.x * (1.f - mm); t.y = t.y * mm + w.y * (1.f - mm);
        t.z = t.z * mm + w.z * (1.f - mm); t.w = t.w * mm + w.w * (1.f - mm);
        reinterpret_cast<float4*>(teacher)[q] = t;
      }
      continue;
    }
    for (int64_t i = i0; i < i0 + 4 && i < n_all; ++i) {
      float w = p[i];
      if (i < n_train) {
        const bool slabs = i >= ex.extra_lo;
        float gi = (slabs && ex.extra_only) ? 0.f : g[i], mi = m[i], vi = v[i];
        for (int64_t z = 0; slabs && z < ex.n_extra; ++z) gi += ex.g_extra[z * ex.extra_pitch + i];
        adam_one(w, gi, mi, vi, lr_over_bc1, bc2s, beta1, beta2, eps, wd, gscale);
        m[i] = mi;
        v[i] = vi;
        p[i] = w;
        if (zero_grad) g[i] = 0.f;
      }
      if (teacher) teacher[i] = teacher[i] * mm + w * (1.f - mm);
    }
  }
}

}  // namespace mhimx

using namespace mhimx;

extern "C" int mhimx_head_fwd_bwd(void* stream, const float* z, const float* t, const float* wp, const float* bp,
                                  const int64_t* label_dev, int64_t E, int64_t C, float temp_t, float main_alpha,
                                  float aux_alpha, float inv_accum, float* logits, float* losses, float* g_z, float* d_wp,
                                  float* d_bp, int32_t accumulate, const float* g_logits_in, const float* g_cl_in) {
  MHIMX_CHECK_ARG(z && wp && logits && losses && g_z, "head: null args");
  MHIMX_CHECK_ARG(C > 0 && C <= 16 && E > 0, "head: bad dims");
  static const bool slow_head = getenv("MHIMX_HEAD_GENERIC") != nullptr;
  if (!slow_head && C <= 4 && E <= 2 * HEAD_THREADS)
    hipLaunchKernelGGL(head_fast_kernel<2>, bgrid(1), dim3(HEAD_THREADS), 0, (hipStream_t)stream, z, t, wp, bp, label_dev, (int)E, (int)C,
                     temp_t, main_alpha, aux_alpha, inv_accum, logits, losses, g_z, d_wp, d_bp, accumulate, g_logits_in, g_cl_in, cur_batch());
  else if (!slow_head && C <= 4 && E <= 4 * HEAD_THREADS)
    hipLaunchKernelGGL(head_fast_kernel<4>, bgrid(1), dim3(HEAD_THREADS), 0, (hipStream_t)stream, z, t, wp, bp, label_dev, (int)E, (int)C,
                     temp_t, main_alpha, aux_alpha, inv_accum, logits, losses, g_z, d_wp, d_bp, accumulate, g_logits_in, g_cl_in, cur_batch());
  else if (cur_batch().n > 0)
    return fail(-1, "head: a bag-batched launch takes the short-chain head (C <= 4, E <= 1024)");
  else
  hipLaunchKernelGGL(head_kernel, dim3(1), dim3(HEAD_THREADS), 0, (hipStream_t)stream, z, t, wp, bp, label_dev, (int)E, (int)C,
                     temp_t, main_alpha, aux_alpha, inv_accum, logits, losses, g_z, d_wp, d_bp, accumulate, g_logits_in, g_cl_in);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

extern "C" int mhimx_dsmil_head(void* stream, const float* logits_bag, const float* logits_ins, const int64_t* label_dev,
                                const float* Bs, const float* Bt, int64_t C, int64_t V, float temp_t, float main_alpha,
                                float aux_alpha, float inv_accum, float* losses, float* g_logits_bag, float* g_logits_ins,
                                float* g_B, const float* g_cl_in) {
  MHIMX_CHECK_ARG(losses && Bs && C > 0 && C <= 16 && V > 0, "dsmil_head: bad args");
  MHIMX_CHECK_ARG(!label_dev || (logits_bag && logits_ins && g_logits_bag && g_logits_ins), "dsmil_head: label needs the logits and their gradients");
  hipLaunchKernelGGL(dsmil_head_kernel, dim3(1), dim3(HEAD_THREADS), 0, (hipStream_t)stream, logits_bag, logits_ins, label_dev, Bs, Bt,
                     (int)C, (int)V, temp_t, main_alpha, aux_alpha, inv_accum, losses, g_logits_bag, g_logits_ins, g_B, g_cl_in);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

extern "C" int mhimx_optim_step(void* stream, const mhimx_optim_args* a) {
  MHIMX_CHECK_ARG(a, "optim_step: null args");
  MHIMX_CHECK_ARG(!a->mm_table || a->mm_len > 0, "optim_step: empty momentum schedule");
  MHIMX_CHECK_ARG(!a->lr_table || a->lr_len > 0, "optim_step: empty learning-rate schedule");
  MHIMX_CHECK_ARG(a->p && a->g && a->m && a->v && a->n_train >= 0 && a->n_all >= a->n_train && (a->step >= 1 || a->step_dev), "optim_step: bad args");
  MHIMX_CHECK_ARG(a->n_extra >= 0 && (a->n_extra == 0 || (a->g_extra && a->extra_pitch >= a->n_train && a->extra_pitch % 4 == 0 && aligned16(a->g_extra))),
                  "optim_step: gradient slabs need a 16-byte aligned base and a pitch >= n_train that is a multiple of 4");
  MHIMX_CHECK_ARG(!(a->clip_norm > 0.f) || (a->ws && a->ws_floats >= 1024), "optim_step: clipping needs a workspace of 1024 floats");
  MHIMX_CHECK_ARG(a->extra_lo >= 0 && a->extra_lo % 4 == 0 && (a->n_extra > 0 || (a->extra_lo == 0 && !a->extra_only)),
                  "optim_step: extra_lo (a multiple of 4) / extra_only describe the gradient slabs of g_extra");
  int64_t step = a->step < 1 ? 1 : a->step;
  if (a->n_all == 0) return 0;
  OptimExtra ex{a->lr_table, a->lr_len, a->g_extra, a->n_extra, a->extra_pitch, a->clip_norm > 0.f ? a->clip_norm : 0.f, a->ws, 0, {}, a->extra_lo, a->extra_only};
  if (a->fold) {
    mhimx_reduce_list* l = a->fold;
    MHIMX_CHECK_ARG(l->n >= 0 && l->n <= MHIMX_REDUCE_MAX, "optim_step: bad reduction list");
    // the vector path of the update kernel is the one that folds: every buffer 16-byte aligned
    const bool vec_ok = aligned16(a->p) && aligned16(a->g) && aligned16(a->m) && aligned16(a->v) && (!a->teacher || aligned16(a->teacher));
    if (!(a->clip_norm > 0.f) && vec_ok) {
      int kept = 0;
      for (int i = 0; i < l->n; ++i) {
        const mhimx_reduce_job& j = l->j[i];
        const int64_t n = j.K1 * j.K2, off = j.out - a->g;
        const bool take = j.kind == 1 && ex.fold.count < FOLD_MAX && j.G >= 1 && j.G <= 64 && j.ldo == j.K2 && n > 0 && n % 4 == 0 && aligned16(j.parts) &&
                          j.out >= a->g && off % 4 == 0 && off + n <= a->n_train / 4 * 4;
        if (take) {
          const int f = ex.fold.count++;
          ex.fold.parts[f] = j.parts; ex.fold.off[f] = off; ex.fold.n[f] = n; ex.fold.G[f] = (int)j.G; ex.fold.accumulate[f] = j.accumulate;
        } else {
          l->j[kept++] = j;
        }
      }
      l->n = kept;
    }
    if (l->n > 0 || l->side.pending || l->parked.pending)
      if (int r = reduce_flush((hipStream_t)stream, l)) return r;
  }
  if (ex.clip_norm > 0.f && a->n_train > 0) {
    const int64_t nb = cdiv(a->n_train, 1024) < 1024 ? cdiv(a->n_train, 1024) : 1024;
    ex.n_parts = (int)nb;
    hipLaunchKernelGGL(grad_sumsq_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, a->g, a->n_train, a->grad_scale, ex, a->ws);
    MHIMX_LAUNCH_CHECK();
  } else {
    ex.clip_norm = 0.f;
  }
  const double bc1 = 1.0 - pow((double)a->beta1, (double)step);
  const double bc2 = 1.0 - pow((double)a->beta2, (double)step);
  const int64_t blocks = cdiv(a->n_all, 1024) < 2048 ? cdiv(a->n_all, 1024) : 2048;
  hipLaunchKernelGGL(adam_ema_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a->p, a->g, a->m, a->v,
                     a->teacher, a->n_train, a->n_all, (float)((double)a->lr / bc1), (float)sqrt(bc2), a->beta1, a->beta2, a->eps, a->weight_decay,
                     a->grad_scale, a->ema_mm, a->zero_grad, a->step_dev, a->lr, a->mm_table, a->mm_len, step, log((double)a->beta1),
                     log((double)a->beta2), ex);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

extern "C" int mhimx_adam_ema(void* stream, float* p, const float* g, float* m, float* v, float* teacher, int64_t n_train,
                              int64_t n_all, int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                              float grad_scale, float ema_mm, int32_t zero_grad, const uint64_t* step_dev,
                              const float* mm_table, int64_t mm_len) {
  mhimx_optim_args a{};
  a.p = p; a.g = const_cast<float*>(g); a.m = m; a.v = v; a.teacher = teacher; a.n_train = n_train; a.n_all = n_all; a.step = step;
  a.step_dev = step_dev; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay; a.grad_scale = grad_scale;
  a.ema_mm = ema_mm; a.mm_table = mm_table; a.mm_len = mm_len; a.zero_grad = zero_grad;
  return mhimx_optim_step(stream, &a);
}

// float4 stream copy: the on-box HBM stream rate bench.py prints beside the 8 TB/s nominal peak.  One 16-byte element
// per thread, non-temporal load and store, as many 256-thread workgroups as elements / 256 (tools/micro/copy_bw.hip, round 4: 6.7-6.8 TB/s
// for 1 GiB -> 1 GiB at 131 072+ workgroups against 5.0-5.6 TB/s for the round-3 form - four loads in flight per lane, 65 536 workgroups,
// plain accesses - and 5.1 TB/s for hipMemcpyAsync; the guide quotes 6.29)
typedef float sc_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_copy_kernel(const sc_f4* __restrict__ src, sc_f4* __restrict__ dst, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
extern "C" int mhimx_stream_copy(void* stream, const float* src, float* dst, int64_t n_floats) {
  MHIMX_CHECK_ARG(src && dst && n_floats >= 0 && n_floats % 4 == 0 && aligned16(src) && aligned16(dst), "stream_copy: 16-byte aligned buffers, n % 4 == 0");
  if (n_floats == 0) return 0;
  const int64_t n4 = n_floats / 4;
  const int64_t blocks = cdiv(n4, 256) < (1 << 20) ? cdiv(n4, 256) : (1 << 20);
  hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const sc_f4*>(src),
                     reinterpret_cast<sc_f4*>(dst), n4);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

__global__ void tick_kernel(uint64_t* c) { c[0] += 1; }
extern "C" int mhimx_tick(void* stream, uint64_t* counter) {
  MHIMX_CHECK_ARG(counter, "tick: null counter");
  hipLaunchKernelGGL(tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
