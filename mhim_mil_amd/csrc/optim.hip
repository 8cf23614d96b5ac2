// optim.hip — the bag-level tail of a train step: predictor + losses (+ their gradients) in one launch,
// and the fused Adam + EMA-teacher update over flat parameter buffers.
#include <math.h>

#include "common.hpp"

namespace mhimx {

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

// torch.optim.Adam semantics (weight decay folded into the gradient, bias-corrected, eps outside the sqrt of
// the corrected second moment) + EMA teacher.  bc1 = 1-beta1^t, bc2s = sqrt(1-beta2^t) come from the host in fp64.
__global__ void adam_ema_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                float* __restrict__ teacher, int64_t n_train, int64_t n_all, float lr_over_bc1, float bc2s,
                                float beta1, float beta2, float eps, float wd, float gscale, float mm, int zero_grad,
                                const uint64_t* __restrict__ step_dev, float lr, const float* __restrict__ mm_table,
                                int64_t mm_len, int64_t step_host) {
  int64_t step = step_host;
  if (step_dev) {                       // graph replay: the step count lives on the device
    step = (int64_t)step_dev[0];
    const double t = (double)step;
    lr_over_bc1 = (float)((double)lr / (1.0 - pow((double)beta1, t)));
    bc2s = (float)sqrt(1.0 - pow((double)beta2, t));
  }
  if (mm_table) {                       // EMA momentum schedule (base_engine.py:160-161): entry of this iteration, last one held
    int64_t i = step - 1;
    i = i < 0 ? 0 : (i >= mm_len ? mm_len - 1 : i);
    mm = mm_table[i];
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_all; i += (int64_t)gridDim.x * blockDim.x) {
    float w = p[i];
    if (i < n_train) {
      float gi = g[i] * gscale + wd * w;
      const float mi = beta1 * m[i] + (1.f - beta1) * gi;
      const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
      m[i] = mi;
      v[i] = vi;
      const float denom = sqrtf(vi) / bc2s + eps;
      w = w - lr_over_bc1 * (mi / denom);
      p[i] = w;
      if (zero_grad) g[i] = 0.f;
    }
    if (teacher) teacher[i] = teacher[i] * mm + w * (1.f - mm);
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

extern "C" int mhimx_adam_ema(void* stream, float* p, const float* g, float* m, float* v, float* teacher, int64_t n_train,
                              int64_t n_all, int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                              float grad_scale, float ema_mm, int32_t zero_grad, const uint64_t* step_dev,
                              const float* mm_table, int64_t mm_len) {
  MHIMX_CHECK_ARG(!mm_table || mm_len > 0, "adam_ema: empty momentum schedule");
  MHIMX_CHECK_ARG(p && g && m && v && n_train >= 0 && n_all >= n_train && (step >= 1 || step_dev), "adam_ema: bad args");
  if (step < 1) step = 1;
  if (n_all == 0) return 0;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const int64_t blocks = cdiv(n_all, 256) < 2048 ? cdiv(n_all, 256) : 2048;
  hipLaunchKernelGGL(adam_ema_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, const_cast<float*>(g), m, v,
                     teacher, n_train, n_all, (float)((double)lr / bc1), (float)sqrt(bc2), beta1, beta2, eps, weight_decay,
                     grad_scale, ema_mm, zero_grad, step_dev, lr, mm_table, mm_len, step);
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
