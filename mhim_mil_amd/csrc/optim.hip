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
    int64_t step_host, double ln_beta1, double ln_beta2) {
  __shared__ float sc[3];
  if (threadIdx.x == 0) {               // the fp64 pow()s of the bias corrections: once per block, not once per thread
    int64_t step = step_host;
    if (step_dev) {                     // graph replay: the step count lives on the device
      step = (int64_t)step_dev[0];
      const double t = (double)step;
      lr_over_bc1 = (float)((double)lr / (1.0 - exp(t * ln_beta1)));       // beta^t = e^(t ln beta), ln beta from the host in fp64
      bc2s = (float)sqrt(1.0 - exp(t * ln_beta2));
    }
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
  const int64_t n4 = n_all / 4;
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                        reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(teacher)) & 15) == 0;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q * 4 < n_all; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i0 = q * 4;
    if (vec_ok && q < n4 && (i0 + 4 <= n_train || i0 >= n_train)) {
      float4 w = reinterpret_cast<float4*>(p)[q];
      if (i0 < n_train) {
        float4 gi = reinterpret_cast<float4*>(g)[q], mi = reinterpret_cast<float4*>(m)[q], vi = reinterpret_cast<float4*>(v)[q];
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
        float gi = g[i], mi = m[i], vi = v[i];
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
  const int64_t blocks = cdiv(n_all, 1024) < 2048 ? cdiv(n_all, 1024) : 2048;
  hipLaunchKernelGGL(adam_ema_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, const_cast<float*>(g), m, v,
                     teacher, n_train, n_all, (float)((double)lr / bc1), (float)sqrt(bc2), beta1, beta2, eps, weight_decay,
                     grad_scale, ema_mm, zero_grad, step_dev, lr, mm_table, mm_len, step, log((double)beta1), log((double)beta2));
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
