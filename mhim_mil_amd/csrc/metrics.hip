// metrics.hip — the validation metrics on the device (replaces the torchmetrics MetricCollection of
// engines/metrics.py:125-159 `metrics_base` / :104-123 `get_cls_metrics`, and the resampling loop of the
// DeterministicBootStrapper, engines/metrics.py:35-78).
//
// All B resamples of a bootstrap (blockIdx.y) are evaluated by the same four launches; the logits never leave HBM:
//   1. range flag   : "are all predictions inside [0,1]?" (torchmetrics' format rule: otherwise softmax / sigmoid)
//   2. probabilities: softmax / sigmoid / identity per sample, hard label (first arg-max / p > 0.5), confusion counts
//   3. ROC areas    : exact Mann-Whitney pair counts one-vs-rest, 2*#{pos > neg} + #{pos == neg} as 64-bit integers
//   4. final        : precision / recall / F1 / kappa / accuracies / macro AUROC in fp64 from the integer counts
// Integer counts throughout: the only floating-point work is the probability transform and the final ratios.
#include <math.h>

#include "common.hpp"

namespace mhimx {

constexpr int MET_MAXC = 16;

__global__ void met_flag_kernel(const float* __restrict__ logits, const int64_t* __restrict__ idx, int64_t n, int C, int ld, int col0,
                                int ncol, int* __restrict__ outside) {
  const int b = blockIdx.y;
  int bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx ? idx[(int64_t)b * n + i] : i;
    for (int c = 0; c < ncol; ++c) {
      const float v = logits[r * ld + col0 + c];
      if (!(v >= 0.f && v <= 1.f)) bad = 1;
    }
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&outside[b], 1);
}

// prob [B][n][ncol], lab [B][n] (target), conf [B][C][C] (conf[target][pred])
__global__ void met_prob_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, const int64_t* __restrict__ idx,
                                int64_t n, int C, int ld, int binary, const int* __restrict__ outside, float* __restrict__ prob,
                                int* __restrict__ lab, unsigned long long* __restrict__ conf) {
  const int b = blockIdx.y;
  const bool tr = outside[b] != 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx ? idx[(int64_t)b * n + i] : i;
    const int y = (int)labels[r];
    int pred;
    if (binary) {
      float s = logits[r * ld + (ld > 1 ? 1 : 0)];
      if (tr) s = 1.f / (1.f + expf(-s));
      prob[((int64_t)b * n + i)] = s;
      pred = s > 0.5f ? 1 : 0;
    } else {
      float v[MET_MAXC];
      float m = -INFINITY;
      for (int c = 0; c < C; ++c) { v[c] = logits[r * ld + c]; m = fmaxf(m, v[c]); }
      if (tr) {
        float sum = 0.f;
        for (int c = 0; c < C; ++c) { v[c] = expf(v[c] - m); sum += v[c]; }
        for (int c = 0; c < C; ++c) v[c] = v[c] / sum;
      }
      pred = 0;
      for (int c = 0; c < C; ++c) {
        prob[((int64_t)b * n + i) * C + c] = v[c];
        if (v[c] > v[pred]) pred = c;                      // first maximum
      }
    }
    lab[(int64_t)b * n + i] = y;
    if (y >= 0 && y < C) atomicAdd(&conf[((int64_t)b * C + y) * C + pred], 1ull);
  }
}

// pairs[b][c] += sum over positives i of class c in this block's slice, over all negatives j: 2*[p_i > p_j] + [p_i == p_j]
__global__ __launch_bounds__(256) void met_auc_kernel(const float* __restrict__ prob, const int* __restrict__ lab, int64_t n, int ncol,
                                                      int binary, unsigned long long* __restrict__ pairs) {
  const int b = blockIdx.y, c = blockIdx.z;
  const float* p = prob + (int64_t)b * n * ncol + (binary ? 0 : c);
  const int* y = lab + (int64_t)b * n;
  const int cls = binary ? 1 : c;
  __shared__ float sp[1024];
  __shared__ int sy[1024];
  unsigned long long acc = 0;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool mine = i < n && y[i] == cls;
  const float pi = mine ? p[i * ncol] : 0.f;
  for (int64_t j0 = 0; j0 < n; j0 += 1024) {
    __syncthreads();
    for (int t = threadIdx.x; t < 1024; t += 256) {
      const int64_t j = j0 + t;
      sp[t] = j < n ? p[j * ncol] : 0.f;
      sy[t] = j < n ? (y[j] == cls ? 1 : 0) : 1;             // 1: not a negative (positive or padding)
    }
    __syncthreads();
    if (mine) {
      unsigned cnt = 0;
      for (int t = 0; t < 1024; ++t)
        if (!sy[t]) cnt += pi > sp[t] ? 2u : (pi == sp[t] ? 1u : 0u);
      acc += cnt;
    }
  }
  // block sum -> one atomic
  __shared__ unsigned long long red[4];
  unsigned long long v = acc;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long s = red[0] + red[1] + red[2] + red[3];
    if (s) atomicAdd(&pairs[(int64_t)b * MET_MAXC + c], s);
  }
}

MHIMX_DEV double met_safe(double a, double b) { return b > 0.0 ? a / b : 0.0; }

// out [B][7]: Acc, AUC, Precision, Recall, F1, CK, Acc_micro
__global__ void met_final_kernel(const unsigned long long* __restrict__ conf, const unsigned long long* __restrict__ pairs, int C,
                                 int binary, int B, float* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const unsigned long long* cf = conf + (int64_t)b * C * C;
  double tp[MET_MAXC], fp[MET_MAXC], fn[MET_MAXC], row[MET_MAXC], col[MET_MAXC];
  double n = 0.0, diag = 0.0;
  for (int c = 0; c < C; ++c) { row[c] = 0.0; col[c] = 0.0; }
  for (int t = 0; t < C; ++t)
    for (int p = 0; p < C; ++p) {
      const double v = (double)cf[t * C + p];
      row[t] += v; col[p] += v; n += v;
      if (t == p) diag += v;
    }
  for (int c = 0; c < C; ++c) { tp[c] = (double)cf[c * C + c]; fp[c] = col[c] - tp[c]; fn[c] = row[c] - tp[c]; }
  double pe = 0.0;
  for (int c = 0; c < C; ++c) pe += row[c] * col[c];
  pe = n > 0.0 ? pe / (n * n) : 0.0;
  const double po = met_safe(diag, n);
  const double ck = (n > 0.0 && pe != 1.0) ? (po - pe) / (1.0 - pe) : 0.0;
  double acc, auc, prec, rec, f1;
  if (binary) {
    acc = po;
    prec = met_safe(tp[1], tp[1] + fp[1]);
    rec = met_safe(tp[1], tp[1] + fn[1]);
    f1 = met_safe(2.0 * tp[1], 2.0 * tp[1] + fp[1] + fn[1]);
    const double P = row[1], N = row[0];
    auc = (P > 0.0 && N > 0.0) ? (double)pairs[(int64_t)b * MET_MAXC + 0] / (2.0 * P * N) : 0.0;
  } else {
    double w = 0.0, sp = 0.0, sr = 0.0, sf = 0.0, sa = 0.0, wa = 0.0;
    for (int c = 0; c < C; ++c) {
      if (tp[c] + fp[c] + fn[c] > 0.0) {
        w += 1.0;
        sp += met_safe(tp[c], tp[c] + fp[c]);
        sr += met_safe(tp[c], tp[c] + fn[c]);
        sf += met_safe(2.0 * tp[c], 2.0 * tp[c] + fp[c] + fn[c]);
      }
      const double P = row[c], N = n - row[c];
      if (P > 0.0 && N > 0.0) { sa += (double)pairs[(int64_t)b * MET_MAXC + c] / (2.0 * P * N); wa += 1.0; }
    }
    prec = met_safe(sp, w); rec = met_safe(sr, w); f1 = met_safe(sf, w);
    acc = rec;
    auc = met_safe(sa, wa);
  }
  float* o = out + (int64_t)b * 7;
  o[0] = (float)acc; o[1] = (float)auc; o[2] = (float)prec; o[3] = (float)rec; o[4] = (float)f1; o[5] = (float)ck; o[6] = (float)po;
}

}  // namespace mhimx

using namespace mhimx;

extern "C" int64_t mhimx_cls_metrics_ws_bytes(int64_t n, int64_t C, int64_t B) {
  if (B < 1) B = 1;
  int64_t bytes = 0;
  bytes += align_up(B * 4, 256);                                   // outside flags
  bytes += align_up(B * n * C * 4, 256);                           // probabilities
  bytes += align_up(B * n * 4, 256);                               // labels
  bytes += align_up(B * C * C * 8, 256);                           // confusion counts
  bytes += align_up(B * MET_MAXC * 8, 256);                        // pair counts
  return bytes;
}

extern "C" int mhimx_cls_metrics(void* stream, const float* logits, int64_t ld, const int64_t* labels, int64_t n, int64_t C,
                                 int32_t bin_metric, const int64_t* sample_idx, int64_t B, float* out, void* ws, int64_t ws_bytes) {
  MHIMX_CHECK_ARG(logits && labels && out && ws && n > 0 && C >= 2 && C <= MET_MAXC && ld >= 1, "cls_metrics: bad args (2 <= C <= 16)");
  MHIMX_CHECK_ARG(!bin_metric || C == 2, "cls_metrics: bin_metric needs two classes");
  MHIMX_CHECK_ARG(B >= 1 && B <= 65535 && (B == 1 || sample_idx), "cls_metrics: B resamples need sample_idx [B, n]");
  MHIMX_CHECK_ARG(ws_bytes >= mhimx_cls_metrics_ws_bytes(n, C, B), "cls_metrics: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  char* p = (char*)ws;
  int* outside = (int*)p; p += align_up(B * 4, 256);
  float* prob = (float*)p; p += align_up(B * n * C * 4, 256);
  int* lab = (int*)p; p += align_up(B * n * 4, 256);
  unsigned long long* conf = (unsigned long long*)p; p += align_up(B * C * C * 8, 256);
  unsigned long long* pairs = (unsigned long long*)p;
  MHIMX_HIP(hipMemsetAsync(outside, 0, (size_t)B * 4, st));
  MHIMX_HIP(hipMemsetAsync(conf, 0, (size_t)(B * C * C * 8), st));
  MHIMX_HIP(hipMemsetAsync(pairs, 0, (size_t)(B * MET_MAXC * 8), st));
  const int binary = bin_metric ? 1 : 0;
  const int ncol = binary ? 1 : (int)C;
  const unsigned gx = (unsigned)(cdiv(n, 256) < 1024 ? cdiv(n, 256) : 1024);
  hipLaunchKernelGGL(met_flag_kernel, dim3(gx, (unsigned)B), dim3(256), 0, st, logits, sample_idx, n, (int)C, (int)ld,
                     binary ? (ld > 1 ? 1 : 0) : 0, ncol, outside);
  MHIMX_LAUNCH_CHECK();
  hipLaunchKernelGGL(met_prob_kernel, dim3(gx, (unsigned)B), dim3(256), 0, st, logits, labels, sample_idx, n, (int)C, (int)ld, binary,
                     outside, prob, lab, conf);
  MHIMX_LAUNCH_CHECK();
  hipLaunchKernelGGL(met_auc_kernel, dim3((unsigned)cdiv(n, 256), (unsigned)B, (unsigned)(binary ? 1 : C)), dim3(256), 0, st, prob, lab, n,
                     ncol, binary, pairs);
  MHIMX_LAUNCH_CHECK();
  hipLaunchKernelGGL(met_final_kernel, dim3((unsigned)cdiv(B, 64)), dim3(64), 0, st, conf, pairs, (int)C, binary, (int)B, out);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
