// attn_ops.hip — non-GEMM pieces of the Nystrom / TransMIL encoder, forward and
// backward: row softmax, landmark means, pseudo-inverse initialisation, a*I + b*X, the depth-wise residual
// convolution along tokens, PPEG's 7x7+5x5+3x3 depth-wise grid convolution, and small element-wise helpers.
// All are streaming (HBM-bound) kernels: coalesced along the 512-wide channel axis, wave/block reductions by DPP.
#include <math.h>

#include "common.hpp"

namespace mhimx {
int transpose(hipStream_t st, const float* in, float* out, int64_t R, int64_t C);   // gemm.hip

constexpr int AT = 256;

MHIMX_DEV float blk_sum4(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
MHIMX_DEV float blk_max4(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ------------------------------------------------------------------------------------------------
// softmax over the last dimension of x[R, L] (scaled: y = softmax(alpha * x)), block per row
// replaces: `.softmax(dim=-1)` at nystrom_attention.py:130
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AT) void softmax_rows_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t R,
                                                              int64_t L, float alpha) {
  __shared__ float red[4];
  for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
    const float* xr = x + r * L;
    float* yr = y + r * L;
    float m = -INFINITY;
    for (int64_t i = threadIdx.x; i < L; i += AT) m = fmaxf(m, xr[i] * alpha);
    m = blk_max4(m, red);
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < L; i += AT) s += __expf(xr[i] * alpha - m);
    s = blk_sum4(s, red);
    const float inv = 1.f / s;
    for (int64_t i = threadIdx.x; i < L; i += AT) yr[i] = __expf(xr[i] * alpha - m) * inv;
    __syncthreads();
  }
}
// long rows, 16-byte aligned (L % 4 == 0): ONE read pass for max and sum (online rescaling), 16-byte loads with four in flight, then
// the write pass re-reads the row while it is still in L2 (a 48 500-token row is 194 KB)
__global__ __launch_bounds__(AT) void softmax_rows_fwd_vec_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t R,
                                                                  int64_t L, float alpha) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  __shared__ float red[4];
  const int64_t L4 = L / 4;
  for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
    const f4* xr = reinterpret_cast<const f4*>(x + r * L);
    f4* yr = reinterpret_cast<f4*>(y + r * L);
    float m = -INFINITY, s = 0.f;
    for (int64_t i = threadIdx.x; i < L4; i += 4 * AT) {
      f4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = (i + q * AT) < L4 ? xr[i + q * AT] * alpha : f4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      float mq = m;
#pragma unroll
      for (int q = 0; q < 4; ++q) mq = fmaxf(mq, fmaxf(fmaxf(v[q][0], v[q][1]), fmaxf(v[q][2], v[q][3])));
      float add = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) add += (__expf(v[q][0] - mq) + __expf(v[q][1] - mq)) + (__expf(v[q][2] - mq) + __expf(v[q][3] - mq));
      s = (m == -INFINITY ? 0.f : s * __expf(m - mq)) + add;
      m = mq;
    }
    const float mb = blk_max4(m, red);
    s = blk_sum4(m == -INFINITY ? 0.f : s * __expf(m - mb), red);
    const float inv = 1.f / s;
    for (int64_t i = threadIdx.x; i < L4; i += AT) {
      const f4 v = xr[i] * alpha;
      yr[i] = f4{__expf(v[0] - mb), __expf(v[1] - mb), __expf(v[2] - mb), __expf(v[3] - mb)} * inv;
    }
    __syncthreads();
  }
}
// wave-per-row variant for short rows (L <= 1024): 4 rows per block
__global__ __launch_bounds__(AT) void softmax_rows_fwd_short_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t R,
                                                                    int L, float alpha) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < R; r += (int64_t)gridDim.x * 4) {
    const float* xr = x + r * L;
    float v[16];
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int i = lane + 64 * q;
      v[q] = i < L ? xr[i] * alpha : -INFINITY;
      m = fmaxf(m, v[q]);
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      v[q] = (lane + 64 * q) < L ? __expf(v[q] - m) : 0.f;
      s += v[q];
    }
    const float inv = 1.f / wave_sum(s);
#pragma unroll
    for (int q = 0; q < 16; ++q)
      if ((lane + 64 * q) < L) y[r * L + lane + 64 * q] = v[q] * inv;
  }
}
// dx = alpha * y * (dy - sum(y*dy))
__global__ __launch_bounds__(AT) void softmax_rows_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                              float* __restrict__ dx, int64_t R, int64_t L, float alpha) {
  __shared__ float red[4];
  for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
    const float* yr = y + r * L;
    const float* gr = dy + r * L;
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < L; i += AT) s += yr[i] * gr[i];
    s = blk_sum4(s, red);
    for (int64_t i = threadIdx.x; i < L; i += AT) dx[r * L + i] = alpha * yr[i] * (gr[i] - s);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// landmark means: out[j, c] = mean_{t in [j*l, (j+1)*l)} x[t*ldx + c]   (nystrom_attention.py:93-109)
// bwd: dx[t, c] (+)= dout[t / l, c] / l
// ------------------------------------------------------------------------------------------------
__global__ void landmark_fwd_kernel(const float* __restrict__ x, int64_t ldx, int l, int C, float* __restrict__ out) {
  const int j = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float acc = 0.f;
    for (int t = 0; t < l; ++t) acc += x[((int64_t)j * l + t) * ldx + c];
    out[(int64_t)j * C + c] = acc / (float)l;
  }
}
// the same sums for 16-byte aligned rows with C % 256 == 0: block = (landmark, 256 columns); its four waves take the rows t = w mod 4
// with four independent 16-byte loads in flight each (the scalar form above is one l-long dependent chain per thread: latency bound)
__global__ __launch_bounds__(256) void landmark_fwd_vec_kernel(const float* __restrict__ x, int64_t ldx, int l, int C, float* __restrict__ out) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  __shared__ f4 part[3][64];
  const int j = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const f4* base = reinterpret_cast<const f4*>(x + (int64_t)j * l * ldx + blockIdx.y * 256) + lane;
  const int64_t ld4 = ldx / 4;
  f4 a0 = f4{0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  int t = w;
  for (; t + 12 < l; t += 16) {
    a0 += base[(int64_t)t * ld4];
    a1 += base[(int64_t)(t + 4) * ld4];
    a2 += base[(int64_t)(t + 8) * ld4];
    a3 += base[(int64_t)(t + 12) * ld4];
  }
  for (; t < l; t += 4) a0 += base[(int64_t)t * ld4];
  f4 acc = (a0 + a1) + (a2 + a3);
  if (w > 0) part[w - 1][lane] = acc;
  __syncthreads();
  if (w == 0) {
    acc = (acc + part[0][lane]) + (part[1][lane] + part[2][lane]);
    const float inv = 1.f / (float)l;
    reinterpret_cast<f4*>(out + (int64_t)j * C + blockIdx.y * 256)[lane] = acc * inv;
  }
}
__global__ void landmark_bwd_kernel(const float* __restrict__ dout, int l, int C, float* __restrict__ dx, int64_t ldx, int64_t T,
                                    int accumulate) {
  const float inv = 1.f / (float)l;
  for (int64_t t = blockIdx.x; t < T; t += gridDim.x)
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const float v = dout[(t / l) * C + c] * inv;
      float* p = dx + t * ldx + c;
      *p = accumulate ? *p + v : v;
    }
}

// the same, 16 bytes per lane (C % 4 == 0, 16-byte aligned rows): one workgroup pass per token row of C / 4 float4 (the scalar form ran the
// 400 MB read-modify-write of a c3 layer at 3.3 TB/s)
__global__ __launch_bounds__(256) void landmark_bwd_vec_kernel(const float* __restrict__ dout, int l, int C4, float* __restrict__ dx, int64_t ldx,
                                                              int64_t T, int accumulate) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const float inv = 1.f / (float)l;
  for (int64_t t = blockIdx.x; t < T; t += gridDim.x) {
    const f4* src = reinterpret_cast<const f4*>(dout + (t / l) * (int64_t)C4 * 4);
    f4* p = reinterpret_cast<f4*>(dx + t * ldx);
    for (int c = threadIdx.x; c < C4; c += 256) {
      const f4 v = src[c] * inv;
      p[c] = accumulate ? p[c] + v : v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// y = a*I + b*x on a batch of square matrices [B, n, n]   (the 13I - ..., 15I - ..., 7I - ... of nystrom:25)
// ------------------------------------------------------------------------------------------------
__global__ void affine_ident_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t total, int n, float a, float b) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i % ((int64_t)n * n);
    const int r = (int)(e / n), c = (int)(e % n);
    y[i] = b * x[i] + (r == c ? a : 0.f);
  }
}

// y = alpha*x + beta*y  (residual adds, gradient accumulation of same-shaped tensors)
__global__ void axpby_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float alpha, float beta) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = alpha * x[i] + (beta == 0.f ? 0.f : beta * y[i]);
}

// ------------------------------------------------------------------------------------------------
// column softmax of x[M,C] (C <= 16 columns, normalised over the M rows): DSMIL's attention over instances
// (mhim_modules/baseline.py:147).  One block per column.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void softmax_cols_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t M, int C,
                                                                float alpha) {
  __shared__ float red[16];
  const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float m = -INFINITY;
  for (int64_t r = threadIdx.x; r < M; r += 1024) m = fmaxf(m, x[r * C + c] * alpha);
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = red[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
  __syncthreads();
  float s = 0.f;
  for (int64_t r = threadIdx.x; r < M; r += 1024) s += __expf(x[r * C + c] * alpha - m);
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) tot += red[w];
  const float inv = 1.f / tot;
  for (int64_t r = threadIdx.x; r < M; r += 1024) y[r * C + c] = __expf(x[r * C + c] * alpha - m) * inv;
}
// dx = alpha * y * (dy - sum_rows(y * dy)) per column
__global__ __launch_bounds__(1024) void softmax_cols_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                                float* __restrict__ dx, int64_t M, int C, float alpha) {
  __shared__ float red[16];
  const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s = 0.f;
  for (int64_t r = threadIdx.x; r < M; r += 1024) s += y[r * C + c] * dy[r * C + c];
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) tot += red[w];
  for (int64_t r = threadIdx.x; r < M; r += 1024) dx[r * C + c] = alpha * y[r * C + c] * (dy[r * C + c] - tot);
}

// ------------------------------------------------------------------------------------------------
// pseudo-inverse initialisation (nystrom_attention.py:15-18): z0 = a^T / (max_{b,i} sum_j |a_ij| * max_{b,j} sum_i |a_ij|)
// with GLOBAL maxima over the batch of heads.  stats: [0]=c (max row sum), [1]=r (max col sum), [2]=argmax row (b*n+i),
// [3]=argmax col (b*n+j), stored as floats.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AT) void pinv_sums_kernel(const float* __restrict__ a, int B, int n, float* __restrict__ rowsum,
                                                       float* __restrict__ colsum) {
  // one block per (b, i): row sum; and per (b, j): col sum (second half of the grid)
  __shared__ float red[4];
  const int idx = blockIdx.x;
  const int total = B * n;
  const bool is_col = idx >= total;
  const int q = is_col ? idx - total : idx;
  const int b = q / n, i = q % n;
  float s = 0.f;
  for (int t = threadIdx.x; t < n; t += AT) s += fabsf(is_col ? a[((int64_t)b * n + t) * n + i] : a[((int64_t)b * n + i) * n + t]);
  s = blk_sum4(s, red);
  if (threadIdx.x == 0) (is_col ? colsum : rowsum)[q] = s;
}
__global__ __launch_bounds__(AT) void pinv_argmax_kernel(const float* __restrict__ rowsum, const float* __restrict__ colsum, int total,
                                                         float* __restrict__ stats) {
  __shared__ float bv[AT];
  __shared__ int bi[AT];
  for (int which = 0; which < 2; ++which) {
    const float* v = which ? colsum : rowsum;
    float best = -INFINITY;
    int arg = 0;
    for (int i = threadIdx.x; i < total; i += AT)
      if (v[i] > best) { best = v[i]; arg = i; }
    bv[threadIdx.x] = best; bi[threadIdx.x] = arg;
    __syncthreads();
    for (int o = AT / 2; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) {
        const float ov = bv[threadIdx.x + o];
        const int oi = bi[threadIdx.x + o];
        if (ov > bv[threadIdx.x] || (ov == bv[threadIdx.x] && oi < bi[threadIdx.x])) { bv[threadIdx.x] = ov; bi[threadIdx.x] = oi; }
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) { stats[which] = bv[0]; stats[2 + which] = (float)bi[0]; }
    __syncthreads();
  }
}
// z0[b,j,i] = a[b,i,j] / (c*r)
__global__ void pinv_init_kernel(const float* __restrict__ a, const float* __restrict__ stats, int B, int n, float* __restrict__ z) {
  __shared__ float tile[32][33];
  const float s = 1.f / (stats[0] * stats[1]);
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) tile[i][threadIdx.x] = a[((int64_t)b * n + r0 + i) * n + c0 + threadIdx.x];
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) z[((int64_t)b * n + c0 + i) * n + r0 + threadIdx.x] = tile[threadIdx.x][i] * s;
}
// backward: da = dz^T * s  - g * s * (1/c on the argmax row, 1/r on the argmax column),  g = sum(dz * z0)/s ... see host
__global__ __launch_bounds__(AT) void dot_total_kernel(const float* __restrict__ x, const float* __restrict__ y, int64_t n,
                                                       float* __restrict__ part) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * AT + threadIdx.x; i < n; i += (int64_t)gridDim.x * AT) s += x[i] * y[i];
  s = blk_sum4(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void pinv_init_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ z0, const float* __restrict__ stats,
                                     const float* __restrict__ part, int npart, int B, int n, float* __restrict__ da) {
  // g = <dz, z0> = sum over partials; d(1/(c r)) terms land on the arg-max row / column of |a| with sign(a) = sign(z0^T).
  // (After a softmax all row sums tie at 1 and the arg-max row is rounding noise — harmless: a constant added to a whole
  //  row of d a is annihilated by the softmax backward.)
  __shared__ float gs;
  if (threadIdx.x == 0 && threadIdx.y == 0) {
    float g = 0.f;
    for (int i = 0; i < npart; ++i) g += part[i];
    gs = g;
  }
  __shared__ float tile[32][33], ztile[32][33];
  __syncthreads();
  const float c = stats[0], r = stats[1], s = 1.f / (c * r);
  const int arow = (int)stats[2], acol = (int)stats[3];
  const float g = gs;                          // = sum(dz * z0);  dL/ds = g / s ; ds/dc = -s/c ; ds/dr = -s/r
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    tile[i][threadIdx.x] = dz[((int64_t)b * n + r0 + i) * n + c0 + threadIdx.x];
    ztile[i][threadIdx.x] = z0[((int64_t)b * n + r0 + i) * n + c0 + threadIdx.x];
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int ai = c0 + i, aj = r0 + threadIdx.x;      // element a[b, ai, aj] <- dz[b, aj, ai]
    float v = tile[threadIdx.x][i] * s;
    const float zz = ztile[threadIdx.x][i];
    const float sg = zz > 0.f ? 1.f : (zz < 0.f ? -1.f : 0.f);
    if (b * n + ai == arow) v -= sg * g / c;
    if (b * n + aj == acol) v -= sg * g / r;
    da[((int64_t)b * n + ai) * n + aj] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// residual depth-wise convolution along tokens (nystrom_attention.py:59-63,135-136): Conv2d(h,h,(33,1),groups=h), zero pad 16
//   out[t, c] (+)= sum_tau w[c / dh, tau] * v[t + tau - P, c]          v, out: [T, C] with row pitches ldv, ldo
// bwd: dv[t, c] (+)= sum_tau w[h, tau] * dout[t - tau + P, c] ;  dw[h, tau] = sum_{t,c in h} dout[t,c] * v[t+tau-P, c]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AT) void resconv_fwd_kernel(const float* __restrict__ v, int64_t ldv, const float* __restrict__ w, int KS,
                                                         int dh, int64_t T, int C, float* __restrict__ out, int64_t ldo,
                                                         int accumulate, int flip) {
  const int P = KS / 2;
  for (int64_t t = blockIdx.x; t < T; t += gridDim.x)
    for (int c = threadIdx.x; c < C; c += AT) {
      const float* wh = w + (c / dh) * KS;
      float acc = 0.f;
      for (int tau = 0; tau < KS; ++tau) {
        const int64_t tt = flip ? t - tau + P : t + tau - P;
        if (tt >= 0 && tt < T) acc += wh[tau] * v[tt * ldv + c];
      }
      float* p = out + t * ldo + c;
      *p = accumulate ? *p + acc : acc;
    }
}
// partial dw per block of tokens: part[blk][h*KS + tau]
__global__ __launch_bounds__(AT) void resconv_dw_kernel(const float* __restrict__ dout, int64_t ldo, const float* __restrict__ v,
                                                        int64_t ldv, int KS, int dh, int64_t T, int C, int64_t chunk,
                                                        float* __restrict__ part) {
  extern __shared__ float sm[];              // [heads*KS]
  const int heads = C / dh, P = KS / 2;
  for (int i = threadIdx.x; i < heads * KS; i += AT) sm[i] = 0.f;
  __syncthreads();
  const int64_t t0 = (int64_t)blockIdx.x * chunk, t1 = t0 + chunk < T ? t0 + chunk : T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // wave w handles heads w, w+4, ...; lane = channel inside the head (dh == 64)
  for (int h = wave; h < heads; h += 4) {
    const int c = h * dh + lane;
    for (int tau = 0; tau < KS; ++tau) {
      float acc = 0.f;
      for (int64_t t = t0; t < t1; ++t) {
        const int64_t tt = t + tau - P;
        if (tt >= 0 && tt < T) acc += dout[t * ldo + c] * v[tt * ldv + c];
      }
      acc = wave_sum(acc);
      if (lane == 0) sm[h * KS + tau] = acc;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < heads * KS; i += AT) part[(int64_t)blockIdx.x * heads * KS + i] = sm[i];
}

// ------------------------------------------------------------------------------------------------
// PPEG (emb_position.py:85-120): tokens [N, C] laid on an H x H grid (first H*H-N tokens appended again: wrap;
// if H < 7 zero-padded to 7x7), y = x + dw7(x) + dw5(x) + dw3(x) (+ biases), first N grid cells returned.
// One combined 7x7 kernel per channel: wc = w7 + pad(w5) + pad(w3) (+1 at the centre for the identity); bias = b7+b5+b3.
// ------------------------------------------------------------------------------------------------
MHIMX_DEV int64_t ppeg_src(int64_t cell, int64_t N, int64_t wrapN) {       // grid cell -> source token or -1 (zero cell)
  if (cell < N) return cell;
  if (cell < wrapN) return cell - N;
  return -1;
}
__global__ __launch_bounds__(AT) void ppeg_fwd_kernel(const float* __restrict__ x, int64_t N, int C, int H, int64_t wrapN,
                                                      const float* __restrict__ wc /*[C,49]*/, const float* __restrict__ bc,
                                                      float* __restrict__ y, int flip) {
  for (int64_t cell = blockIdx.x; cell < N; cell += gridDim.x) {
    const int gy = (int)(cell / H), gx = (int)(cell % H);
    for (int c = threadIdx.x; c < C; c += AT) {
      float acc = flip ? 0.f : bc[c];
      for (int dy = -3; dy <= 3; ++dy) {
        const int yy = gy + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dx = -3; dx <= 3; ++dx) {
          const int xx = gx + dx;
          if (xx < 0 || xx >= H) continue;
          const int64_t src = ppeg_src((int64_t)yy * H + xx, N, wrapN);
          if (src < 0) continue;
          const int tap = flip ? (3 - dy) * 7 + (3 - dx) : (dy + 3) * 7 + (dx + 3);
          acc += wc[c * 49 + tap] * x[src * C + c];
        }
      }
      y[cell * C + c] = acc;
    }
  }
}
// The wrapped cells (N <= cell < wrapN) also receive inputs/gradients: handled by a second pass over those cells that
// ACCUMULATES into the first tokens (backward: dx[cell-N] += sum over the cell's neighbourhood of dy * w flipped).
__global__ __launch_bounds__(AT) void ppeg_bwd_dx_kernel(const float* __restrict__ dy, int64_t N, int C, int H, int64_t wrapN,
                                                         const float* __restrict__ wc, float* __restrict__ dx, int64_t cell0,
                                                         int64_t cell1, int accumulate) {
  // dx for the source token of grid cell `cell`: sum over output cells o (< N: only those are returned) in the 7x7
  // neighbourhood of dy[o] * wc[tap(o -> cell)]
  for (int64_t cell = cell0 + blockIdx.x; cell < cell1; cell += gridDim.x) {
    const int64_t src = ppeg_src(cell, N, wrapN);
    if (src < 0) continue;
    const int gy = (int)(cell / H), gx = (int)(cell % H);
    for (int c = threadIdx.x; c < C; c += AT) {
      float acc = 0.f;
      for (int dyy = -3; dyy <= 3; ++dyy) {
        const int oy = gy - dyy;
        if (oy < 0 || oy >= H) continue;
        for (int dxx = -3; dxx <= 3; ++dxx) {
          const int ox = gx - dxx;
          if (ox < 0 || ox >= H) continue;
          const int64_t o = (int64_t)oy * H + ox;
          if (o >= N) continue;
          acc += wc[c * 49 + (dyy + 3) * 7 + (dxx + 3)] * dy[o * C + c];
        }
      }
      float* p = dx + src * C + c;
      *p = accumulate ? *p + acc : acc;
    }
  }
}
// dwc partials: part[blk][c*49 + tap] = sum over the block's output cells of dy[o,c] * x[src(o + tap)]
__global__ __launch_bounds__(AT) void ppeg_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x, int64_t N, int C, int H,
                                                     int64_t wrapN, int64_t chunk, float* __restrict__ part,
                                                     float* __restrict__ part_b) {
  const int64_t o0 = (int64_t)blockIdx.x * chunk, o1 = o0 + chunk < N ? o0 + chunk : N;
  for (int c = threadIdx.x; c < C; c += AT) {
    float acc[49];
#pragma unroll
    for (int i = 0; i < 49; ++i) acc[i] = 0.f;
    float ab = 0.f;
    for (int64_t o = o0; o < o1; ++o) {
      const float g = dy[o * C + c];
      ab += g;
      const int gy = (int)(o / H), gx = (int)(o % H);
#pragma unroll
      for (int dyy = -3; dyy <= 3; ++dyy)
#pragma unroll
        for (int dxx = -3; dxx <= 3; ++dxx) {
          const int yy = gy + dyy, xx = gx + dxx;
          if (yy < 0 || yy >= H || xx < 0 || xx >= H) continue;
          const int64_t src = ppeg_src((int64_t)yy * H + xx, N, wrapN);
          if (src < 0) continue;
          acc[(dyy + 3) * 7 + (dxx + 3)] += g * x[src * C + c];
        }
    }
#pragma unroll
    for (int i = 0; i < 49; ++i) part[((int64_t)blockIdx.x * C + c) * 49 + i] = acc[i];
    part_b[(int64_t)blockIdx.x * C + c] = ab;
  }
}

// ------------------------------------------------------------------------------------------------
// Strip forms of the stencils (the production path).  A thread owns ONE channel and a strip of consecutive output
// positions; the input window slides through registers so that every input value is loaded once per strip instead of
// once per tap, all tap indices are compile-time constants and the lanes of a wave read consecutive channels (coalesced).
// ------------------------------------------------------------------------------------------------
// (Round 3: FOUR channels per thread - 16-byte loads and stores, 128 threads per 512-channel row - measured SLOWER: c3 9.00 vs 8.90 ms.)
constexpr int RS = 16;                  // residual conv: outputs per thread along the token axis
template <int KS>
__global__ __launch_bounds__(AT) void resconv_strip_kernel(const float* __restrict__ v, int64_t ldv, const float* __restrict__ w, int dh,
                                                           int64_t T, int C, float* __restrict__ out, int64_t ldo, int accumulate,
                                                           int flip) {
  constexpr int P = KS / 2;
  const int c = blockIdx.y * AT + threadIdx.x;
  if (c >= C) return;
  const int64_t t0 = (int64_t)blockIdx.x * RS;
  const float* wh = w + (c / dh) * KS;
  float wr[KS];
#pragma unroll
  for (int tau = 0; tau < KS; ++tau) wr[tau] = flip ? wh[KS - 1 - tau] : wh[tau];      // flip: the transposed stencil
  float acc[RS];
#pragma unroll
  for (int j = 0; j < RS; ++j) acc[j] = 0.f;
#pragma unroll
  for (int r = 0; r < RS + KS - 1; ++r) {
    const int64_t tt = t0 - P + r;
    const float x = (tt >= 0 && tt < T) ? v[tt * ldv + c] : 0.f;
#pragma unroll
    for (int j = 0; j < RS; ++j)
      if (r - j >= 0 && r - j < KS) acc[j] = fmaf(wr[r - j], x, acc[j]);
  }
#pragma unroll
  for (int j = 0; j < RS; ++j)
    if (t0 + j < T) {
      float* p = out + (t0 + j) * ldo + c;
      *p = accumulate ? *p + acc[j] : acc[j];
    }
}
// The same stencil with the input window staged ONCE in LDS: a workgroup = 128 output tokens x 64 channels, the 160 input rows (16 of
// halo on each side) arrive by 16-byte loads (a row's 256 bytes from 16 lanes) and every thread slides its 16-output strips over the
// LDS copy (64 lanes = 64 consecutive channels of a row: conflict-free).  The register-sliding form above reads every input row three
// times through L2 (48 rows per 16 outputs).
constexpr int RT_ROWS = 128, RT_CH = 64;
template <int KS>
__global__ __launch_bounds__(256) void resconv_tile_kernel(const float* __restrict__ v, int64_t ldv, const float* __restrict__ w, int dh,
                                                           int64_t T, int C, float* __restrict__ out, int64_t ldo, int accumulate,
                                                           int flip) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  constexpr int P = KS / 2, IN = RT_ROWS + 2 * P;
  __shared__ __attribute__((aligned(16))) float tile[IN * RT_CH];
  const int tid = threadIdx.x, c0 = blockIdx.y * RT_CH;
  const int64_t t0 = (int64_t)blockIdx.x * RT_ROWS;
  for (int idx = tid; idx < IN * (RT_CH / 4); idx += 256) {
    const int row = idx / (RT_CH / 4), c4 = idx % (RT_CH / 4);
    const int64_t tt = t0 - P + row;
    f4 x = {0.f, 0.f, 0.f, 0.f};
    if (tt >= 0 && tt < T) x = *reinterpret_cast<const f4*>(v + tt * ldv + c0 + 4 * c4);
    *reinterpret_cast<f4*>(tile + row * RT_CH + 4 * c4) = x;
  }
  const int c = tid & 63, rg = tid >> 6;
  const float* wh = w + ((c0 + c) / dh) * KS;
  float wr[KS];
#pragma unroll
  for (int tau = 0; tau < KS; ++tau) wr[tau] = flip ? wh[KS - 1 - tau] : wh[tau];
  __syncthreads();
#pragma unroll 1
  for (int strip = 0; strip < 2; ++strip) {
    const int r0 = 32 * rg + 16 * strip;                       // first output row of the strip (tile-relative); its window starts at r0
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll
    for (int r = 0; r < 16 + KS - 1; ++r) {
      const float x = tile[(r0 + r) * RT_CH + c];
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (r - j >= 0 && r - j < KS) acc[j] = fmaf(wr[r - j], x, acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (t0 + r0 + j < T) {
        float* p = out + (t0 + r0 + j) * ldo + c0 + c;
        *p = accumulate ? *p + acc[j] : acc[j];
      }
  }
}
// dw partials: block (x = token chunk, y = 256-channel group = 4 heads of 64); part[x][h*KS + tau]
template <int KS>
__global__ __launch_bounds__(AT) void resconv_dw_strip_kernel(const float* __restrict__ dout, int64_t ldo, const float* __restrict__ v,
                                                              int64_t ldv, int dh, int64_t T, int C, int64_t chunk,
                                                              float* __restrict__ part) {
  constexpr int P = KS / 2;
  const int c = blockIdx.y * AT + threadIdx.x;          // C % 256 == 0 and dh == 64 (checked by the host): wave = one head
  const int64_t c0 = (int64_t)blockIdx.x * chunk, c1 = c0 + chunk < T ? c0 + chunk : T;
  float acc[KS];
#pragma unroll
  for (int tau = 0; tau < KS; ++tau) acc[tau] = 0.f;
  for (int64_t t0 = c0; t0 < c1; t0 += RS) {
    float g[RS];
#pragma unroll
    for (int j = 0; j < RS; ++j) g[j] = (t0 + j < c1) ? dout[(t0 + j) * ldo + c] : 0.f;
#pragma unroll
    for (int r = 0; r < RS + KS - 1; ++r) {
      const int64_t tt = t0 - P + r;
      const float x = (tt >= 0 && tt < T) ? v[tt * ldv + c] : 0.f;
#pragma unroll
      for (int j = 0; j < RS; ++j)
        if (r - j >= 0 && r - j < KS) acc[r - j] = fmaf(g[j], x, acc[r - j]);
    }
  }
  const int heads = C / dh, h = c / dh, lane = threadIdx.x & 63;
#pragma unroll
  for (int tau = 0; tau < KS; ++tau) {
    const float a = wave_sum(acc[tau]);
    if (lane == 0) part[((int64_t)blockIdx.x * heads + h) * KS + tau] = a;
  }
}

constexpr int PS = 8;                   // PPEG: outputs per thread along a grid row
// FLIP = 0: y[cell] = bc + sum_tap wc[tap] * x[src(cell + tap)]            (forward)
// FLIP = 1: dx[cell] = sum_tap wc[flipped tap] * dy[cell + tap] (dy = 0 beyond N)   (backward w.r.t. x for cell < N)
// A thread owns one channel of a PS x PSY patch of grid cells: every input row it loads (PS + 6 values) feeds up to PSY output rows,
// (PSY + 6)(PS + 6) loads for PS PSY outputs = 4.4 per output instead of 12.25 with one-row strips - the stencil is bound by L2 -> L1
// traffic of the re-read neighbour rows (1.2 GB per launch at N = 50 000, C = 512 with one-row strips).
// (Round 3: the same patches fed from an LDS copy of a 22 x 22 x 32-channel window - 1.9 loads per output through L2 - measured SLOWER,
// c3 8.73 vs 8.65 ms: 61 KB of LDS leave 8 waves per CU to hide 140 dependent LDS reads per thread; the 33-tap token-axis convolution
// above, with 3 reads per output, does gain from its LDS window.)
constexpr int PSY = 4;
template <int FLIP>
__global__ __launch_bounds__(AT) void ppeg_strip_kernel(const float* __restrict__ in, int64_t N, int C, int H, int64_t wrapN,
                                                        const float* __restrict__ wc, const float* __restrict__ bc,
                                                        float* __restrict__ out, int nsx) {
  const int c = blockIdx.y * AT + threadIdx.x;
  if (c >= C) return;
  const int gy0 = (blockIdx.x / nsx) * PSY, gx0 = (blockIdx.x % nsx) * PS;
  float w[49];
#pragma unroll
  for (int i = 0; i < 49; ++i) w[i] = wc[c * 49 + i];
  float acc[PSY][PS];
  const float b0 = FLIP ? 0.f : bc[c];
#pragma unroll
  for (int jy = 0; jy < PSY; ++jy)
#pragma unroll
    for (int j = 0; j < PS; ++j) acc[jy][j] = b0;
#pragma unroll
  for (int r = 0; r < PSY + 6; ++r) {
    const int yy = gy0 - 3 + r;
    if (yy < 0 || yy >= H) continue;
    float xin[PS + 6];
#pragma unroll
    for (int i = 0; i < PS + 6; ++i) {
      const int xx = gx0 - 3 + i;
      float val = 0.f;
      if (xx >= 0 && xx < H) {
        const int64_t cell = (int64_t)yy * H + xx;
        const int64_t src = FLIP ? (cell < N ? cell : -1) : ppeg_src(cell, N, wrapN);
        if (src >= 0) val = in[src * C + c];
      }
      xin[i] = val;
    }
#pragma unroll
    for (int jy = 0; jy < PSY; ++jy) {
      const int dy = r - 3 - jy;                              // input row relative to output row gy0 + jy
      if (dy < -3 || dy > 3) continue;
#pragma unroll
      for (int j = 0; j < PS; ++j)
#pragma unroll
        for (int dx = -3; dx <= 3; ++dx) {
          const int tap = FLIP ? (3 - dy) * 7 + (3 - dx) : (dy + 3) * 7 + (dx + 3);
          acc[jy][j] = fmaf(w[tap], xin[j + dx + 3], acc[jy][j]);
        }
    }
  }
#pragma unroll
  for (int jy = 0; jy < PSY; ++jy)
#pragma unroll
    for (int j = 0; j < PS; ++j) {
      const int64_t cell = (int64_t)(gy0 + jy) * H + gx0 + j;
      if (gy0 + jy < H && gx0 + j < H && cell < N) out[cell * C + c] = acc[jy][j];
    }
}
// dwc / dbc partials, coalesced layout: part[blk][tap*C + c], part_b[blk][c]
__global__ __launch_bounds__(AT) void ppeg_dw_strip_kernel(const float* __restrict__ dy, const float* __restrict__ x, int64_t N, int C,
                                                           int H, int64_t wrapN, int nsx, int spb, float* __restrict__ part,
                                                           float* __restrict__ part_b) {
  const int c = blockIdx.y * AT + threadIdx.x;
  if (c >= C) return;
  float acc[49];
#pragma unroll
  for (int i = 0; i < 49; ++i) acc[i] = 0.f;
  float ab = 0.f;
  const int npatch = ((H + PSY - 1) / PSY) * nsx;            // PS x PSY patches, row-major (as the stencil kernel: input rows are reused)
  for (int s = blockIdx.x * spb; s < (int)(blockIdx.x + 1) * spb && s < npatch; ++s) {
    const int gy0 = (s / nsx) * PSY, gx0 = (s % nsx) * PS;
    if ((int64_t)gy0 * H + gx0 >= N) break;
    float g[PSY][PS];
#pragma unroll
    for (int jy = 0; jy < PSY; ++jy)
#pragma unroll
      for (int j = 0; j < PS; ++j) {
        const int64_t cell = (int64_t)(gy0 + jy) * H + gx0 + j;
        g[jy][j] = (gy0 + jy < H && gx0 + j < H && cell < N) ? dy[cell * C + c] : 0.f;
        ab += g[jy][j];
      }
#pragma unroll
    for (int r = 0; r < PSY + 6; ++r) {
      const int yy = gy0 - 3 + r;
      if (yy < 0 || yy >= H) continue;
      float xin[PS + 6];
#pragma unroll
      for (int i = 0; i < PS + 6; ++i) {
        const int xx = gx0 - 3 + i;
        float val = 0.f;
        if (xx >= 0 && xx < H) {
          const int64_t src = ppeg_src((int64_t)yy * H + xx, N, wrapN);
          if (src >= 0) val = x[src * C + c];
        }
        xin[i] = val;
      }
#pragma unroll
      for (int jy = 0; jy < PSY; ++jy) {
        const int ddy = r - 3 - jy;
        if (ddy < -3 || ddy > 3) continue;
#pragma unroll
        for (int j = 0; j < PS; ++j)
#pragma unroll
          for (int dx = -3; dx <= 3; ++dx) acc[(ddy + 3) * 7 + dx + 3] = fmaf(g[jy][j], xin[j + dx + 3], acc[(ddy + 3) * 7 + dx + 3]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 49; ++i) part[((int64_t)blockIdx.x * 49 + i) * C + c] = acc[i];
  part_b[(int64_t)blockIdx.x * C + c] = ab;
}

// out[t, c] = v[t*ldv + c] * a[(c / dh) * lda + t]    (scoring.py:25: v * attn per head, heads interleaved as (h d))
__global__ void scale_heads_kernel(const float* __restrict__ v, int64_t ldv, const float* __restrict__ a, int64_t lda, int dh,
                                   int64_t T, int C, float* __restrict__ out) {
  for (int64_t t = blockIdx.x; t < T; t += gridDim.x)
    for (int c = threadIdx.x; c < C; c += blockDim.x) out[t * C + c] = v[t * ldv + c] * a[(int64_t)(c / dh) * lda + t];
}

// out[n, :] = x[n, :] + 2-d sin-cos embedding of the patch coordinate (px, py) (emb_position.py:5-83, SINCOS): four blocks of C/4 columns,
// sin(px w_k), cos(px w_k), sin(py w_k), cos(py w_k), w_k = 10000^(-k / (C/4)).  (The reference builds the H x W table and gathers row
// py * W + px: the same value.)
__global__ void sincos_add_kernel(const float* __restrict__ x, const int64_t* __restrict__ pos, int64_t N, int C, float* __restrict__ out) {
  const int Q = C / 4;
  for (int64_t n = blockIdx.x; n < N; n += gridDim.x) {
    const float px = (float)pos[2 * n], py = (float)pos[2 * n + 1];
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const int blk = c / Q, k = c % Q;
      const float omega = 1.f / powf(10000.f, (float)k / (float)Q);
      const float a = (blk < 2 ? px : py) * omega;
      out[n * C + c] = x[n * C + c] + ((blk & 1) ? cosf(a) : sinf(a));
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// BatchNorm1d over the M instances of ONE bag (mil_norm='bn': abmil.py:167-169,206-225, transmil.py:79-81,112-115): per column c,
//   training:  mean_c, var_c (biased) over the rows;  y = (x - mean) rstd w + b;     eval: the running statistics instead.
// Two launches each way: column statistics as per-row-chunk partials (thread = column: coalesced rows) + fixed-order reduction,
// then one element-wise pass.
// ---------------------------------------------------------------------------------------------------------------------------
// part[blk][0][c] = sum_m a[m,c] (b ? b[m,c]-weighted: sum a*xhat) ... two statistics per column:
//   MODE 0: s0 = sum x,  s1 = sum x^2                       (forward)
//   MODE 1: s0 = sum dy, s1 = sum dy * xhat, xhat = (x - mean) rstd      (backward)
template <int MODE>
__global__ __launch_bounds__(AT) void bn_stats_kernel(const float* __restrict__ a, const float* __restrict__ x, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, int64_t M, int C, int64_t chunk, float* __restrict__ part) {
  const int c = blockIdx.y * AT + threadIdx.x;
  if (c >= C) return;
  const int64_t m0 = (int64_t)blockIdx.x * chunk, m1 = m0 + chunk < M ? m0 + chunk : M;
  float s0 = 0.f, s1 = 0.f;
  const float mu = MODE ? mean[c] : 0.f, rs = MODE ? rstd[c] : 0.f;
  for (int64_t m = m0; m < m1; ++m) {
    const float v = a[m * C + c];
    s0 += v;
    s1 += MODE ? v * (x[m * C + c] - mu) * rs : v * v;
  }
  part[((int64_t)blockIdx.x * 2) * C + c] = s0;
  part[((int64_t)blockIdx.x * 2 + 1) * C + c] = s1;
}
// stats[2][C] = (sum, sumsq) -> mean, rstd (biased variance, eps); var_out = biased variance (for the running statistics)
__global__ void bn_finish_kernel(const float* __restrict__ stats, int64_t M, int C, float eps, float* __restrict__ mean, float* __restrict__ rstd,
                                 float* __restrict__ var_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float mu = stats[c] / (float)M;
  const float var = fmaxf(stats[C + c] / (float)M - mu * mu, 0.f);
  mean[c] = mu;
  rstd[c] = rsqrtf(var + eps);
  var_out[c] = var;
}
// y[m,c] = p[m,c] * A_c + q[m,c] * B_c + D_c   (q may be null)
__global__ void bn_apply_kernel(const float* __restrict__ p, const float* __restrict__ q, const float* __restrict__ A, const float* __restrict__ B,
                                const float* __restrict__ D, int64_t M, int C, float* __restrict__ y) {
  for (int64_t m = blockIdx.x; m < M; m += gridDim.x)
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float v = p[m * C + c] * A[c] + D[c];
      if (q) v += q[m * C + c] * B[c];
      y[m * C + c] = v;
    }
}
// coefficient vectors.  forward: A = rstd w, D = b - mean A.   backward (xhat = (x - mean) rstd, db = s0, dw = s1):
//   dx = w rstd (dy - db/M - xhat dw/M) = dy * (w rstd) + x * (-w rstd^2 dw / M) + (w rstd (mean rstd dw - db) / M)
__global__ void bn_coef_kernel(int bwd, const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ mean,
                               const float* __restrict__ rstd, const float* __restrict__ stats, int64_t M, int C, float* __restrict__ A,
                               float* __restrict__ B, float* __restrict__ D) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float wr = w[c] * rstd[c];
  if (!bwd) { A[c] = wr; B[c] = 0.f; D[c] = (b ? b[c] : 0.f) - mean[c] * wr; return; }
  const float db = stats[c], dw = stats[C + c], inv = 1.f / (float)M;
  A[c] = wr;
  B[c] = -wr * rstd[c] * dw * inv;
  D[c] = wr * (mean[c] * rstd[c] * dw - db) * inv;
}

}  // namespace mhimx

using namespace mhimx;

static inline unsigned grid1d(int64_t n, int per, int cap) {
  int64_t g = cdiv(n, per);
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}

extern "C" int mhimx_softmax_rows(void* stream, const float* x, float* y, int64_t R, int64_t L, float alpha) {
  MHIMX_CHECK_ARG(x && y && R > 0 && L > 0, "softmax_rows: bad args");
  if (L <= 1024)
    hipLaunchKernelGGL(softmax_rows_fwd_short_kernel, dim3(grid1d(R, 4, 65535)), dim3(AT), 0, (hipStream_t)stream, x, y, R, (int)L, alpha);
  else if (L % 4 == 0 && aligned16(x) && aligned16(y))
    hipLaunchKernelGGL(softmax_rows_fwd_vec_kernel, dim3(grid1d(R, 1, 65535)), dim3(AT), 0, (hipStream_t)stream, x, y, R, L, alpha);
  else
    hipLaunchKernelGGL(softmax_rows_fwd_kernel, dim3(grid1d(R, 1, 65535)), dim3(AT), 0, (hipStream_t)stream, x, y, R, L, alpha);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
extern "C" int mhimx_softmax_rows_bwd(void* stream, const float* y, const float* dy, float* dx, int64_t R, int64_t L, float alpha) {
  MHIMX_CHECK_ARG(y && dy && dx && R > 0 && L > 0, "softmax_rows_bwd: bad args");
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(grid1d(R, 1, 65535)), dim3(AT), 0, (hipStream_t)stream, y, dy, dx, R, L, alpha);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
extern "C" int mhimx_softmax_cols(void* stream, const float* x, float* y, int64_t M, int64_t C, float alpha) {
  MHIMX_CHECK_ARG(x && y && M > 0 && C > 0 && C <= 64, "softmax_cols: bad args");
  hipLaunchKernelGGL(softmax_cols_fwd_kernel, dim3((unsigned)C), dim3(1024), 0, (hipStream_t)stream, x, y, M, (int)C, alpha);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
extern "C" int mhimx_softmax_cols_bwd(void* stream, const float* y, const float* dy, float* dx, int64_t M, int64_t C, float alpha) {
  MHIMX_CHECK_ARG(y && dy && dx && M > 0 && C > 0 && C <= 64, "softmax_cols_bwd: bad args");
  hipLaunchKernelGGL(softmax_cols_bwd_kernel, dim3((unsigned)C), dim3(1024), 0, (hipStream_t)stream, y, dy, dx, M, (int)C, alpha);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
extern "C" int mhimx_landmark_mean(void* stream, const float* x, int64_t ldx, int64_t T, int64_t l, int64_t C, float* out) {
  MHIMX_CHECK_ARG(x && out && l > 0 && T % l == 0 && C > 0, "landmark_mean: T must be a multiple of l");
  if (C % 256 == 0 && ldx % 4 == 0 && aligned16(x) && aligned16(out))
    hipLaunchKernelGGL(landmark_fwd_vec_kernel, dim3((unsigned)(T / l), (unsigned)(C / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, (int)l, (int)C, out);
  else
    hipLaunchKernelGGL(landmark_fwd_kernel, dim3((unsigned)(T / l)), dim3(AT), 0, (hipStream_t)stream, x, ldx, (int)l, (int)C, out);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
extern "C" int mhimx_landmark_mean_bwd(void* stream, const float* dout, int64_t T, int64_t l, int64_t C, float* dx, int64_t ldx,
                                       int32_t accumulate) {
  MHIMX_CHECK_ARG(dout && dx && l > 0 && T % l == 0, "landmark_mean_bwd: bad args");
  if (C % 4 == 0 && ldx % 4 == 0 && aligned16(dout) && aligned16(dx))
    hipLaunchKernelGGL(landmark_bwd_vec_kernel, dim3(grid1d(T, 1, 16384)), dim3(256), 0, (hipStream_t)stream, dout, (int)l, (int)(C / 4), dx, ldx, T, accumulate);
  else
    hipLaunchKernelGGL(landmark_bwd_kernel, dim3(grid1d(T, 1, 8192)), dim3(AT), 0, (hipStream_t)stream, dout, (int)l, (int)C, dx, ldx, T, accumulate);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
extern "C" int mhimx_affine_ident(void* stream, const float* x, float* y, int64_t B, int64_t n, float a, float b) {
  MHIMX_CHECK_ARG(x && y && B > 0 && n > 0, "affine_ident: bad args");
  hipLaunchKernelGGL(affine_ident_kernel, dim3(grid1d(B * n * n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, x, y, B * n * n, (int)n, a, b);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
extern "C" int mhimx_axpby(void* stream, const float* x, float* y, int64_t n, float alpha, float beta) {
  MHIMX_CHECK_ARG(x && y && n >= 0, "axpby: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(axpby_kernel, dim3(grid1d(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, y, n, alpha, beta);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
// ws: (2*B*n + 4 + 1024) floats.  stats (4 floats) is kept by the caller for the backward.
extern "C" int mhimx_pinv_init(void* stream, const float* a, int64_t B, int64_t n, float* z, float* stats, float* ws) {
  MHIMX_CHECK_ARG(a && z && stats && ws && n % 32 == 0, "pinv_init: n must be a multiple of 32");
  float* rowsum = ws;
  float* colsum = ws + B * n;
  hipLaunchKernelGGL(pinv_sums_kernel, dim3((unsigned)(2 * B * n)), dim3(AT), 0, (hipStream_t)stream, a, (int)B, (int)n, rowsum, colsum);
  MHIMX_LAUNCH_CHECK();
  hipLaunchKernelGGL(pinv_argmax_kernel, dim3(1), dim3(AT), 0, (hipStream_t)stream, rowsum, colsum, (int)(B * n), stats);
  MHIMX_LAUNCH_CHECK();
  hipLaunchKernelGGL(pinv_init_kernel, dim3((unsigned)(n / 32), (unsigned)(n / 32), (unsigned)B), dim3(32, 8), 0, (hipStream_t)stream, a, stats, (int)B, (int)n, z);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
extern "C" int mhimx_pinv_init_bwd(void* stream, const float* dz, const float* z0, const float* stats, int64_t B, int64_t n, float* da,
                                   float* ws) {
  MHIMX_CHECK_ARG(dz && z0 && stats && da && ws && n % 32 == 0, "pinv_init_bwd: bad args");
  const int npart = 256;
  hipLaunchKernelGGL(dot_total_kernel, dim3(npart), dim3(AT), 0, (hipStream_t)stream, dz, z0, B * n * n, ws);
  MHIMX_LAUNCH_CHECK();
  hipLaunchKernelGGL(pinv_init_bwd_kernel, dim3((unsigned)(n / 32), (unsigned)(n / 32), (unsigned)B), dim3(32, 8), 0, (hipStream_t)stream, dz, z0, stats,
                     ws, npart, (int)B, (int)n, da);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
extern "C" int mhimx_resconv(void* stream, const float* v, int64_t ldv, const float* w, int64_t KS, int64_t dh, int64_t T, int64_t C,
                             float* out, int64_t ldo, int32_t accumulate, int32_t flip) {
  MHIMX_CHECK_ARG(v && w && out && KS % 2 == 1 && C % dh == 0, "resconv: bad args");
  static const bool strips = getenv("MHIMX_RESCONV_STRIPS") != nullptr;  // (experiments: the register-sliding strips)
  if (KS == 33 && !strips && C % RT_CH == 0 && ldv % 4 == 0 && ((uintptr_t)v & 15) == 0) {
    hipLaunchKernelGGL(resconv_tile_kernel<33>, dim3((unsigned)cdiv(T, RT_ROWS), (unsigned)(C / RT_CH)), dim3(256), 0, (hipStream_t)stream, v, ldv,
                       w, (int)dh, T, (int)C, out, ldo, accumulate, flip);
    MHIMX_LAUNCH_CHECK();
    return 0;
  }
  if (KS == 33) {        // the reference's residual_conv_kernel (nystrom_attention.py:43): register-sliding strips
    hipLaunchKernelGGL(resconv_strip_kernel<33>, dim3((unsigned)cdiv(T, RS), (unsigned)cdiv(C, AT)), dim3(AT), 0, (hipStream_t)stream, v, ldv,
                       w, (int)dh, T, (int)C, out, ldo, accumulate, flip);
    MHIMX_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(resconv_fwd_kernel, dim3(grid1d(T, 1, 16384)), dim3(AT), 0, (hipStream_t)stream, v, ldv, w, (int)KS, (int)dh, T, (int)C, out,
                     ldo, accumulate, flip);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
// fixed-order reduction of partial rows: out[j] = sum_b part[b*W + j]
__global__ __launch_bounds__(1024) void attn_reduce_kernel(const float* __restrict__ part, int G, int64_t W, float* __restrict__ out) {
  __shared__ float red[32][33];
  const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
  for (int64_t j0 = (int64_t)blockIdx.x * 32; j0 < W; j0 += (int64_t)gridDim.x * 32) {
    const int64_t j = j0 + c;
    float acc = 0.f;
    if (j < W)
      for (int b = rg; b < G; b += 32) acc += part[(int64_t)b * W + j];
    red[rg][c] = acc;
    __syncthreads();
    if (rg == 0 && j < W) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < 32; ++q) v += red[q][c];
      out[j] = v;
    }
    __syncthreads();
  }
}
static int attn_reduce(hipStream_t st, const float* part, int G, int64_t W, float* out) {
  hipLaunchKernelGGL(attn_reduce_kernel, dim3(grid1d(W, 32, 4096)), dim3(1024), 0, st, part, G, W, out);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

// dw[h, tau] of the residual convolution.  ws: mhimx_resconv_dw_ws_floats(...) floats.
constexpr int64_t RC_CHUNK = 64;        // tokens per workgroup of the dw partials (256: 18 serial load rounds on 1.5 workgroups per CU)
extern "C" int64_t mhimx_resconv_dw_ws_floats(int64_t T, int64_t C, int64_t dh, int64_t KS) { return cdiv(T, RC_CHUNK) * (C / dh) * KS; }
extern "C" int mhimx_resconv_dw(void* stream, const float* dout, int64_t ldo, const float* v, int64_t ldv, int64_t KS, int64_t dh,
                                int64_t T, int64_t C, float* dw, float* ws) {
  MHIMX_CHECK_ARG(dout && v && dw && ws && dh == 64 && C % dh == 0, "resconv_dw: dim_head must be 64");
  const int nblk = (int)cdiv(T, RC_CHUNK);
  const int W = (int)((C / dh) * KS);
  if (KS == 33 && C % AT == 0) {
    hipLaunchKernelGGL(resconv_dw_strip_kernel<33>, dim3((unsigned)nblk, (unsigned)(C / AT)), dim3(AT), 0, (hipStream_t)stream, dout, ldo, v,
                       ldv, (int)dh, T, (int)C, RC_CHUNK, ws);
    MHIMX_LAUNCH_CHECK();
    return attn_reduce((hipStream_t)stream, ws, nblk, W, dw);
  }
  hipLaunchKernelGGL(resconv_dw_kernel, dim3(nblk), dim3(AT), (size_t)W * 4, (hipStream_t)stream, dout, ldo, v, ldv, (int)KS, (int)dh, T, (int)C,
                     RC_CHUNK, ws);
  MHIMX_LAUNCH_CHECK();
  return attn_reduce((hipStream_t)stream, ws, nblk, W, dw);
}

// wc[c, 7x7] = w7 + centre-padded w5 + centre-padded w3 + identity ; bc = b7 + b5 + b3   (emb_position.py:115)
__global__ void ppeg_combine_kernel(const float* __restrict__ w7, const float* __restrict__ w5, const float* __restrict__ w3,
                                    const float* __restrict__ b7, const float* __restrict__ b5, const float* __restrict__ b3, int C,
                                    float* __restrict__ wc, float* __restrict__ bc) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < C * 49; i += gridDim.x * blockDim.x) {
    const int c = i / 49, t = i % 49, y = t / 7, x = t % 7;
    float v = w7[i];
    if (y >= 1 && y <= 5 && x >= 1 && x <= 5) v += w5[c * 25 + (y - 1) * 5 + (x - 1)];
    if (y >= 2 && y <= 4 && x >= 2 && x <= 4) v += w3[c * 9 + (y - 2) * 3 + (x - 2)];
    if (t == 24) v += 1.f;
    wc[i] = v;
    if (t == 0) bc[c] = b7[c] + b5[c] + b3[c];
  }
}
extern "C" int mhimx_ppeg_combine(void* stream, const float* w7, const float* w5, const float* w3, const float* b7, const float* b5,
                                  const float* b3, int64_t C, float* wc, float* bc) {
  MHIMX_CHECK_ARG(w7 && w5 && w3 && b7 && b5 && b3 && wc && bc, "ppeg_combine: null args");
  hipLaunchKernelGGL(ppeg_combine_kernel, dim3(grid1d(C * 49, 256, 1024)), dim3(256), 0, (hipStream_t)stream, w7, w5, w3, b7, b5, b3, (int)C, wc, bc);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
// grid == 0: the rule of emb_position.PPEG (MHIM's encoder): side ceil(sqrt(N)), the first side^2 - N tokens appended again, zero-padded
// to 7 x 7 below that.  grid > 0: an explicit grid x grid layout (modules/transmil.PPEG.forward(x, H, W): the caller has padded the tokens).
static void ppeg_geom(int64_t N, int64_t grid, int* H, int64_t* wrapN) {
  if (grid > 0) {
    *H = (int)grid;
    *wrapN = grid * grid;
    return;
  }
  int h0 = (int)ceil(sqrt((double)N));
  while ((int64_t)h0 * h0 < N) ++h0;
  while (h0 > 1 && (int64_t)(h0 - 1) * (h0 - 1) >= N) --h0;
  *wrapN = (int64_t)h0 * h0;
  *H = h0 < 7 ? 7 : h0;
}
extern "C" int mhimx_ppeg_fwd(void* stream, const float* x, int64_t N, int64_t C, const float* wc, const float* bc, float* y, int64_t grid) {
  MHIMX_CHECK_ARG(x && wc && bc && y && N > 0, "ppeg_fwd: bad args");
  MHIMX_CHECK_ARG(grid == 0 || (grid * grid >= N && grid * grid - N <= N && grid <= 32768), "ppeg_fwd: a %lld x %lld grid does not hold %lld tokens",
                  (long long)grid, (long long)grid, (long long)N);
  int H; int64_t wrapN;
  ppeg_geom(N, grid, &H, &wrapN);
  const int nsx = (int)cdiv(H, PS);
  hipLaunchKernelGGL(ppeg_strip_kernel<0>, dim3((unsigned)(nsx * cdiv(H, PSY)), (unsigned)cdiv(C, AT)), dim3(AT), 0, (hipStream_t)stream, x, N, (int)C, H,
                     wrapN, wc, bc, y, nsx);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
extern "C" int64_t mhimx_ppeg_bwd_ws_floats(int64_t N, int64_t C) { return cdiv(N, 64) * C * 50 + 49 * C; }
extern "C" int mhimx_ppeg_bwd(void* stream, const float* dy, const float* x, int64_t N, int64_t C, const float* wc, float* dx, float* dwc,
                              float* dbc, float* ws, int64_t grid) {
  MHIMX_CHECK_ARG(dy && x && wc && dx && dwc && dbc && ws, "ppeg_bwd: null args");
  MHIMX_CHECK_ARG(grid == 0 || (grid * grid >= N && grid * grid - N <= N && grid <= 32768), "ppeg_bwd: bad grid");
  int H; int64_t wrapN;
  ppeg_geom(N, grid, &H, &wrapN);
  hipStream_t st = (hipStream_t)stream;
  const int nsx = (int)cdiv(H, PS);
  hipLaunchKernelGGL(ppeg_strip_kernel<1>, dim3((unsigned)(nsx * cdiv(H, PSY)), (unsigned)cdiv(C, AT)), dim3(AT), 0, st, dy, N, (int)C, H, wrapN, wc,
                     (const float*)nullptr, dx, nsx);
  MHIMX_LAUNCH_CHECK();
  if (wrapN > N) {
    hipLaunchKernelGGL(ppeg_bwd_dx_kernel, dim3(grid1d(wrapN - N, 1, 32768)), dim3(AT), 0, st, dy, N, (int)C, H, wrapN, wc, dx, N, wrapN, 1);
    MHIMX_LAUNCH_CHECK();
  }
  const int nblk = (int)cdiv(N, 64);          // strips of one block are a latency chain: many short blocks (1368 -> 5472 waves at N = 43 776)
  float* part = ws;
  float* part_b = ws + (int64_t)nblk * C * 49;
  float* dwc_t = part_b + (int64_t)nblk * C;                    // [49, C], transposed into dwc [C, 49] at the end
  const int spb = (int)cdiv((int64_t)cdiv(H, PSY) * nsx, nblk);        // patches per block
  hipLaunchKernelGGL(ppeg_dw_strip_kernel, dim3((unsigned)nblk, (unsigned)cdiv(C, AT)), dim3(AT), 0, st, dy, x, N, (int)C, H, wrapN, nsx, spb,
                     part, part_b);
  MHIMX_LAUNCH_CHECK();
  if (int r = attn_reduce(st, part, nblk, C * 49, dwc_t)) return r;
  if (int r = transpose(st, dwc_t, dwc, 49, C)) return r;
  return attn_reduce(st, part_b, nblk, C, dbc);
}
// ------------------------------------------------------------------------------------------------
// PPEG on a BAND of the token grid (sequence-parallel encoder, nystrom_sharded.py): a rank holds the grid cells [cell0, cell0 + ncell)
// (whole grid rows: its own tokens' rows and three halo rows either side; the cells that wrap - the first side^2 - N tokens appended again,
// emb_position.py:100-103 - filled in by the exchange, cells past them zero) and produces the cells [out0, out1).  The patch schedule of
// ppeg_strip_kernel / ppeg_dw_strip_kernel with explicit buffers: the band IS the padded grid, no index wraps.
// FLIP = 0: y[cell - out0] = bc + sum_tap wc[tap] xb[cell + tap];  FLIP = 1: dx[cell - out0] = sum_tap wc[flipped tap] dyb[cell + tap]
// ------------------------------------------------------------------------------------------------
struct PpegBand { int H; int row0, nrows; int64_t cell0, ncell, out0, out1; };
template <int FLIP>
__global__ __launch_bounds__(AT) void ppeg_band_strip_kernel(const float* __restrict__ in, PpegBand b, int C, const float* __restrict__ wc,
                                                             const float* __restrict__ bc, float* __restrict__ out, int nsx, int prow0) {
  const int c = blockIdx.y * AT + threadIdx.x;
  if (c >= C) return;
  const int H = b.H;
  const int gy0 = prow0 + (blockIdx.x / nsx) * PSY, gx0 = (blockIdx.x % nsx) * PS;       // (global grid coordinates)
  float w[49];
#pragma unroll
  for (int i = 0; i < 49; ++i) w[i] = wc[c * 49 + i];
  float acc[PSY][PS];
  const float b0 = FLIP ? 0.f : bc[c];
#pragma unroll
  for (int jy = 0; jy < PSY; ++jy)
#pragma unroll
    for (int j = 0; j < PS; ++j) acc[jy][j] = b0;
#pragma unroll
  for (int r = 0; r < PSY + 6; ++r) {
    const int yy = gy0 - 3 + r;
    if (yy < 0 || yy >= H) continue;
    float xin[PS + 6];
#pragma unroll
    for (int i = 0; i < PS + 6; ++i) {
      const int xx = gx0 - 3 + i;
      float val = 0.f;
      if (xx >= 0 && xx < H) {
        const int64_t cell = (int64_t)yy * H + xx - b.cell0;
        if (cell >= 0 && cell < b.ncell) val = in[cell * C + c];
      }
      xin[i] = val;
    }
#pragma unroll
    for (int jy = 0; jy < PSY; ++jy) {
      const int dy = r - 3 - jy;
      if (dy < -3 || dy > 3) continue;
#pragma unroll
      for (int j = 0; j < PS; ++j)
#pragma unroll
        for (int dx = -3; dx <= 3; ++dx) {
          const int tap = FLIP ? (3 - dy) * 7 + (3 - dx) : (dy + 3) * 7 + (dx + 3);
          acc[jy][j] = fmaf(w[tap], xin[j + dx + 3], acc[jy][j]);
        }
    }
  }
#pragma unroll
  for (int jy = 0; jy < PSY; ++jy)
#pragma unroll
    for (int j = 0; j < PS; ++j) {
      const int64_t cell = (int64_t)(gy0 + jy) * H + gx0 + j;
      if (gy0 + jy < H && gx0 + j < H && cell >= b.out0 && cell < b.out1) out[(cell - b.out0) * C + c] = acc[jy][j];
    }
}
// dwc / dbc partials over the produced cells [out0, out1): g = dyb at the cell, inputs from the band
__global__ __launch_bounds__(AT) void ppeg_band_dw_kernel(const float* __restrict__ dyb, const float* __restrict__ xb, PpegBand b, int C, int nsx,
                                                          int prow0, int npatch, int spb, float* __restrict__ part, float* __restrict__ part_b) {
  const int c = blockIdx.y * AT + threadIdx.x;
  if (c >= C) return;
  const int H = b.H;
  float acc[49];
#pragma unroll
  for (int i = 0; i < 49; ++i) acc[i] = 0.f;
  float ab = 0.f;
  for (int s = blockIdx.x * spb; s < (int)(blockIdx.x + 1) * spb && s < npatch; ++s) {
    const int gy0 = prow0 + (s / nsx) * PSY, gx0 = (s % nsx) * PS;
    float g[PSY][PS];
#pragma unroll
    for (int jy = 0; jy < PSY; ++jy)
#pragma unroll
      for (int j = 0; j < PS; ++j) {
        const int64_t cell = (int64_t)(gy0 + jy) * H + gx0 + j;
        g[jy][j] = (gy0 + jy < H && gx0 + j < H && cell >= b.out0 && cell < b.out1) ? dyb[(cell - b.cell0) * C + c] : 0.f;
        ab += g[jy][j];
      }
#pragma unroll
    for (int r = 0; r < PSY + 6; ++r) {
      const int yy = gy0 - 3 + r;
      if (yy < 0 || yy >= H) continue;
      float xin[PS + 6];
#pragma unroll
      for (int i = 0; i < PS + 6; ++i) {
        const int xx = gx0 - 3 + i;
        float val = 0.f;
        if (xx >= 0 && xx < H) {
          const int64_t cell = (int64_t)yy * H + xx - b.cell0;
          if (cell >= 0 && cell < b.ncell) val = xb[cell * C + c];
        }
        xin[i] = val;
      }
#pragma unroll
      for (int jy = 0; jy < PSY; ++jy) {
        const int ddy = r - 3 - jy;
        if (ddy < -3 || ddy > 3) continue;
#pragma unroll
        for (int j = 0; j < PS; ++j)
#pragma unroll
          for (int dx = -3; dx <= 3; ++dx) acc[(ddy + 3) * 7 + dx + 3] = fmaf(g[jy][j], xin[j + dx + 3], acc[(ddy + 3) * 7 + dx + 3]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 49; ++i) part[((int64_t)blockIdx.x * 49 + i) * C + c] = acc[i];
  part_b[(int64_t)blockIdx.x * C + c] = ab;
}

extern "C" int64_t mhimx_ppeg_side(int64_t N, int64_t* wrapN) {
  int H; int64_t w;
  ppeg_geom(N, 0, &H, &w);
  if (wrapN) *wrapN = w;
  return H;
}
static int ppeg_band_check(const mhimx_ppeg_band* b, PpegBand* o) {
  MHIMX_CHECK_ARG(b && b->H >= 7 && b->H <= 32768 && b->cell0 >= 0 && b->cell0 % b->H == 0 && b->ncell > 0 && b->out0 >= b->cell0 &&
                      b->out1 > b->out0 && b->out1 <= b->cell0 + b->ncell && b->out1 <= b->H * b->H,
                  "ppeg_band: the band holds whole grid rows from cell0 and contains the cells it produces (side >= 7)");
  o->H = (int)b->H; o->cell0 = b->cell0; o->ncell = b->ncell; o->out0 = b->out0; o->out1 = b->out1;
  o->row0 = (int)(b->out0 / b->H);                                  // patch rows start at the first produced cell's row
  o->nrows = (int)((b->out1 - 1) / b->H) - o->row0 + 1;
  return 0;
}
extern "C" int mhimx_ppeg_band_fwd(void* stream, const float* xb, const mhimx_ppeg_band* band, int64_t C, const float* wc, const float* bc, float* y) {
  MHIMX_CHECK_ARG(xb && wc && bc && y, "ppeg_band_fwd: null args");
  PpegBand b;
  if (int r = ppeg_band_check(band, &b)) return r;
  const int nsx = (int)cdiv(b.H, PS);
  hipLaunchKernelGGL(ppeg_band_strip_kernel<0>, dim3((unsigned)(nsx * cdiv(b.nrows, PSY)), (unsigned)cdiv(C, AT)), dim3(AT), 0, (hipStream_t)stream, xb, b,
                     (int)C, wc, bc, y, nsx, b.row0);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
extern "C" int64_t mhimx_ppeg_band_bwd_ws_floats(int64_t n_out, int64_t C) { return cdiv(n_out, 64) * C * 50 + 49 * C; }
extern "C" int mhimx_ppeg_band_bwd(void* stream, const float* dyb, const float* xb, const mhimx_ppeg_band* band, int64_t n_dw, int64_t C, const float* wc,
                                   float* dx, float* dwc, float* dbc, float* ws) {
  MHIMX_CHECK_ARG(dyb && xb && wc && dx && dwc && dbc && ws, "ppeg_band_bwd: null args");
  PpegBand b;
  if (int r = ppeg_band_check(band, &b)) return r;
  MHIMX_CHECK_ARG(n_dw >= 1 && b.out0 + n_dw <= b.out1, "ppeg_band_bwd: the weight gradient runs over the first n_dw produced cells");
  hipStream_t st = (hipStream_t)stream;
  const int nsx = (int)cdiv(b.H, PS);
  hipLaunchKernelGGL(ppeg_band_strip_kernel<1>, dim3((unsigned)(nsx * cdiv(b.nrows, PSY)), (unsigned)cdiv(C, AT)), dim3(AT), 0, st, dyb, b, (int)C, wc,
                     (const float*)nullptr, dx, nsx, b.row0);
  MHIMX_LAUNCH_CHECK();
  PpegBand bw = b;                                                  // weight gradient: over the cells that ARE outputs of the forward
  bw.out1 = b.out0 + n_dw;
  bw.nrows = (int)((bw.out1 - 1) / b.H) - bw.row0 + 1;
  const int nblk = (int)cdiv(b.out1 - b.out0, 64);
  float* part = ws;
  float* part_b = ws + (int64_t)nblk * C * 49;
  float* dwc_t = part_b + (int64_t)nblk * C;
  const int npatch = (int)cdiv(bw.nrows, PSY) * nsx;
  const int spb = (int)cdiv(npatch, nblk);
  hipLaunchKernelGGL(ppeg_band_dw_kernel, dim3((unsigned)nblk, (unsigned)cdiv(C, AT)), dim3(AT), 0, st, dyb, xb, bw, (int)C, nsx, bw.row0, npatch, spb,
                     part, part_b);
  MHIMX_LAUNCH_CHECK();
  if (int r = attn_reduce(st, part, nblk, C * 49, dwc_t)) return r;
  if (int r = transpose(st, dwc_t, dwc, 49, C)) return r;
  return attn_reduce(st, part_b, nblk, C, dbc);
}
extern "C" int mhimx_scale_heads(void* stream, const float* v, int64_t ldv, const float* a, int64_t lda, int64_t dh, int64_t T, int64_t C,
                                 float* out) {
  MHIMX_CHECK_ARG(v && a && out && C % dh == 0, "scale_heads: bad args");
  hipLaunchKernelGGL(scale_heads_kernel, dim3(grid1d(T, 1, 16384)), dim3(AT), 0, (hipStream_t)stream, v, ldv, a, lda, (int)dh, T, (int)C, out);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

extern "C" int mhimx_sincos_add(void* stream, const float* x, const int64_t* pos_xy, int64_t N, int64_t C, float* out) {
  MHIMX_CHECK_ARG(x && pos_xy && out && C % 4 == 0 && N >= 0, "sincos_add: bad args");
  if (N == 0) return 0;
  hipLaunchKernelGGL(sincos_add_kernel, dim3(grid1d(N, 1, 16384)), dim3(AT), 0, (hipStream_t)stream, x, pos_xy, N, (int)C, out);
  MHIMX_LAUNCH_CHECK();
  return 0;
}

// ws: mhimx_bn_ws_floats(M, C) floats
extern "C" int64_t mhimx_bn_ws_floats(int64_t M, int64_t C) { return (cdiv(M, 64) * 2 + 8) * C; }
static int bn_stats(hipStream_t st, int mode, const float* a, const float* x, const float* mean, const float* rstd, int64_t M, int64_t C, float* ws,
                    float* stats) {
  const int64_t chunk = 64;
  const int nblk = (int)cdiv(M, chunk);
  if (mode == 0) hipLaunchKernelGGL(bn_stats_kernel<0>, dim3((unsigned)nblk, (unsigned)cdiv(C, AT)), dim3(AT), 0, st, a, x, mean, rstd, M, (int)C, chunk, ws);
  else hipLaunchKernelGGL(bn_stats_kernel<1>, dim3((unsigned)nblk, (unsigned)cdiv(C, AT)), dim3(AT), 0, st, a, x, mean, rstd, M, (int)C, chunk, ws);
  MHIMX_LAUNCH_CHECK();
  // partial rows are [blk][2][C]: even rows = s0, odd rows = s1 -> reduce as width 2C over nblk rows
  return attn_reduce(st, ws, nblk, 2 * C, stats);
}
extern "C" int mhimx_bn_fwd(void* stream, const float* x, int64_t M, int64_t C, const float* w, const float* b, float eps, int32_t training,
                            float* mean /* in (eval: running mean) / out */, float* rstd /* out; eval: in = running VAR */, float* var_out,
                            float* y, float* ws) {
  MHIMX_CHECK_ARG(x && w && mean && rstd && y && ws && M >= 1 && C >= 1, "bn_fwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  float* stats = ws + cdiv(M, 64) * 2 * C;
  float* A = stats + 2 * C; float* B = A + C; float* D = B + C;
  const unsigned gc = (unsigned)cdiv(C, 256);
  if (training) {
    MHIMX_CHECK_ARG(var_out, "bn_fwd: training needs var_out");
    if (int r = bn_stats(st, 0, x, nullptr, nullptr, nullptr, M, C, ws, stats)) return r;
    hipLaunchKernelGGL(bn_finish_kernel, dim3(gc), dim3(256), 0, st, stats, M, (int)C, eps, mean, rstd, var_out);
  }
  hipLaunchKernelGGL(bn_coef_kernel, dim3(gc), dim3(256), 0, st, 0, w, b, mean, rstd, stats, M, (int)C, A, B, D);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid1d(M, 1, 16384)), dim3(AT), 0, st, x, (const float*)nullptr, A, B, D, M, (int)C, y);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
extern "C" int mhimx_bn_bwd(void* stream, const float* dy, const float* x, int64_t M, int64_t C, const float* w, const float* mean,
                            const float* rstd, int32_t training, float* dx /* may be NULL */, float* dw, float* db, float* ws) {
  MHIMX_CHECK_ARG(dy && x && w && mean && rstd && dw && db && ws && M >= 1, "bn_bwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  float* stats = ws + cdiv(M, 64) * 2 * C;
  float* A = stats + 2 * C; float* B = A + C; float* D = B + C;
  if (int r = bn_stats(st, 1, dy, x, mean, rstd, M, C, ws, stats)) return r;
  MHIMX_HIP(hipMemcpyAsync(db, stats, C * sizeof(float), hipMemcpyDeviceToDevice, st));
  MHIMX_HIP(hipMemcpyAsync(dw, stats + C, C * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (!dx) return 0;
  const unsigned gc = (unsigned)cdiv(C, 256);
  if (training) {
    hipLaunchKernelGGL(bn_coef_kernel, dim3(gc), dim3(256), 0, st, 1, w, (const float*)nullptr, mean, rstd, stats, M, (int)C, A, B, D);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid1d(M, 1, 16384)), dim3(AT), 0, st, dy, x, A, B, D, M, (int)C, dx);
  } else {                                                    // eval: the statistics are constants, dx = dy w rstd
    hipLaunchKernelGGL(bn_coef_kernel, dim3(gc), dim3(256), 0, st, 0, w, (const float*)nullptr, mean, rstd, stats, M, (int)C, A, B, D);
    hipMemsetAsync(D, 0, C * sizeof(float), st);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid1d(M, 1, 16384)), dim3(AT), 0, st, dy, (const float*)nullptr, A, B, D, M, (int)C, dx);
  }
  MHIMX_LAUNCH_CHECK();
  return 0;
}
