// shard.hip — the three index kernels of the instance-sharded step (mhim_mil_amd/sharded.py): a shard owns bag
// rows [lo, lo + n); the student's row lists are replicated.  They replace torch nonzero / index_copy / index_select and the host
// read-back of the data-dependent local counts: every launch shape of the sharded step is fixed by (n, R, Lk, k).
#include "common.hpp"

namespace mhimx {

__global__ void shard_fill_kernel(uint8_t* __restrict__ excl, int64_t n, int64_t k_tokens, int tokens_live) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) excl[i] = 1;
  else if (i < n + k_tokens) excl[i] = tokens_live ? 0 : 1;
}
__global__ void shard_stay_kernel(const int64_t* __restrict__ stay, int64_t Lk, int64_t lo, int64_t n, uint8_t* __restrict__ excl) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= Lk) return;
  const int64_t r = stay[j] - lo;
  if (r >= 0 && r < n) excl[r] = 0;
}

// one wave per list entry, 16 bytes per lane
__global__ __launch_bounds__(256) void shard_gather_kernel(const float* __restrict__ H, int E4, const int64_t* __restrict__ rows, int64_t R,
                                                          int64_t lo, int64_t n, float* __restrict__ out) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= R) return;
  const int64_t r = rows[j] - lo;
  const bool own = r >= 0 && r < n;
  const f4* src = reinterpret_cast<const f4*>(H) + (own ? r : 0) * E4;
  f4* dst = reinterpret_cast<f4*>(out) + j * E4;
  for (int c = threadIdx.x & 63; c < E4; c += 64) dst[c] = own ? src[c] : f4{0.f, 0.f, 0.f, 0.f};
}
__global__ __launch_bounds__(256) void shard_scatter_kernel(const float* __restrict__ dX, int E4, const int64_t* __restrict__ rows, int64_t R,
                                                           int64_t lo, int64_t n, float* __restrict__ dH) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= R) return;
  const int64_t r = rows[j] - lo;
  if (r < 0 || r >= n) return;
  const f4* src = reinterpret_cast<const f4*>(dX) + j * E4;
  f4* dst = reinterpret_cast<f4*>(dH) + r * E4;
  for (int c = threadIdx.x & 63; c < E4; c += 64) dst[c] = src[c];
}

}  // namespace mhimx

using namespace mhimx;

extern "C" int mhimx_shard_flags(void* stream, const int64_t* rows_all, int64_t R, int64_t Lk, int64_t lo, int64_t n, int64_t k_tokens,
                                 int32_t tokens_live, uint8_t* excl) {
  MHIMX_CHECK_ARG(rows_all && excl && R >= 0 && Lk >= 0 && n >= 1 && k_tokens >= 0 && lo >= 0, "shard_flags: bad args");
  hipLaunchKernelGGL(shard_fill_kernel, dim3((unsigned)cdiv(n + k_tokens, 256)), dim3(256), 0, (hipStream_t)stream, excl, n, k_tokens, tokens_live);
  MHIMX_LAUNCH_CHECK();
  if (Lk > 0) {
    hipLaunchKernelGGL(shard_stay_kernel, dim3((unsigned)cdiv(Lk, 256)), dim3(256), 0, (hipStream_t)stream, rows_all + R, Lk, lo, n, excl);
    MHIMX_LAUNCH_CHECK();
  }
  return 0;
}
extern "C" int mhimx_shard_gather(void* stream, const float* H, int64_t E, const int64_t* rows, int64_t R, int64_t lo, int64_t n, float* out) {
  MHIMX_CHECK_ARG(H && rows && out && E > 0 && E % 4 == 0 && aligned16(H) && aligned16(out) && n >= 1 && R >= 0, "shard_gather: bad args");
  if (R == 0) return 0;
  hipLaunchKernelGGL(shard_gather_kernel, dim3((unsigned)cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, H, (int)(E / 4), rows, R, lo, n, out);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
extern "C" int mhimx_shard_scatter(void* stream, const float* dX, int64_t E, const int64_t* rows, int64_t R, int64_t lo, int64_t n, float* dH) {
  MHIMX_CHECK_ARG(dX && rows && dH && E > 0 && E % 4 == 0 && aligned16(dX) && aligned16(dH) && n >= 1 && R >= 0, "shard_scatter: bad args");
  if (R == 0) return 0;
  hipLaunchKernelGGL(shard_scatter_kernel, dim3((unsigned)cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, dX, (int)(E / 4), rows, R, lo, n, dH);
  MHIMX_LAUNCH_CHECK();
  return 0;
}
