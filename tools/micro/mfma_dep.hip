// mfma_dep.hip - cycles per v_mfma_f32_16x16x32_bf16 as a function of the number of independent accumulators in the round robin
// (1 = every MFMA waits for the previous one) and of the waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 24 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC>
void run(int waves_per_simd) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
  const int iters = 2000, threads = 256 * waves_per_simd;
  k<NACC><<<256, threads>>>(out, cyc, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<NACC><<<256, threads>>>(out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 24;
  printf("acc %2d waves/SIMD %d: %.1f shader cycles per MFMA per wave, wall %.3f ms -> %.1f TFLOP/s\n", NACC, waves_per_simd, c / n, ms,
         n * 4 * waves_per_simd * 256 * 16384.0 / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int w = 1; w <= 4; w *= 2) { run<1>(w); run<2>(w); run<3>(w); run<4>(w); run<8>(w); }
  return 0;
}
