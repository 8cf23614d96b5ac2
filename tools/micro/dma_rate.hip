// microbenchmark: LDS-DMA (global_load_lds_dwordx4) ingest rate per CU vs in-flight depth and access pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// each WG: NW waves; a "chunk" = NW KiB (each wave moves 1 KiB per instruction, PER instructions per chunk per wave)
template <int DEPTH, int PER>
__global__ void dma_kernel(const float* __restrict__ src, int64_t wg_stride_floats, int iters, int pattern, int64_t pitch_floats,
                           float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const float* base = src + (int64_t)blockIdx.x * wg_stride_floats;
  const int chunk_bytes = nw * 1024 * PER;
  auto issue = [&](int it, int stage) {
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const float* g;
      const int slot = (wave * PER + p) * 64 + lane;          // 16-B slot index inside the chunk
      if (pattern == 0) {                                     // contiguous stream
        g = base + ((int64_t)it * chunk_bytes / 4) + slot * 4;
      } else {                                                // GEMM-like: 8 rows x 128 B per wave instruction, row pitch given
        const int row = slot >> 3, c = slot & 7;
        g = base + (int64_t)row * pitch_floats + (int64_t)it * 32 + c * 4;
      }
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(smem + stage * chunk_bytes + (wave * PER + p) * 1024), 16, 0, 0);
    }
  };
  for (int s = 0; s < DEPTH - 1; ++s) issue(s, s);
  for (int it = 0; it < iters; ++it) {
    if (it + DEPTH - 1 < iters) issue(it + DEPTH - 1, (it + DEPTH - 1) % DEPTH);
    // wait until only the (DEPTH-1) younger chunks are outstanding
    if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (it + DEPTH - 1 < iters) {
      if constexpr ((DEPTH - 1) * PER == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else if constexpr ((DEPTH - 1) * PER == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if constexpr ((DEPTH - 1) * PER == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if constexpr ((DEPTH - 1) * PER == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if constexpr ((DEPTH - 1) * PER == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if constexpr ((DEPTH - 1) * PER == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr ((DEPTH - 1) * PER == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if constexpr ((DEPTH - 1) * PER == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if constexpr ((DEPTH - 1) * PER == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (sink && tid == 0) sink[blockIdx.x] = *(float*)smem;
}

template <int DEPTH, int PER>
void run(const char* name, const float* d, int nwg, int nw, int iters, int pattern, int64_t wg_stride, int64_t pitch) {
  size_t smem = (size_t)DEPTH * nw * 1024 * PER;
  hipFuncSetAttribute((const void*)dma_kernel<DEPTH, PER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((dma_kernel<DEPTH, PER>), dim3(nwg), dim3(nw * 64), smem, 0, d, wg_stride, iters, pattern, pitch, (float*)nullptr);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((dma_kernel<DEPTH, PER>), dim3(nwg), dim3(nw * 64), smem, 0, d, wg_stride, iters, pattern, pitch, (float*)nullptr);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double bytes = (double)nwg * iters * nw * 1024 * PER;
  printf("%-44s nwg=%3d nw=%d depth=%d chunk=%3dKB lds=%3zuKB: %7.1f us  %6.2f TB/s  %6.1f GB/s/WG\n", name, nwg, nw, DEPTH, nw * PER,
         smem / 1024, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / nwg);
}

int main() {
  float* d; size_t n = (size_t)512 << 20; hipMalloc(&d, n); hipMemset(d, 0, n);
  // L2/MALL-resident: every WG re-reads the same 2 MB region (pattern 1: rows at 4 KiB pitch, k advancing)
  printf("--- all WGs read the SAME 128 rows x 4 KiB (L2-resident), GEMM-like pattern\n");
  run<1, 4>("same-region gemm-like", d, 256, 8, 32, 1, 0, 1024);
  run<2, 4>("same-region gemm-like", d, 256, 8, 32, 1, 0, 1024);
  run<3, 4>("same-region gemm-like", d, 256, 8, 32, 1, 0, 1024);
  run<4, 4>("same-region gemm-like", d, 256, 8, 32, 1, 0, 1024);
  run<2, 2>("same-region gemm-like 4 waves", d, 256, 4, 64, 1, 0, 1024);
  run<3, 2>("same-region gemm-like 4 waves", d, 256, 4, 64, 1, 0, 1024);
  printf("--- each WG streams its OWN 128 rows (HBM), GEMM-like pattern\n");
  run<1, 4>("own-rows gemm-like", d, 256, 8, 32, 1, 128 * 1024, 1024);
  run<2, 4>("own-rows gemm-like", d, 256, 8, 32, 1, 128 * 1024, 1024);
  run<3, 4>("own-rows gemm-like", d, 256, 8, 32, 1, 128 * 1024, 1024);
  run<4, 4>("own-rows gemm-like", d, 256, 8, 32, 1, 128 * 1024, 1024);
  printf("--- each WG streams its OWN contiguous 1 MiB (HBM)\n");
  run<1, 4>("own contiguous", d, 256, 8, 32, 0, 262144, 0);
  run<2, 4>("own contiguous", d, 256, 8, 32, 0, 262144, 0);
  run<4, 4>("own contiguous", d, 256, 8, 32, 0, 262144, 0);
  run<4, 4>("own contiguous 512 WGs", d, 512, 8, 32, 0, 262144, 0);
  return 0;
}
