// HBM stream-copy variants on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/micro/copy_bw.hip -o /tmp/copy_bw && /tmp/copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void copyk(const f4v* __restrict__ s, f4v* __restrict__ d, long n4) {
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f4v v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(s + i + u * stride) : s[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) __builtin_nontemporal_store(v[u], d + i + u * stride); else d[i + u * stride] = v[u];
    }
  }
  for (; i < n4; i += stride) d[i] = s[i];
}
template <int U, bool NT> float run(const f4v* s, f4v* d, long n4, int blocks) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e30f;
  for (int r = 0; r < 6; ++r) {
    hipEventRecord(a);
    hipLaunchKernelGGL((copyk<U, NT>), dim3(blocks), dim3(256), 0, 0, s, d, n4);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
  }
  return best;
}
int main() {
  const long n = 1L << 28, n4 = n / 4;
  f4v *s, *d; hipMalloc((void**)&s, n * 4); hipMalloc((void**)&d, n * 4);
  hipMemset(s, 1, n * 4); hipMemset(d, 0, n * 4);
  for (int blocks : {1024, 2048, 4096, 8192, 16384, 65536}) {
    printf("blocks %6d: U1 %.0f  U4 %.0f  U8 %.0f  U4nt %.0f  U8nt %.0f GB/s\n", blocks,
           2.0 * n * 4 / run<1, false>(s, d, n4, blocks) / 1e6, 2.0 * n * 4 / run<4, false>(s, d, n4, blocks) / 1e6,
           2.0 * n * 4 / run<8, false>(s, d, n4, blocks) / 1e6, 2.0 * n * 4 / run<4, true>(s, d, n4, blocks) / 1e6,
           2.0 * n * 4 / run<8, true>(s, d, n4, blocks) / 1e6);
  }
  for (int blocks : {16384, 32768, 65536, 131072, 262144})
    printf("blocks %6d: U1nt %.0f  U2nt %.0f  U2 %.0f GB/s\n", blocks, 2.0 * n * 4 / run<1, true>(s, d, n4, blocks) / 1e6,
           2.0 * n * 4 / run<2, true>(s, d, n4, blocks) / 1e6, 2.0 * n * 4 / run<2, false>(s, d, n4, blocks) / 1e6);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a); hipMemcpyAsync(d, s, n * 4, hipMemcpyDeviceToDevice, 0); hipEventRecord(b); hipEventSynchronize(b);
  hipEventRecord(a); hipMemcpyAsync(d, s, n * 4, hipMemcpyDeviceToDevice, 0); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("hipMemcpyAsync D2D: %.0f GB/s\n", 2.0 * n * 4 / ms / 1e6);
  return 0;
}
