// LDS read bandwidth per CU: 8 waves, each issuing conflict-free ds_read_b128 / ds_read_b64 / ds_read_b32 in a loop (tools/micro/run.sh)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int BYTES, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(float* out, int iters, long long* clk) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds + wave * 4096 + lane * BYTES;
  f4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const long long t0 = wall_clock64();
  const long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (BYTES == 16) {
      f4 x0, x1, x2, x3;
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(base) : "memory");
      a0 += x0; a1 += x1; a2 += x2; a3 += x3;
    } else if (BYTES == 8) {
      f2 x0, x1, x2, x3;
      asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:512\n\tds_read_b64 %2, %4 offset:1024\n\tds_read_b64 %3, %4 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(base) : "memory");
      a0[0] += x0[0]; a1[0] += x1[0]; a2[0] += x2[0]; a3[0] += x3[0];
    } else {
      float x0, x1, x2, x3;
      asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:256\n\tds_read_b32 %2, %4 offset:512\n\tds_read_b32 %3, %4 offset:768\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(base) : "memory");
      a0[0] += x0; a1[0] += x1; a2[0] += x2; a3[0] += x3;
    }
  }
  const long long c1 = clock64();
  const long long t1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + a0[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = t1 - t0; }
}
template <int BYTES, int WAVES>
void run(const char* name) {
  float* out; long long* clk; long long h[2];
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 16);
  const int iters = 20000;
  k<BYTES, WAVES><<<256, WAVES * 64>>>(out, iters, clk);
  hipDeviceSynchronize();
  k<BYTES, WAVES><<<256, WAVES * 64>>>(out, iters, clk);
  hipDeviceSynchronize();
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double bytes = (double)iters * 4 * 64 * BYTES * WAVES;
  printf("%-28s %d waves: %.1f B/clk/CU (shader clocks), %.1f B per 100MHz-tick\n", name, WAVES, bytes / h[0], bytes / h[1]);
}
int main() {
  run<16, 8>("ds_read_b128"); run<16, 4>("ds_read_b128"); run<16, 16>("ds_read_b128");
  run<8, 8>("ds_read_b64"); run<4, 8>("ds_read_b32");
  return 0;
}
